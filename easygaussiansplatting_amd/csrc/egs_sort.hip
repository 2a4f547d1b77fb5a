// Stable LSD radix sort of (u32 key, u32 value) pairs and the exclusive prefix sum of the binning stage, for gfx950
// (CDNA4, wave64).  Replaces the reference's thrust::sort_by_key / thrust::inclusive_scan (gsplatcu/gausplat.cu:64, 82;
// Thrust/CUB of the CUDA toolkit, version unpinned): results are exact integers, parity is defined by the spec -- a
// stable sort, ties keep their input order.
#include "egs_raster.h"

#include <algorithm>

namespace egs {

// ============================================================================
// stable LSD radix sort, digits of up to 8 bits (dmask), (u32 key, u32 value)
// ============================================================================
constexpr int RS_THREADS = 256;
// items per thread: 16 (4096-item tiles) for long arrays; 8 for short ones, where 4096-item tiles would
// leave fewer workgroups than there are CUs (1 M depth keys = 245 tiles)
#ifndef EGS_RS_SHORT           // A/B knob
#define EGS_RS_SHORT (5 << 19)
#endif
constexpr int64_t RS_SHORT = EGS_RS_SHORT;         // <= 2.6 M items: 2048-item tiles (measured: 4 M patches prefer 4096)
static int rs_ipt(int64_t n) { return n <= RS_SHORT ? 8 : 16; }

// `maxkey` (nullable, device): upper bound of all keys.  A pass whose digit is 0 for every key
// ((*maxkey >> shift) == 0) is the identity permutation: hist returns at once and
// scatter degenerates to a coalesced copy.
//
// Per pass TWO kernels, not three.  The per-workgroup digit counts go to hist[block][digit] (block-major: coalesced
// rows) and, by one atomic per non-empty digit, into the sums of SUPERBLOCKS of RS_SB workgroups, sup[superblock][digit]
// (zeroed by the kernel that ran before the sort).  The scatter kernel then builds its own offsets from at most
// RS_SB - 1 hist rows of its superblock and the <= 32 superblock rows: no row-scan launch in between (it was a
// 256-workgroup kernel over <= 1 MB, 5-6 us of launch and drain four times per step).
//
// mk_parts != NULL (first pass of the depth sort): one extra workgroup folds the per-workgroup maxima of the keys
// (maxkey[1 + i], left by the kernel that produced them) into maxkey[0] -- the later passes test it -- and into
// up to two more places (mk_out: next to P in device memory; mk_host: the page-locked mailbox slot).  The first
// pass itself never consults maxkey: with shift 0 it could only detect "every key is 0", where the pass is the
// identity anyway.
constexpr int RS_SB = 32;           // workgroups per superblock
template <int RS_IPT>
__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const uint32_t* __restrict__ keys, int64_t n,
                                                           int shift, uint32_t dmask, int nblocks,
                                                           uint32_t* __restrict__ hist, uint32_t* __restrict__ sup,
                                                           const uint32_t* __restrict__ maxkey,
                                                           const uint32_t* __restrict__ n_dev,
                                                           uint32_t* __restrict__ mk_parts, int nparts,
                                                           uint32_t* __restrict__ mk_out,
                                                           uint32_t* __restrict__ mk_host) {
  constexpr int RS_TILE = RS_THREADS * RS_IPT;
  __shared__ uint32_t h[256];
  __shared__ uint32_t sm[4];
  const int tid = threadIdx.x;
  // (mk_parts: the launch has ONE MORE workgroup, the first; it does the fold and nothing else -- as a side job of
  // a counting workgroup the 3906 partial maxima made that workgroup the last to finish, 2.5 us after the others)
  const int blk = mk_parts ? (int)blockIdx.x - 1 : (int)blockIdx.x;
  if (mk_parts && blockIdx.x == 0) {
    uint32_t mk = 0u;
    for (int i = tid; i < nparts; i += 256) mk = max(mk, mk_parts[1 + i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mk = max(mk, (uint32_t)__shfl_xor((int)mk, d, 64));
    if ((tid & 63) == 0) sm[tid >> 6] = mk;
    __syncthreads();
    if (tid == 0) {
      const uint32_t m = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
      mk_parts[0] = m;
      if (mk_out) *mk_out = m;
      if (mk_host) *mk_host = m;
    }
    return;
  }
  if (!mk_parts && maxkey && ((*maxkey >> shift) == 0u)) return;
  if (n_dev) n = min(n, (int64_t)*n_dev);   // `n` is a capacity: the real count is on the device
  h[tid] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blk * RS_TILE;
  if (base < n) {
    // all keys of the thread requested before the first LDS atomic (clamped addresses, no branches: with the load
    // inside the guarded loop every one of the 16 rounds waited for its own round trip to memory)
    const uint32_t* kb = keys + base;
    const uint32_t rlast = (uint32_t)min(n - base, (int64_t)RS_TILE) - 1u;
    uint32_t k[RS_IPT];
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) k[r] = kb[min((uint32_t)(r * RS_THREADS + tid), rlast)];
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r)
      if ((uint32_t)(r * RS_THREADS + tid) <= rlast) atomicAdd(&h[(k[r] >> shift) & dmask], 1u);
  }
  __syncthreads();
  const uint32_t c = h[tid];
  hist[(size_t)blk * 256 + tid] = c;          // block-major: row = workgroup
  if (c) atomicAdd(&sup[(size_t)(blk / RS_SB) * 256 + tid], c);
}

// Scatter with local reordering: every item's stable rank inside the workgroup's 4096-item tile
// is found with per-wave ballot multi-split (deterministic, no LDS atomics), the tile is written
// digit-sorted into LDS, and then streamed out so that consecutive lanes write consecutive
// addresses inside each digit run (coalesced) instead of 64 scattered dwords per instruction.
// EXTRA (compiled in, so that the plain instance keeps its registers: 4 / 6 resident workgroup-waves per SIMD instead
// of 3 / 5 with the code below merely present): 1 = gather records on the way out (last pass of the depth sort).
// (getRanges folded into the last scatter pass of the tile sort -- one thread per digit run, atomicMin / Max at the run
// ends -- was built, bit-identical, and measured 5 us SLOWER per step than the separate k_tile_ranges pass: LAB 3.2.)
template <int RS_IPT, int EXTRA>
__global__ __launch_bounds__(RS_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_radix_scatter(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int64_t n, int shift, uint32_t dmask,
    int nblocks, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ sup,
    const uint32_t* __restrict__ maxkey, const uint32_t* __restrict__ n_dev,
    const uint4* __restrict__ gsrc, uint4* __restrict__ gdst, uint32_t* __restrict__ cdst) {
  // gsrc != NULL (last pass of the depth sort): the 16-byte compact bin record gsrc[value] of every item is gathered
  // into sorted order on the way out (gdst[pos]) -- the random reads hide behind this kernel's stores instead of
  // heading the dependent scan kernel that follows -- and its patch count goes to cdst[pos]: the two scan kernels
  // then stream 4 bytes per Gaussian
  if (n_dev) n = min(n, (int64_t)*n_dev);
  constexpr int RS_TILE = RS_THREADS * RS_IPT;       // items per workgroup
  constexpr int RS_WAVE_ITEMS = EGS_WAVE * RS_IPT;   // contiguous items per wave
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // (wave: a scalar)
  const int64_t blockbase = (int64_t)blockIdx.x * RS_TILE;
  if (blockbase >= n) return;                  // (the launch covers the capacity of the list)
  if (maxkey && ((*maxkey >> shift) == 0u)) {  // identity pass: plain copy
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
      const int64_t idx = blockbase + r * RS_THREADS + tid;
      if (idx < n) {
        const uint32_t v = vals_in[idx];
        keys_out[idx] = keys_in[idx];
        vals_out[idx] = v;
        if constexpr (EXTRA == 1) { const uint4 c = gsrc[v]; gdst[idx] = c; cdst[idx] = cr_count(c); }
      }
    }
    return;
  }
  __shared__ uint32_t wcount[4][256];  // per-wave running digit counters -> per-wave offsets
  __shared__ uint32_t dstart[256];     // first local slot of each digit inside this tile
  __shared__ uint32_t gadj[256];       // global position of local slot i with digit d: gadj[d] + i
  __shared__ uint32_t sm[4];
  __shared__ uint32_t skey[RS_TILE], sval[RS_TILE];
#pragma unroll
  for (int w = 0; w < 4; ++w) wcount[w][tid] = 0;
  const int64_t base = blockbase + (int64_t)wave * RS_WAVE_ITEMS;
  uint32_t key[RS_IPT], val[RS_IPT], rank[RS_IPT];
  // Global base of digit d = (sum of the totals of smaller digits) + (digit d in earlier workgroups): superblock sums
  // for the total and for the superblocks before this workgroup's, hist rows inside its superblock.  Wave w takes
  // rows w, w + 4, ... and every lane four digits (one dwordx4 per 1-KB row): for up to 32 superblocks the whole
  // prefix is ONE group of loads, issued before the keys -- the counter the waits use retires loads in order, so a
  // second dependent group behind the keys would wait for all of them.
  // All addresses are clamped instead of guarded: without branches the compiler counts the loads in flight exactly
  // and the ranking loop waits for key r only.
  uint4 tot = make_uint4(0u, 0u, 0u, 0u), bef = tot;
  {
    const int sb = blockIdx.x / RS_SB, nsb = (nblocks + RS_SB - 1) / RS_SB;
    const uint4* sup4 = reinterpret_cast<const uint4*>(sup) + lane;
    const uint4* hist4 = reinterpret_cast<const uint4*>(hist) + lane;
    uint4 v[8], w[8];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = sup4[(size_t)min(wave + 4 * r, nsb - 1) * 64];
#pragma unroll
    for (int r = 0; r < 8; ++r) w[r] = hist4[(size_t)min(sb * RS_SB + wave + 4 * r, (int)blockIdx.x) * 64];
    // ALL of the workgroup's keys and values are requested before anything waits.  (The ranking loop below goes
    // through LDS every round; with the loads inside that loop each of its 16 rounds waited for its own global
    // round trip.)
    // (32-bit offsets from the workgroup's scalar base: one address register per round, shared by keys and values)
    const uint32_t* kb = keys_in + blockbase;
    const uint32_t* vb = vals_in + blockbase;
    const uint32_t rlast = (uint32_t)min(n - blockbase, (int64_t)RS_TILE) - 1u;
    uint32_t off[RS_IPT];
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) off[r] = min((uint32_t)(wave * RS_WAVE_ITEMS + r * EGS_WAVE + lane), rlast);
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) key[r] = kb[off[r]];
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) val[r] = vb[off[r]];   // (not waited for by the ranking)
    __builtin_amdgcn_sched_barrier(0);   // (the instruction scheduler keeps this order)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int q = wave + 4 * r;
      const uint32_t mt = q < nsb ? ~0u : 0u, mb = q < sb ? ~0u : 0u;
      tot.x += v[r].x & mt; tot.y += v[r].y & mt; tot.z += v[r].z & mt; tot.w += v[r].w & mt;
      bef.x += v[r].x & mb; bef.y += v[r].y & mb; bef.z += v[r].z & mb; bef.w += v[r].w & mb;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint32_t mb = (sb * RS_SB + wave + 4 * r < (int)blockIdx.x) ? ~0u : 0u;
      bef.x += w[r].x & mb; bef.y += w[r].y & mb; bef.z += w[r].z & mb; bef.w += w[r].w & mb;
    }
    for (int q0 = wave + 32; q0 < nsb; q0 += 32) {   // more than 32 superblocks (lists beyond 4 M patches)
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = sup4[(size_t)min(q0 + 4 * r, nsb - 1) * 64];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int q = q0 + 4 * r;
        const uint32_t mt = q < nsb ? ~0u : 0u, mb = q < sb ? ~0u : 0u;
        tot.x += v[r].x & mt; tot.y += v[r].y & mt; tot.z += v[r].z & mt; tot.w += v[r].w & mt;
        bef.x += v[r].x & mb; bef.y += v[r].y & mb; bef.z += v[r].z & mb; bef.w += v[r].w & mb;
      }
    }
  }
  // the four waves' shares meet in LDS (skey / sval are free until the tile is reordered)
  reinterpret_cast<uint4*>(skey)[wave * 64 + lane] = tot;
  reinterpret_cast<uint4*>(sval)[wave * 64 + lane] = bef;
  __syncthreads();
  const uint32_t dtotal = skey[tid] + skey[256 + tid] + skey[512 + tid] + skey[768 + tid];
  const uint32_t before = sval[tid] + sval[256 + tid] + sval[512 + tid] + sval[768 + tid];
  const uint32_t dig_ex = block256_exclusive_scan(dtotal, sm, nullptr);
  const uint32_t gbase = dig_ex + before;

  // Ranking inside the wave.  peers = lanes of this wave holding the same digit (multi-split: one ballot per digit
  // bit); the lane's rank among them comes from mbcnt, the lowest peer moves the wave's counter of that digit.
  // The counters are read and written with wavefront-scope relaxed atomics: plain ds_read / ds_write, which one
  // wave issues in order -- a `volatile` pointer here turned them into flat loads and stores with system-scope
  // cache bits and a full wait after each, two LDS round trips through the flat path per round.
  uint32_t* wc = wcount[wave];
  const int nbits = __popc(dmask);
#pragma unroll
  for (int r = 0; r < RS_IPT; ++r) {
    const int64_t idx = base + r * EGS_WAVE + lane;
    const bool valid = idx < n;
    const uint32_t d = (key[r] >> shift) & dmask;
    const uint64_t vb = __ballot(valid);
    uint32_t plo = (uint32_t)vb, phi = (uint32_t)(vb >> 32);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      if (b >= nbits) break;                                           // (uniform: 13 tile bits are 7 + 6)
      const uint32_t m = (uint32_t)((int32_t)(d << (31 - b)) >> 31);   // 0 or ~0: this lane's bit b
      const uint64_t bal = __ballot(m != 0u);
      plo &= ~((uint32_t)bal ^ m);            // bit set: keep the lanes in bal, clear: keep the others
      phi &= ~((uint32_t)(bal >> 32) ^ m);
    }
    const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
    uint32_t prev = 0;
    if (valid) prev = __hip_atomic_load(&wc[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (valid && below == 0)   // lowest peer updates
      __hip_atomic_store(&wc[d], prev + (uint32_t)(__popc(plo) + __popc(phi)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    rank[r] = prev + below;
  }
  __syncthreads();
  uint32_t cnt = 0;
  {  // digit `tid`: exclusive scan of its per-wave counts, and its count in the whole tile
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t c = wcount[w][tid];
      wcount[w][tid] = cnt;
      cnt += c;
    }
  }
  const uint32_t ds = block256_exclusive_scan(cnt, sm, nullptr);  // (contains barriers)
  dstart[tid] = ds;
  gadj[tid] = gbase - ds;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_IPT; ++r) {
    const int64_t idx = base + r * EGS_WAVE + lane;
    if (idx < n) {
      const uint32_t d = (key[r] >> shift) & dmask;
      const uint32_t slot = dstart[d] + wcount[wave][d] + rank[r];
      skey[slot] = key[r];
      sval[slot] = val[r];
    }
  }
  __syncthreads();
  const int64_t rem = n - blockbase;
  const int nvalid = rem < RS_TILE ? (int)rem : RS_TILE;
  if constexpr (EXTRA == 1) {   // eight rounds of gathers in flight before their first store
    constexpr int G = 8;
#pragma unroll
    for (int r0 = 0; r0 < RS_IPT; r0 += G) {
      uint32_t ok[G], ov[G], op[G];
      uint4 oc[G];
#pragma unroll
      for (int r = 0; r < G; ++r) {
        const int slot = (r0 + r) * RS_THREADS + tid;
        ok[r] = 0u; ov[r] = 0u; op[r] = 0u; oc[r] = make_uint4(0u, 0u, 0u, 0u);
        if (slot < nvalid) {
          ok[r] = skey[slot];
          op[r] = gadj[(ok[r] >> shift) & dmask] + (uint32_t)slot;
          ov[r] = sval[slot];
          oc[r] = gsrc[ov[r]];
        }
      }
      // (one wait for all eight here; otherwise the compiler, which counts loads and stores on the same in-order
      // counter and gives up at the branches, waits for the previous round's STORES before each round)
#pragma unroll
      for (int r = 0; r < G; ++r) asm volatile("" ::"v"(oc[r].x), "v"(oc[r].y), "v"(oc[r].z), "v"(oc[r].w));
#pragma unroll
      for (int r = 0; r < G; ++r) {
        const int slot = (r0 + r) * RS_THREADS + tid;
        if (slot < nvalid) {
          keys_out[op[r]] = ok[r];
          vals_out[op[r]] = ov[r];
          gdst[op[r]] = oc[r];
          cdst[op[r]] = cr_count(oc[r]);
        }
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
      const int slot = r * RS_THREADS + tid;
      if (slot < nvalid) {
        const uint32_t k = skey[slot];
        const uint32_t pos = gadj[(k >> shift) & dmask] + (uint32_t)slot;
        keys_out[pos] = k;
        vals_out[pos] = sval[slot];
      }
    }
  }
}

static size_t sort_sup_words(int64_t n) {
  const int nb = n > 0 ? div_up(n, RS_THREADS * 8) : 1;   // sized for the smaller tile
  return (size_t)4 * div_up(nb, RS_SB) * 256;
}
size_t sort_ws_bytes(int64_t n) {
  const int nb = n > 0 ? div_up(n, RS_THREADS * 8) : 1;
  return align_up((size_t)256 * nb * 4, 256) + align_up(sort_sup_words(n) * 4, 256) + 512;
}
bool sort_ws_carve(Carver& cv, int64_t n, SortWs* w) {
  w->nblocks = n > 0 ? div_up(n, RS_THREADS * rs_ipt(n)) : 1;
  w->hist = cv.take<uint32_t>((size_t)256 * w->nblocks);
  w->sup_words = sort_sup_words(n);
  w->sup = cv.take<uint32_t>(w->sup_words);
  return cv.ok();
}
int sort_passes(int begin_bit, int end_bit) { return (end_bit - begin_bit + 7) / 8; }

// enqueue all passes; result ends in (keys,vals) if the pass count is even
int radix_sort(int64_t n, uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt, int begin_bit,
               int end_bit, const SortWs& w, hipStream_t s, const uint32_t* maxkey, const uint32_t* n_dev,
               uint32_t* mk_parts, int nparts, uint32_t* mk_out, uint32_t* mk_host, const uint4* gsrc, uint4* gdst,
               uint32_t* cdst) {
  // gsrc/gdst: gdst[j] = gsrc[value of the j-th item of the sorted sequence], written by the last pass
  if (n <= 0) return 0;
  uint32_t *ki = keys, *vi = vals, *ko = keys_alt, *vo = vals_alt;
  // the bits are spread evenly over the passes (13 tile bits = 7 + 6, not 8 + 5): fewer buckets per pass
  // mean longer coalesced runs out of every tile
  const int passes = sort_passes(begin_bit, end_bit);
  const int width = (end_bit - begin_bit + passes - 1) / passes;
  int pass = 0;
  for (int shift = begin_bit; shift < end_bit; shift += width) {
    const int nb = end_bit - shift < width ? end_bit - shift : width;  // the last digit may be narrower
    const uint32_t dmask = (1u << nb) - 1u;
    const bool first = shift == begin_bit && mk_parts != nullptr;      // this pass produces maxkey[0]
    const uint32_t* mk = first ? nullptr : maxkey;
    const bool last = shift + width >= end_bit;
    const uint4* gs = last ? gsrc : nullptr;
    uint32_t* sup = w.sup + (size_t)pass * div_up(w.nblocks, RS_SB) * 256;    // this pass's (zeroed) superblock sums
    uint32_t* mkp = first ? mk_parts : (uint32_t*)nullptr;
    if (rs_ipt(n) == 8)
      EGS_LAUNCH("k_radix_hist", k_radix_hist<8>, dim3(w.nblocks + (mkp ? 1 : 0)), dim3(RS_THREADS), s, ki, n, shift, dmask,
                 w.nblocks, w.hist, sup, mk, n_dev, mkp, nparts, mk_out, mk_host);
    else
      EGS_LAUNCH("k_radix_hist", k_radix_hist<16>, dim3(w.nblocks + (mkp ? 1 : 0)), dim3(RS_THREADS), s, ki, n, shift, dmask,
                 w.nblocks, w.hist, sup, mk, n_dev, mkp, nparts, mk_out, mk_host);
    ++pass;
#define EGS_SCATTER(IPT, EXTRA)                                                                                     \
  EGS_LAUNCH("k_radix_scatter", (k_radix_scatter<IPT, EXTRA>), dim3(w.nblocks), dim3(RS_THREADS), s, ki, vi, ko, vo, n, \
             shift, dmask, w.nblocks, w.hist, sup, mk, n_dev, gs, gdst, cdst)
    if (rs_ipt(n) == 8) {
      if (gs) EGS_SCATTER(8, 1); else EGS_SCATTER(8, 0);
    } else {
      if (gs) EGS_SCATTER(16, 1); else EGS_SCATTER(16, 0);
    }
#undef EGS_SCATTER
    uint32_t* t = ki; ki = ko; ko = t;
    t = vi; vi = vo; vo = t;
  }
  EGS_LAUNCH_OK();
  return 0;
}

// ============================================================================
// exclusive prefix sum of u32 with optional gather: out[i] = sum_{j<i} in[g[j]]
// ============================================================================

__global__ __launch_bounds__(256) void k_scan_partials(const uint32_t* __restrict__ in,
                                                       const uint32_t* __restrict__ gather, int64_t n,
                                                       uint32_t* __restrict__ partials) {
  __shared__ uint32_t sm[4];
  const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_IPT;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SC_IPT; ++k) {
    const int64_t i = base + k;
    if (i < n) s += in[gather ? gather[i] : i];
  }
  s = wave_inclusive_scan(s);
  if ((threadIdx.x & 63) == 63) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// second pass: every workgroup sums the partials of its predecessors itself (a few hundred values out
// of L2 -- cheaper than a separate single-workgroup spine launch), then scans its tile
__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* __restrict__ in,
                                                    const uint32_t* __restrict__ gather, int64_t n,
                                                    const uint32_t* __restrict__ partials,
                                                    uint32_t* __restrict__ out, uint32_t* __restrict__ total) {
  __shared__ uint32_t sm[4];
  __shared__ uint32_t s_prefix;
  uint32_t pre = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) pre += partials[i];
  pre = wave_inclusive_scan(pre);
  if ((threadIdx.x & 63) == 63) sm[threadIdx.x >> 6] = pre;
  __syncthreads();
  if (threadIdx.x == 0) s_prefix = sm[0] + sm[1] + sm[2] + sm[3];
  __syncthreads();
  const uint32_t prefix = s_prefix;
  const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_IPT;
  uint32_t v[SC_IPT];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SC_IPT; ++k) {
    const int64_t i = base + k;
    v[k] = (i < n) ? in[gather ? gather[i] : i] : 0u;
    s += v[k];
  }
  uint32_t blocksum;
  uint32_t ex = block256_exclusive_scan(s, sm, &blocksum) + prefix;
  if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = prefix + blocksum;
#pragma unroll
  for (int k = 0; k < SC_IPT; ++k) {
    const int64_t i = base + k;
    if (i < n) out[i] = ex;
    ex += v[k];
  }
}

size_t scan_ws_bytes(int64_t n) { return align_up((size_t)(n > 0 ? div_up(n, SC_TILE) : 1) * 4, 256) + 256; }

static int exclusive_scan(int64_t n, const uint32_t* in, const uint32_t* gather, uint32_t* out, uint32_t* total,
                          uint32_t* partials, hipStream_t s) {
  if (n <= 0) {
    if (total) EGS_HIP(hipMemsetAsync(total, 0, 4, s));
    return 0;
  }
  const int nb = div_up(n, SC_TILE);
  EGS_LAUNCH("k_scan_partials", k_scan_partials, dim3(nb), dim3(256), s, in, gather, n, partials);
  EGS_LAUNCH("k_scan_apply", k_scan_apply, dim3(nb), dim3(256), s, in, gather, n, partials, out, total);
  EGS_LAUNCH_OK();
  return 0;
}

}  // namespace egs

using namespace egs;

extern "C" size_t egs_sort_pairs_ws_bytes(int64_t n) { return sort_ws_bytes(n); }

extern "C" int egs_sort_pairs(int64_t n, uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt,
                              int begin_bit, int end_bit, void* ws, size_t ws_bytes, int* result_in_alt_host,
                              void* stream) {
  EGS_CHECK_ARG(n >= 0 && begin_bit >= 0 && end_bit <= 32 && begin_bit <= end_bit);
  if (result_in_alt_host) *result_in_alt_host = (n > 0) ? (sort_passes(begin_bit, end_bit) & 1) : 0;
  if (n == 0 || begin_bit == end_bit) {
    if (result_in_alt_host) *result_in_alt_host = 0;
    return 0;
  }
  EGS_CHECK_ARG(keys && vals && keys_alt && vals_alt && ws);
  Carver cv(ws, ws_bytes);
  SortWs w;
  if (!sort_ws_carve(cv, n, &w)) {
    set_error(EGS_ERR_WORKSPACE, "sort workspace too small", __FILE__, __LINE__);
    return EGS_ERR_WORKSPACE;
  }
  EGS_HIP(hipMemsetAsync(w.sup, 0, w.sup_words * 4, (hipStream_t)stream));   // (the binning kernels zero theirs on the side)
  return radix_sort(n, keys, vals, keys_alt, vals_alt, begin_bit, end_bit, w, (hipStream_t)stream);
}

extern "C" size_t egs_scan_ws_bytes(int64_t n) { return scan_ws_bytes(n); }

extern "C" int egs_exclusive_scan_u32(int64_t n, const uint32_t* in, const uint32_t* gather, uint32_t* out,
                                      uint32_t* total, void* ws, size_t ws_bytes, void* stream) {
  EGS_CHECK_ARG(n >= 0);
  EGS_CHECK_ARG(n == 0 || (in && out && ws));
  if (ws_bytes < scan_ws_bytes(n)) {
    set_error(EGS_ERR_WORKSPACE, "scan workspace too small", __FILE__, __LINE__);
    return EGS_ERR_WORKSPACE;
  }
  return exclusive_scan(n, in, gather, out, total, (uint32_t*)ws, (hipStream_t)stream);
}
