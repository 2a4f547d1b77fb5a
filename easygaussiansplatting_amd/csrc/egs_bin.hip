// Tile binning for gfx950 (CDNA4, wave64): getRects + the depth key (reference gsplatcu/kernel.cu:82-122, :73), the
// offsets scan (gausplat.cu:64), createKeys in depth order (kernel.cu:46-80), getRanges (kernel.cu:125-150), the packed
// 48-B records of the draw kernels (fetch2shared, kernel.cu:13-44) and the content stamps of the public splat / splatB pair.
#include "egs_raster.h"

#include <stdlib.h>

#include <algorithm>

namespace egs {

// ============================================================================
// binning
// ============================================================================
// getRects (reference kernel.cu:82-122) + the depth key of createKeys (kernel.cu:73); the fused forward
// kernel (egs_preprocess.hip) does the same through bin_count_one and skips this launch
__global__ __launch_bounds__(256) void k_bin_count(int n, BinParams p, const float* __restrict__ us,
                                                   int32_t* __restrict__ areas, float* __restrict__ depths,
                                                   uint4* __restrict__ cr,
                                                   uint32_t* __restrict__ dkeys, uint32_t* __restrict__ ids,
                                                   uint32_t* __restrict__ maxkey, uint32_t* __restrict__ sort_sup,
                                                   uint32_t sort_sup_words) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  for (uint32_t z = (uint32_t)i; z < sort_sup_words; z += gridDim.x * 256u) sort_sup[z] = 0u;   // for the depth sort
  uint32_t key = 0u;
  if (i < n) {
    uint4 rect;
    bool cull;
    const uint32_t cnt = bin_count_one(p, us[2 * (size_t)i], us[2 * (size_t)i + 1], (float)areas[2 * (size_t)i],
                                       (float)areas[2 * (size_t)i + 1], depths[i], rect, key, cull);
    if (cull) {  // the in-place contract of the reference (kernel.cu:114-119)
      depths[i] = EGS_BAD_MARKER;
      areas[2 * (size_t)i] = 0;
      areas[2 * (size_t)i + 1] = 0;
    }
    ids[i] = (uint32_t)i;
    // the seven-op surface returns the reference's lists: every tile of the rect (a full bitmap / an unculled big rect)
    uint4 c = make_uint4(0u, 0u, 0u, 0u);
    if (cnt) {
      const uint32_t w = rect.z - rect.x, h = rect.w - rect.y;
      c.x = rect.x | (rect.y << 16);
      c.y = w | (h << 16);
      if (w <= 4u && h <= 4u) {
        const unsigned long long row = (1ull << (2 * w)) - 1ull;
        unsigned long long bits = 0ull;
        for (uint32_t r = 0; r < 2 * h; ++r) bits |= row << (8 * r);
        c.z = (uint32_t)bits; c.w = (uint32_t)(bits >> 32);
      } else {
        c.y |= EGS_CR_BIG; c.z = cnt; c.w = 0u;
      }
    }
    cr[i] = c;
    dkeys[i] = key;
  }
  __shared__ uint32_t wm[4];
  block_max_key(key, maxkey, wm);  // upper bound of the depth keys: lets the radix sort skip all-zero high digits
}

// (content stamps: see "content stamp of the 2D Gaussians" above k_pack_records)
__device__ __forceinline__ uint32_t row_stamp(float ux, float uy, float c0, float c1, float c2, float al) {
  uint32_t h = __float_as_uint(ux) * 0x9E3779B1u;
  h = (h ^ (h >> 15)) + __float_as_uint(uy) * 0x85EBCA77u;
  h = (h ^ (h >> 13)) + __float_as_uint(c0) * 0xC2B2AE3Du;
  h = (h ^ (h >> 16)) + __float_as_uint(c1) * 0x27D4EB2Fu;
  h = (h ^ (h >> 15)) + __float_as_uint(c2) * 0x165667B1u;
  h = (h ^ (h >> 13)) + __float_as_uint(al) * 0x9E3779B1u;
  return h ^ (h >> 16);
}
// all 256 threads call this; `red` = 8 words of LDS; stamp[2 wg], stamp[2 wg + 1] receive the workgroup's two sums;
// ref / same (nullable): same[wg] = 1 iff they equal ref[2 wg], ref[2 wg + 1] (the stamps an earlier pass left)
__device__ __forceinline__ void block_stamp(uint32_t h, uint32_t* __restrict__ stamp, uint32_t* red,
                                            const uint32_t* __restrict__ ref = nullptr,
                                            uint8_t* __restrict__ same = nullptr) {
  uint32_t s1 = h, s2 = h * (2u * threadIdx.x + 1u);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { s1 += (uint32_t)__shfl_xor((int)s1, d, 64); s2 += (uint32_t)__shfl_xor((int)s2, d, 64); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s1; red[4 + (threadIdx.x >> 6)] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t a = red[0] + red[1] + red[2] + red[3], b = red[4] + 3u * red[5] + 5u * red[6] + 7u * red[7];
    stamp[2 * (size_t)blockIdx.x] = a;
    stamp[2 * (size_t)blockIdx.x + 1] = b;
    if (ref && same) same[blockIdx.x] = (ref[2 * (size_t)blockIdx.x] == a && ref[2 * (size_t)blockIdx.x + 1] == b) ? 1 : 0;
  }
}
// The seven-op surface's splat, tile-footprint policies: k_pack_records and k_bin_count as ONE pass over the 2D
// Gaussians -- the packed 48-B record of the draw kernels, getRects + depth key (kernel.cu:82-122, :73), and, new in
// round 4, the EXACT block masks of the fused path for the reference's UNCULLED lists: the Gaussian is emitted for
// every tile of its rect (gsid_per_patch stays bit-exact), each list value carrying the 4-bit mask of the 8x8 blocks
// its footprint alpha' >= alpha_skip can reach in that tile (EGS_CR_ALLTILES, egs_common.h) -- the draw kernels then
// evaluate 1.95 instead of 2.35 blocks per entry and skip outright the 11 % of the entries that reach none.
__global__ __launch_bounds__(256) void k_pack_bin(int n, BinParams p, float alpha_skip, const float* __restrict__ us,
                                                  const float* __restrict__ cinv, const float* __restrict__ alphas,
                                                  const float* __restrict__ colors, int32_t* __restrict__ areas,
                                                  float* __restrict__ depths, float4* __restrict__ rec,
                                                  uint4* __restrict__ cr, BinRec* __restrict__ br,
                                                  uint32_t* __restrict__ dkeys, uint32_t* __restrict__ ids,
                                                  uint32_t* __restrict__ maxkey, uint32_t* __restrict__ sort_sup,
                                                  uint32_t sort_sup_words, uint32_t* __restrict__ stamp,
                                                  uint8_t* __restrict__ visible) {
  // visible (nullable): depth > 0.2 AFTER the in-place cull below -- the mask GSFunction returns (gsmodel.py:50)
  const int i = blockIdx.x * 256 + threadIdx.x;
  for (uint32_t z = (uint32_t)i; z < sort_sup_words; z += gridDim.x * 256u) sort_sup[z] = 0u;   // for the depth sort
  uint32_t key = 0u, hst = 0u;
  // the 48-B records leave as full lines: deposited in LDS (row stride 5 x 16 B: conflict-free), stored as the
  // workgroup's one contiguous span (lane-strided 16-B pieces cost three times the write requests)
  __shared__ float4 st[256 * 5];
  float4 r3[3] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  if (i < n) {
    const float ux = us[2 * (size_t)i], uy = us[2 * (size_t)i + 1];
    const float c0 = cinv[3 * (size_t)i], c1 = cinv[3 * (size_t)i + 1], c2 = cinv[3 * (size_t)i + 2];
    const float al = alphas[i];
    make_record(ux, uy, c0, c1, c2, al, colors[3 * (size_t)i], colors[3 * (size_t)i + 1], colors[3 * (size_t)i + 2],
                0, 0, p.W, p.H, 0, alpha_skip, r3);
    hst = row_stamp(ux, uy, c0, c1, c2, al);
    uint4 rect;
    bool cull;
    const uint32_t cnt = bin_count_one(p, ux, uy, (float)areas[2 * (size_t)i], (float)areas[2 * (size_t)i + 1],
                                       depths[i], rect, key, cull);
    const float depth_in = depths[i];
    if (cull) {  // the in-place contract of the reference (kernel.cu:114-119)
      depths[i] = EGS_BAD_MARKER;
      areas[2 * (size_t)i] = 0;
      areas[2 * (size_t)i + 1] = 0;
    }
    if (visible) visible[i] = (cull ? EGS_BAD_MARKER : depth_in) > 0.2f;
    ids[i] = (uint32_t)i;
    uint4 c = make_uint4(0u, 0u, 0u, 0u);
    if (cnt) {
      const BinRec b = make_binrec(ux, uy, c0, c1, c2, al, alpha_skip, true, rect, cnt);
      const uint32_t w = b.wh & 0xFFFFu, h = b.wh >> 16;
      if (w <= 4u && h <= 4u) {
        const unsigned long long bits = foot_bitmap(b);     // (all blocks when not cullable, none when alpha < skip)
        c = make_uint4(b.xy, b.wh | EGS_CR_ALLTILES, (uint32_t)bits, (uint32_t)(bits >> 32));
      } else {
        const bool walk = b.m < __int_as_float(0x7f800000);
        c = make_uint4(b.xy, b.wh | EGS_CR_BIG, cnt, walk ? 2u : 0u);
        if (walk) {
          float4* o = reinterpret_cast<float4*>(br + i);
          o[0] = make_float4(b.ux, b.uy, b.A, b.Bh);
          o[1] = make_float4(b.C, b.m, __uint_as_float(b.xy), __uint_as_float(b.wh));
        }
      }
    }
    cr[i] = c;
    dkeys[i] = key;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) st[threadIdx.x * 5 + j] = r3[j];
  __shared__ uint32_t wm[4];
  block_max_key(key, maxkey, wm);     // (its barrier also orders the deposits above)
  if (stamp) {   // content stamps for the splatB that may follow (see row_stamp)
    __shared__ uint32_t red[8];
    block_stamp(hst, stamp, red);
  }
  {
    const int base = blockIdx.x * 256, rows = min(256, n - base);
    float4* __restrict__ d4 = rec + 3 * (size_t)base;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int f = (int)threadIdx.x + 256 * j;
      if (f < 3 * rows) { const int rr = f / 3; d4[f] = st[rr * 5 + (f - 3 * rr)]; }
    }
  }
}

// ---- offsets of the Gaussians' patch runs, in depth order -------------------------------------------------
// The depth sort moves (key, id) pairs only; what the binning needs of a Gaussian afterwards is its footprint record
// (32 bytes) and its patch count.  Both are gathered ONCE into depth order -- by the last scatter pass of the depth
// sort, next to its stores -- and the scan kernels (counts) and k_bin_emit (records) stream contiguous arrays.  (The
// first version gathered counts[ids[j]] in both scan kernels and rects[ids[j]] in k_bin_emit: three dependent
// gathers through the sorted ids, 2.6-4x the algorithmic traffic by the PMC counters.)
__global__ __launch_bounds__(256) void k_bin_scan_partials(const uint32_t* __restrict__ cnt_sorted, int64_t n,
                                                           uint32_t* __restrict__ partials) {
  __shared__ uint32_t sm[4];
  const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_IPT;
  uint32_t s = 0;
  if (base + SC_IPT <= n) {
    const uint4 a = *reinterpret_cast<const uint4*>(cnt_sorted + base), b = *reinterpret_cast<const uint4*>(cnt_sorted + base + 4);
    s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
  } else {
#pragma unroll
    for (int k = 0; k < SC_IPT; ++k)
      if (base + k < n) s += cnt_sorted[base + k];
  }
  s = wave_inclusive_scan(s);
  if ((threadIdx.x & 63) == 63) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(256) void k_bin_scan_apply(const uint32_t* __restrict__ cnt_sorted, int64_t n,
                                                        const uint32_t* __restrict__ partials,
                                                        uint32_t* __restrict__ out, uint32_t* __restrict__ total,
                                                        uint32_t* __restrict__ total_host) {
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
  if (base + SC_IPT <= n) {
    const uint4 a = *reinterpret_cast<const uint4*>(cnt_sorted + base), b = *reinterpret_cast<const uint4*>(cnt_sorted + base + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < SC_IPT; ++k) v[k] = (base + k < n) ? cnt_sorted[base + k] : 0u;
  }
#pragma unroll
  for (int k = 0; k < SC_IPT; ++k) s += v[k];
  uint32_t blocksum;
  uint32_t ex = block256_exclusive_scan(s, sm, &blocksum) + prefix;
  if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    *total = prefix + blocksum;
    if (total_host) *total_host = prefix + blocksum;   // the mailbox slot (page-locked host memory), no copy
  }
  if (base + SC_IPT <= n) {
    uint4 a, b;
    a.x = ex; a.y = a.x + v[0]; a.z = a.y + v[1]; a.w = a.z + v[2];
    b.x = a.w + v[3]; b.y = b.x + v[4]; b.z = b.y + v[5]; b.w = b.z + v[6];
    *reinterpret_cast<uint4*>(out + base) = a;
    *reinterpret_cast<uint4*>(out + base + 4) = b;
  } else {
#pragma unroll
    for (int k = 0; k < SC_IPT; ++k) {
      const int64_t i = base + k;
      if (i < n) out[i] = ex;
      ex += v[k];
    }
  }
}

// createKeys (reference kernel.cu:46-80) in depth-sorted Gaussian order; the depth half of the key is
// implicit in the emission order.  The reference (and the first version here) lets every thread loop over
// its own rect: lanes idle while the largest rect of the wave finishes and every store instruction is 64
// scattered dwords (measured HBM traffic 149 MB for 33 MB of output).  Here the workgroup's 256 Gaussians
// own ONE contiguous output span (offsets are a prefix sum): output slot s finds its owner by binary
// search over the 256 offsets in LDS, so all lanes work and consecutive lanes write consecutive addresses.
// The Gaussian is emitted for the tiles its compact bin record names (egs_common.h): for a rect of at most 4 x 4 tiles
// slot r of its run is the r-th tile (row-major) with a block set in the record's bitmap -- bit arithmetic; a big
// cullable rect is walked row by row with its full footprint record (foot_row, the function that COUNTED its tiles in
// k_preprocess_fwd).  with_masks: the list value carries the tile's 4-bit block mask above the Gaussian index.
__global__ __launch_bounds__(256) void k_bin_emit(int n, int gx, const uint32_t* __restrict__ ids,
                                                  const uint32_t* __restrict__ offsets,
                                                  const uint4* __restrict__ cr_sorted,
                                                  const BinRec* __restrict__ br,
                                                  uint32_t* __restrict__ tkeys, uint32_t* __restrict__ gsid,
                                                  uint32_t cap, int32_t* __restrict__ ranges, int n_ranges,
                                                  int with_masks, uint32_t* __restrict__ sort_sup,
                                                  uint32_t sort_sup_words) {
  __shared__ uint32_t s_off[257];   // offsets relative to the workgroup's first one; [256] = span length
  __shared__ uint32_t s_g[256];
  __shared__ uint4 s_cr[256];
  const int tid = threadIdx.x;
  const int j = blockIdx.x * 256 + tid;
  for (int i = j; i < n_ranges; i += gridDim.x * 256) ranges[i] = 0;   // tiles without patches: (0, 0), as the reference
  for (uint32_t z = (uint32_t)j; z < sort_sup_words; z += gridDim.x * 256u) sort_sup[z] = 0u;   // for the tile sort
  uint32_t off = 0, g = 0;
  uint4 c = make_uint4(0u, 0u, 0u, 0u);
  if (j < n) {
    g = ids[j];
    c = cr_sorted[j];
    off = offsets[j];
  }
  // first offset of the workgroup (thread 0 always has a valid j) and the span length
  __shared__ uint32_t s_first, s_last;
  if (tid == 0) s_first = off;
  const int last = min(255, n - 1 - blockIdx.x * 256);
  if (tid == last) s_last = off + cr_count(c);
  __syncthreads();
  const uint32_t first = s_first;
  s_off[tid] = (j < n) ? off - first : 0xFFFFFFFFu;   // lanes past the end never own a slot
  s_g[tid] = g;
  s_cr[tid] = c;
  const uint32_t span = s_last - first;
  __syncthreads();
  for (uint32_t s0 = tid; s0 < span; s0 += 256) {
    // owner = last t with s_off[t] <= s0 (Gaussians without patches share their successor's offset and
    // are skipped by taking the LAST such t: it is the only one with cnt > 0 covering s0)
    int lo = 0;
#pragma unroll
    for (int step = 128; step >= 1; step >>= 1)
      if (s_off[lo + step] <= s0) lo += step;     // s_off[lo + step] with lo + step <= 255
    uint32_t r = s0 - s_off[lo];
    const uint4 cc = s_cr[lo];
    const int x0 = (int)(cc.x & 0xFFFFu), y0 = (int)(cc.x >> 16);
    const int w = (int)(cc.y & 0xFFFFu);
    uint32_t tile = 0u, mask = 0xFu;
    if (cc.y & EGS_CR_TILEMAP) {                    // <= 8 x 8 tiles: the r-th set bit of the tile bitmap
      const unsigned long long tb = ((unsigned long long)cc.w << 32) | cc.z;
      int ty = 0, tx = 0;
      uint32_t rb = 0u;
      bool found = false;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint32_t bq = (uint32_t)(tb >> (8 * q)) & 0xFFu;
        const uint32_t pc = (uint32_t)__popc(bq);
        if (!found) {
          if (r < pc) { rb = bq; ty = q; found = true; }
          else r -= pc;
        }
      }
      found = false;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (((rb >> q) & 1u) && !found) {
          if (r == 0u) { tx = q; found = true; }
          else --r;
        }
      }
      tile = (uint32_t)(y0 + ty) * (uint32_t)gx + (uint32_t)(x0 + tx);
      const BinRec b = br[s_g[lo]];                 // the block mask: the two slabs of this tile only
      const Foot f = foot_setup(b);
      const SlabPx sa = foot_slab(f, (y0 + ty) * EGS_TILE), sb = foot_slab(f, (y0 + ty) * EGS_TILE + 8);
      mask = foot_mask(sa, sb, x0 + tx);
    } else if (!(cc.y & EGS_CR_BIG)) {
      const unsigned long long blocks = ((unsigned long long)cc.w << 32) | cc.z;
      int ty = 0, tx = 0;
      if (cc.y & EGS_CR_ALLTILES) {                 // the reference's list: every tile of the rect, row-major
        ty = (int)(r / (uint32_t)w);
        tx = (int)(r - (uint32_t)ty * (uint32_t)w);
      } else {
      const unsigned long long tb = cr_tile_bits(blocks);
      uint32_t rb = 0u;
      bool found = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {                 // the tile row that holds the r-th tile
        const uint32_t bq = (uint32_t)(tb >> (16 * q)) & 0x55u;
        const uint32_t pc = (uint32_t)__popc(bq);
        if (!found) {
          if (r < pc) { rb = bq; ty = q; found = true; }
          else r -= pc;
        }
      }
      found = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {                 // the r-th set tile of that row
        if (((rb >> (2 * q)) & 1u) && !found) {
          if (r == 0u) { tx = q; found = true; }
          else --r;
        }
      }
      }
      tile = (uint32_t)(y0 + ty) * (uint32_t)gx + (uint32_t)(x0 + tx);
      mask = ((uint32_t)(blocks >> (16 * ty + 2 * tx)) & 3u) | (((uint32_t)(blocks >> (16 * ty + 8 + 2 * tx)) & 3u) << 2);
    } else if (cc.w == 0u) {                        // big rect, not cullable: every tile, every block
      const uint32_t ry = r / (uint32_t)w, rx = r - ry * (uint32_t)w;
      tile = (uint32_t)(y0 + (int)ry) * (uint32_t)gx + (uint32_t)x0 + rx;
    } else if (cc.w == 2u) {                        // big rect, the reference's list: every tile, masks from the footprint
      const uint32_t ry = r / (uint32_t)w, rx = r - ry * (uint32_t)w;
      tile = (uint32_t)(y0 + (int)ry) * (uint32_t)gx + (uint32_t)x0 + rx;
      const BinRec b = br[s_g[lo]];
      mask = 0u;
      if (!(b.m < 0.f)) {                           // (m < 0: alpha < alpha_skip, blends nowhere)
        const Foot f = foot_setup(b);
        SlabPx sa, sb;
        int tlo, thi;
        foot_row(f, y0 + (int)ry, sa, sb, tlo, thi);
        const int tx = x0 + (int)rx;
        if (tx >= tlo && tx <= thi) mask = foot_mask(sa, sb, tx);
      }
    } else {                                        // big cullable rect: emitted row by row by its wave, below
      continue;
    }
    if (first + s0 >= cap) break;   // (only when the buffers were sized from an earlier call: see egs_splat_draw_rec_dev)
    tkeys[first + s0] = tile;
    gsid[first + s0] = with_masks ? (s_g[lo] | (mask << EGS_GSID_BITS)) : s_g[lo];
  }
  // Big cullable rects (more than 8 x 8 tiles, footprint-culled): the tiles of a row are the interval foot_row names,
  // so a slot can only find its tile by summing the row widths in front of it.  Done per SLOT (the first version: every
  // slot walked the rows from the top) a screen-filling Gaussian costs rows x tiles footprint evaluations -- 68 x 7820
  // at 1080p: k_bin_emit 17 -> 515 us on a scene with 160 of them (profiles/r5_skewed_baseline.json).  Now the WAVE
  // that holds such a Gaussian emits it: lane r evaluates row r once (the two slabs, the interval), a wave scan
  // gives the rows' offsets inside the run, and the rows are then written one after the other by all 64 lanes --
  // rows + tiles / 64 steps, consecutive lanes on consecutive addresses.
  {
    const int lane = tid & 63, wbase = tid & ~63;
    unsigned long long bigs = __ballot(j < n && (c.y & EGS_CR_BIG) && c.w == 1u && c.z != 0u);
    while (bigs != 0ull) {
      const int t = wbase + (int)__builtin_ctzll(bigs);
      bigs &= bigs - 1ull;
      const uint4 cc = s_cr[t];
      const uint32_t gg = s_g[t];
      const uint32_t obase = first + s_off[t];
      const int y0 = (int)(cc.x >> 16), h = (int)((cc.y & EGS_CR_WH_MASK) >> 16);
      const BinRec b = br[gg];
      const Foot f = foot_setup(b);
      uint32_t run = 0u;                           // tiles of the rows above
      for (int rb = 0; rb < h; rb += 64) {
        SlabPx sa, sb;
        sa.pl = sb.pl = 0x7fffffff; sa.pr = sb.pr = (int)0x80000000;
        int tlo = 1, thi = 0;
        if (rb + lane < h) foot_row(f, y0 + rb + lane, sa, sb, tlo, thi);
        const uint32_t wd = thi >= tlo ? (uint32_t)(thi - tlo + 1) : 0u;
        const uint32_t inc = wave_inclusive_scan(wd);
        const uint32_t ex = run + inc - wd;
        const int rows = min(64, h - rb);
        for (int q = 0; q < rows; ++q) {
          const int qlo = __shfl(tlo, q, 64), qhi = __shfl(thi, q, 64);
          const uint32_t qoff = (uint32_t)__shfl((int)ex, q, 64);
          SlabPx qa, qb;
          qa.pl = __shfl(sa.pl, q, 64); qa.pr = __shfl(sa.pr, q, 64);
          qb.pl = __shfl(sb.pl, q, 64); qb.pr = __shfl(sb.pr, q, 64);
          for (int tx = qlo + lane; tx <= qhi; tx += 64) {
            const uint32_t o = obase + qoff + (uint32_t)(tx - qlo);
            if (o < cap) {
              tkeys[o] = (uint32_t)(y0 + rb + q) * (uint32_t)gx + (uint32_t)tx;
              gsid[o] = with_masks ? (gg | (foot_mask(qa, qb, tx) << EGS_GSID_BITS)) : gg;
            }
          }
        }
        run += (uint32_t)__shfl((int)inc, 63, 64);
      }
    }
  }
}

// getRanges (reference kernel.cu:125-150; its P==1 hole is closed here)
__global__ __launch_bounds__(256) void k_tile_ranges(int64_t P, const uint32_t* __restrict__ tkeys,
                                                     int32_t* __restrict__ ranges,
                                                     const uint32_t* __restrict__ n_dev,
                                                     const uint32_t* __restrict__ masked, int32_t* __restrict__ plain,
                                                     int32_t* __restrict__ zero, int zero_words,
                                                     int32_t* __restrict__ walk_word, uint32_t* __restrict__ walk_host) {
  // zero (nullable): words cleared on the side -- the bins and counters of the segment plan that may follow
  for (int z = blockIdx.x * 256 + threadIdx.x; z < zero_words; z += gridDim.x * 256) zero[z] = 0;
  // walk_word (nullable): the longest walk the previous render on this stream gathered (complete: stream order) goes to
  // the host's hint word; -1 = nothing gathered since the last publication
  // ([0] the longest walk, [1] the longest list: a PAIR from one render -- the host compares them)
  if (walk_word && blockIdx.x == 0 && threadIdx.x == 0) {
    const int w = walk_word[0], l = walk_word[1];
    if (w >= 0 && l >= 0 && walk_host) { walk_host[0] = (uint32_t)l; walk_host[1] = (uint32_t)w; }
    walk_word[0] = -1; walk_word[1] = -1;
  }
  // masked / plain (nullable pair, seven-op surface): gsid_per_patch as the reference returns it -- the sorted list
  // values without their block masks -- written on the way (this kernel is a chain of latencies: the 8 bytes per
  // patch ride along; as a launch of its own, k_strip_masks, they cost 6-8 us)
  if (n_dev) P = min(P, (int64_t)*n_dev);
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;   // four keys per thread: one dwordx4
  if (p0 >= P) return;
  if (plain) {
    if (p0 + 4 <= P) {
      uint4 v = *reinterpret_cast<const uint4*>(masked + p0);
      v.x &= EGS_GSID_MASK; v.y &= EGS_GSID_MASK; v.z &= EGS_GSID_MASK; v.w &= EGS_GSID_MASK;
      *reinterpret_cast<uint4*>(plain + p0) = v;
    } else {
      for (int64_t q = p0; q < P; ++q) plain[q] = (int32_t)(masked[q] & EGS_GSID_MASK);
    }
  }
  uint32_t k[5];
  k[0] = p0 > 0 ? tkeys[p0 - 1] : 0u;
  if (p0 + 4 <= P) {
    const uint4 v = *reinterpret_cast<const uint4*>(tkeys + p0);
    k[1] = v.x; k[2] = v.y; k[3] = v.z; k[4] = v.w;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) k[1 + i] = (p0 + i < P) ? tkeys[p0 + i] : 0u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t p = p0 + i;
    if (p < P) {
      const uint32_t cur = k[1 + i];
      if (p == 0) ranges[2 * (size_t)cur] = 0;
      else if (k[i] != cur) {
        ranges[2 * (size_t)k[i] + 1] = (int32_t)p;
        ranges[2 * (size_t)cur] = (int32_t)p;
      }
      if (p == P - 1) ranges[2 * (size_t)cur + 1] = (int32_t)P;
    }
  }
}

// gsid_per_patch as the reference returns it: the list values without their block masks
__global__ __launch_bounds__(256) void k_strip_masks(int64_t P, const uint32_t* __restrict__ n_dev,
                                                     const uint32_t* __restrict__ masked, int32_t* __restrict__ plain) {
  if (n_dev) P = min(P, (int64_t)*n_dev);
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (p0 >= P) return;
  if (p0 + 4 <= P) {
    uint4 v = *reinterpret_cast<const uint4*>(masked + p0);
    v.x &= EGS_GSID_MASK; v.y &= EGS_GSID_MASK; v.z &= EGS_GSID_MASK; v.w &= EGS_GSID_MASK;
    *reinterpret_cast<uint4*>(plain + p0) = v;
  } else {
    for (int64_t q = p0; q < P; ++q) plain[q] = (int32_t)(masked[q] & EGS_GSID_MASK);
  }
}

// ---- content stamp of the 2D Gaussians a masked list was built from ---------------------------------------------
// splat and the splatB that follows it are two independent calls of the reference's API; what splat leaves for splatB
// (the list with block masks) is valid only if splatB is given the SAME us / cinv2ds / alphas values.  No pointer or
// version comparison can know that (tensor.data writes, other libraries' kernels), so both pack kernels -- which read
// those values anyway -- leave a stamp per workgroup of 256 Gaussians: two position-dependent 32-bit sums over a hash of
// the six floats' bits.  k_pair_validate compares the two stamp arrays (and the caller's list with the kept one) on
// the device; k_draw_bwd takes the masks only if everything matched and otherwise walks the caller's own list with the
// per-entry box test: the result never depends on what was kept.
// kept[i] stays as it is iff it is the caller's entry plain[i] with a mask AND its Gaussian's block of 256 has the same
// stamp now as in the forward pass (same[block], written by k_pack_records: a 4-KB table at 1 M Gaussians); otherwise it
// becomes plain[i] with ALL four blocks set -- a mask that is valid for any data (the blocks are then decided by the
// exponent test alone).  k_draw_bwd needs no flag: it walks `kept` either way, and what it walks is the caller's list.
__global__ __launch_bounds__(256) void k_pair_fix(int64_t P, uint32_t* __restrict__ kept,
                                                  const int32_t* __restrict__ plain,
                                                  const uint8_t* __restrict__ same, uint32_t n) {
  // n: Gaussians (same[] has one byte per 256 of them): a list value beyond it -- a stale or foreign gsid tensor of the
  // right length -- is never looked up; the entry degrades to the all-blocks mask like any other mismatch
  const int64_t p0 = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x);
  if (p0 >= P) return;
  uint32_t k[4], q[4];
  if (p0 + 4 <= P) {
    const uint4 kv = *reinterpret_cast<const uint4*>(kept + p0);
    const uint4 qv = *reinterpret_cast<const uint4*>(plain + p0);
    k[0] = kv.x; k[1] = kv.y; k[2] = kv.z; k[3] = kv.w;
    q[0] = qv.x; q[1] = qv.y; q[2] = qv.z; q[3] = qv.w;
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) { k[t] = (p0 + t < P) ? kept[p0 + t] : 0u; q[t] = (p0 + t < P) ? (uint32_t)plain[p0 + t] : 0u; }
  }
  bool changed = false;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint32_t gid = q[t] & EGS_GSID_MASK;
    const bool good = (k[t] & EGS_GSID_MASK) == q[t] && gid < n && same[gid >> 8] != 0;
    if (!good) { k[t] = (q[t] & EGS_GSID_MASK) | (0xFu << EGS_GSID_BITS); changed = true; }
  }
  if (changed) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (p0 + t < P) kept[p0 + t] = k[t];
  }
}

// 48-byte packed 2D record per Gaussian: one aligned gather (3 x dwordx4)
// instead of the reference's four (fetch2shared, kernel.cu:13-44).
//   A = {u.x, u.y, qxx, qxy}   B = {qyy, alpha, col.r, col.g}   C = {col.b, c1, c2, thr}
// (qxx, qxy, qyy) = -0.5*log2(e) * (cinv.x, 2 cinv.y, cinv.z): the conic pre-scaled so that
//   power = qxx dx dx + qyy dy dy + qxy dx dy = log2 of exp(-maha/2), no per-pixel scaling.
// thr = log2(alpha_skip / alpha): alpha' = alpha 2^power >= alpha_skip  <=>  power >= thr, so
//   the skip test of kernel.cu:246 is made BEFORE the exponential and v_exp_f32 (3x the
//   cost of a plain VALU op on gfx950) is only issued for blocks that do blend.
// tile footprint (gsplatcu):   c1 = ex, c2 = ey -- half extents of the axis-aligned box
//   around u outside of which alpha' < alpha_skip is CERTAIN:  alpha' >= skip  =>
//   maha <= m* = 2 ln(alpha/skip) and maha >= dx^2 / Sigma_xx, so |dx| <= sqrt(m* Sigma_xx).
//   The draw kernels skip whole 8x8 pixel blocks that this box cannot reach; the pixels
//   skipped are exactly pixels the reference would `continue` on (kernel.cu:246), so the
//   result is unchanged.  Slack (x1.01 + 0.05 px) covers float rounding; a non positive-
//   definite cinv or skip == 0 disables the cull (extent = +inf).
// pixel-box footprint (forward_cpu): c1 = x0 | x1<<16, c2 = y0 | y1<<16 (gausplat.py:212-215)
__global__ __launch_bounds__(256) void k_pack_records(int n, int W, int H, int footprint, float alpha_skip,
                                                      const float* __restrict__ us,
                                                      const float* __restrict__ cinv,
                                                      const float* __restrict__ alphas,
                                                      const float* __restrict__ colors,
                                                      const int32_t* __restrict__ areas,
                                                      float4* __restrict__ rec, uint32_t* __restrict__ stamp,
                                                      const uint32_t* __restrict__ stamp_ref, uint8_t* __restrict__ same) {
  // stamp (nullable): content stamps per workgroup (see row_stamp); stamp_ref / same: compared on the spot
  const int i = blockIdx.x * 256 + threadIdx.x;
  uint32_t h = 0u;
  if (i < n) {
    const float ux = us[2 * (size_t)i], uy = us[2 * (size_t)i + 1];
    const float c0 = cinv[3 * (size_t)i], c1 = cinv[3 * (size_t)i + 1], c2 = cinv[3 * (size_t)i + 2];
    const float al = alphas[i];
    make_record(ux, uy, c0, c1, c2, al, colors[3 * (size_t)i], colors[3 * (size_t)i + 1],
                colors[3 * (size_t)i + 2], footprint == 1 ? areas[2 * (size_t)i] : 0,
                footprint == 1 ? areas[2 * (size_t)i + 1] : 0, W, H, footprint, alpha_skip, rec + 3 * (size_t)i);
    h = row_stamp(ux, uy, c0, c1, c2, al);
  }
  if (stamp) {   // (kernel argument: uniform)
    __shared__ uint32_t red[8];
    block_stamp(h, stamp, red, stamp_ref, same);
  }
}

// ============================================================================
// host side
// ============================================================================
size_t bin_ws_bytes(int n) {
  const size_t N = (size_t)(n > 0 ? n : 1);
  return 2 * align_up(N * 16, 256) + align_up(N * 32, 256) + 6 * align_up(N * 4, 256) + scan_ws_bytes(n) +
         sort_ws_bytes(n) + align_up((64 + N / 256 + 1) * 4, 256) + 4096;
}
bool bin_carve(void* ws, size_t bytes, int n, BinLayout* L) {
  Carver cv(ws, bytes);
  const size_t N = (size_t)(n > 0 ? n : 1);
  L->cr = cv.take<uint4>(N);
  L->cr_sorted = cv.take<uint4>(N);
  L->br = cv.take<BinRec>(N);
  L->cnt_sorted = cv.take<uint32_t>(N);
  L->dkeys = cv.take<uint32_t>(N);
  L->dkeys_alt = cv.take<uint32_t>(N);
  L->ids = cv.take<uint32_t>(N);
  L->ids_alt = cv.take<uint32_t>(N);
  L->offsets = cv.take<uint32_t>(N);
  L->scan_partials = cv.take<uint32_t>(scan_ws_bytes(n) / 4);
  L->maxkey = cv.take<uint32_t>(64 + div_up(N, 256));  // [0] = max depth key, [1..] per-workgroup partials
  return sort_ws_carve(cv, n, &L->sort) && cv.ok();
}


int bin_emit(int n, int gx, const BinLayout& B, uint32_t* tkeys, uint32_t* gsid, uint32_t cap, int32_t* ranges,
             int n_ranges, int with_masks, uint32_t* sort_sup, uint32_t sort_sup_words, hipStream_t s) {
  EGS_LAUNCH("k_bin_emit", k_bin_emit, dim3(div_up(n, 256)), dim3(256), s, n, gx, B.ids, B.offsets, B.cr_sorted, B.br,
             tkeys, gsid, cap, ranges, n_ranges, with_masks, sort_sup, sort_sup_words);
  EGS_LAUNCH_OK();
  return 0;
}

int tile_ranges(int64_t P, const uint32_t* tkeys, int32_t* ranges, const uint32_t* n_dev, const uint32_t* masked,
                int32_t* plain, hipStream_t s, int32_t* zero, int zero_words, int32_t* walk_word, uint32_t* walk_host) {
  EGS_LAUNCH("k_tile_ranges", k_tile_ranges, dim3(div_up(P, 1024)), dim3(256), s, P, tkeys, ranges, n_dev, masked, plain,
             zero, zero_words, walk_word, walk_host);
  EGS_LAUNCH_OK();
  return 0;
}

int pack_records(int n, int width, int height, int footprint, float alpha_skip, const float* us, const float* cinv2ds,
                 const float* alphas, const float* colors, const int32_t* areas, float4* rec, uint32_t* stamp,
                 const uint32_t* stamp_ref, uint8_t* same, hipStream_t s) {
  EGS_LAUNCH("k_pack_records", k_pack_records, dim3(div_up(n, 256)), dim3(256), s, n, width, height, footprint,
             alpha_skip, us, cinv2ds, alphas, colors, areas, rec, stamp, stamp_ref, same);
  EGS_LAUNCH_OK();
  return 0;
}

}  // namespace egs

using namespace egs;

extern "C" size_t egs_splat_bin_ws_bytes(int n) { return bin_ws_bytes(n); }
static int splat_bin_impl(int n, int width, int height, const float* us, int32_t* areas, float* depths,
                          const EgsPolicy* pol, int key_bits_hint, void* ws_bin, size_t ws_bin_bytes,
                          uint32_t* total_patches, uint32_t* host_totals, void* stream);

extern "C" int egs_splat_bin(int n, int width, int height, const float* us, int32_t* areas, float* depths,
                             const EgsPolicy* pol, int key_bits_hint, void* ws_bin, size_t ws_bin_bytes,
                             uint32_t* total_patches, void* stream) {
  return splat_bin_impl(n, width, height, us, areas, depths, pol, key_bits_hint, ws_bin, ws_bin_bytes, total_patches,
                        nullptr, stream);
}

// the same, the kernels also storing {P, max depth key} into a page-locked mailbox slot (egs_mailbox_slot)
extern "C" int egs_splat_bin_mb(int n, int width, int height, const float* us, int32_t* areas, float* depths,
                                const EgsPolicy* pol, int key_bits_hint, void* ws_bin, size_t ws_bin_bytes,
                                uint32_t* total_patches, uint32_t* host_totals, void* stream) {
  return splat_bin_impl(n, width, height, us, areas, depths, pol, key_bits_hint, ws_bin, ws_bin_bytes, total_patches,
                        host_totals, stream);
}

// egs_pack_records + egs_splat_bin(_mb) in one pass over the 2D Gaussians (k_pack_bin), tile-footprint policies with a
// skip threshold only: the lists that egs_splat_draw_rec* then emits (flags = EGS_DRAW_MASKED_LISTS) are the
// reference's, their values carry exact block masks.
extern "C" int egs_splat_bin_pack(int n, int width, int height, const float* us, const float* cinv2ds,
                                  const float* alphas, const float* colors, int32_t* areas, float* depths,
                                  const EgsPolicy* pol, int key_bits_hint, void* ws_bin, size_t ws_bin_bytes,
                                  uint32_t* total_patches, uint32_t* host_totals, void* rec, uint32_t* stamp,
                                  uint8_t* visible, void* stream) {
  // visible (nullable, n bytes): depth > 0.2 after this call's in-place cull (the mask of gsmodel.py:50)
  // stamp (nullable, egs_pair_stamp_words(n) words): content stamps of us / cinv2ds / alphas for a later
  // egs_pack_records_validate
  EGS_CHECK_ARG(n >= 0 && width > 0 && height > 0 && pol && total_patches);
  EGS_CHECK_ARG(width < 32768 && height < 32768);
  EGS_CHECK_ARG(pol->footprint == 0 && pol->alpha_skip > 0.f && n < (1 << EGS_GSID_BITS));
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    EGS_HIP(hipMemsetAsync(total_patches, 0, 8, s));
    if (host_totals) EGS_HIP(hipMemcpyAsync(host_totals, total_patches, 8, hipMemcpyDeviceToHost, s));
    return 0;
  }
  EGS_CHECK_ARG(us && cinv2ds && alphas && colors && areas && depths && ws_bin && rec);
  EGS_CHECK_ARG(((uintptr_t)rec & 15) == 0);
  BinLayout L;
  if (!bin_carve(ws_bin, ws_bin_bytes, n, &L)) {
    set_error(EGS_ERR_WORKSPACE, "bin workspace too small", __FILE__, __LINE__);
    return EGS_ERR_WORKSPACE;
  }
  const BinParams p = make_bin_params(width, height, pol);
  EGS_LAUNCH("k_pack_bin", k_pack_bin, dim3(div_up(n, 256)), dim3(256), s, n, p, pol->alpha_skip, us, cinv2ds, alphas,
             colors, areas, depths, (float4*)rec, L.cr, L.br, L.dkeys, L.ids, L.maxkey, L.sort.sup,
             (uint32_t)L.sort.sup_words, stamp, visible);
  EGS_LAUNCH_OK();
  return splat_bin_after_count(n, key_bits_hint, ws_bin, ws_bin_bytes, total_patches, stream, host_totals);
}

// plain[i] = masked[i] & EGS_GSID_MASK for the first min(count, *count_dev) list values (count_dev nullable)
extern "C" int egs_strip_list_masks(int64_t count, const uint32_t* count_dev, const void* masked, int32_t* plain,
                                    void* stream) {
  EGS_CHECK_ARG(count >= 0);
  if (count == 0) return 0;
  EGS_CHECK_ARG(masked && plain && (((uintptr_t)masked | (uintptr_t)plain) & 15) == 0);
  EGS_LAUNCH("k_strip_masks", k_strip_masks, dim3(div_up(count, 1024)), dim3(256), (hipStream_t)stream, count,
             count_dev, (const uint32_t*)masked, plain);
  EGS_LAUNCH_OK();
  return 0;
}

static int splat_bin_impl(int n, int width, int height, const float* us, int32_t* areas, float* depths,
                          const EgsPolicy* pol, int key_bits_hint, void* ws_bin, size_t ws_bin_bytes,
                          uint32_t* total_patches, uint32_t* host_totals, void* stream) {
  EGS_CHECK_ARG(n >= 0 && width > 0 && height > 0 && pol && total_patches);
  EGS_CHECK_ARG(width < 32768 && height < 32768);
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {  // the reference dereferences patch_offset_per_gs[-1] here (gausplat.cu:67)
    EGS_HIP(hipMemsetAsync(total_patches, 0, 8, s));
    if (host_totals) EGS_HIP(hipMemcpyAsync(host_totals, total_patches, 8, hipMemcpyDeviceToHost, s));
    return 0;
  }
  EGS_CHECK_ARG(us && areas && depths && ws_bin);
  BinLayout L;
  if (!bin_carve(ws_bin, ws_bin_bytes, n, &L)) {
    set_error(EGS_ERR_WORKSPACE, "bin workspace too small", __FILE__, __LINE__);
    return EGS_ERR_WORKSPACE;
  }
  const BinParams p = make_bin_params(width, height, pol);
  EGS_LAUNCH("k_bin_count", k_bin_count, dim3(div_up(n, 256)), dim3(256), s, n, p, us, areas, depths, L.cr,
             L.dkeys, L.ids, L.maxkey, L.sort.sup, (uint32_t)L.sort.sup_words);
  EGS_LAUNCH_OK();
  return splat_bin_after_count(n, key_bits_hint, ws_bin, ws_bin_bytes, total_patches, stream, host_totals);
}

namespace egs {
BinParams make_bin_params(int width, int height, const EgsPolicy* pol, bool cull_lists) {
  BinParams p;
  p.W = width; p.H = height;
  p.gx = div_up(width, EGS_TILE); p.gy = div_up(height, EGS_TILE);
  p.footprint = pol->footprint; p.far_cull = pol->far_cull; p.depth_key = pol->depth_key;
  p.mutate = (pol->footprint == 0);
  // footprint culling needs the skip test it is derived from (kernel.cu:246) and the tile footprint rule
  p.cull_lists = cull_lists && pol->footprint == 0 && pol->alpha_skip > 0.f;
  return p;
}

bool bin_count_outputs(void* ws_bin, size_t ws_bin_bytes, int n, BinCountOut* out) {
  BinLayout L;
  if (!bin_carve(ws_bin, ws_bin_bytes, n, &L)) return false;
  out->cr = L.cr; out->br = L.br; out->dkeys = L.dkeys; out->ids = L.ids; out->maxkey = L.maxkey;
  out->sort_sup = L.sort.sup; out->sort_sup_words = (uint32_t)L.sort.sup_words;
  return true;
}

int splat_bin_after_count(int n, int key_bits_hint, void* ws_bin, size_t ws_bin_bytes, uint32_t* total_patches,
                          void* stream, uint32_t* host_totals) {
  hipStream_t s = (hipStream_t)stream;
  BinLayout L;
  if (!bin_carve(ws_bin, ws_bin_bytes, n, &L)) {
    set_error(EGS_ERR_WORKSPACE, "bin workspace too small", __FILE__, __LINE__);
    return EGS_ERR_WORKSPACE;
  }
  // Only the depth-key bits the caller expects to be significant are sorted (hint from the previous
  // call's max key, which comes back in total_patches[1]); if the hint turns out too small the
  // caller re-runs the stage with hint = 32.  Within the launched passes, digits that are zero
  // in every key still degenerate to copies (maxkey check on the device).
  const int end_bit = (key_bits_hint <= 0 || key_bits_hint > 32) ? 32 : key_bits_hint;
  // (the largest depth key -- total_patches[1], and the mailbox slot's second word -- comes out of the first
  // pass's rowscan kernel)
  int rc = radix_sort(n, L.dkeys, L.ids, L.dkeys_alt, L.ids_alt, 0, end_bit, L.sort, s, L.maxkey, nullptr, L.maxkey,
                      div_up(n, 256), total_patches + 1, host_totals ? host_totals + 1 : nullptr, L.cr, L.cr_sorted,
                      L.cnt_sorted);
  if (rc) return rc;
  if (sort_passes(0, end_bit) & 1) {  // odd pass count: bring the result back to the primary buffers
    EGS_HIP(hipMemcpyAsync(L.dkeys, L.dkeys_alt, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    EGS_HIP(hipMemcpyAsync(L.ids, L.ids_alt, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  }
  const int nb = div_up(n, SC_TILE);
  EGS_LAUNCH("k_bin_scan_partials", k_bin_scan_partials, dim3(nb), dim3(256), s, L.cnt_sorted, (int64_t)n,
             L.scan_partials);
  EGS_LAUNCH("k_bin_scan_apply", k_bin_scan_apply, dim3(nb), dim3(256), s, L.cnt_sorted, (int64_t)n, L.scan_partials,
             L.offsets, total_patches, host_totals);
  EGS_LAUNCH_OK();
  return 0;
}
}  // namespace egs

// The packed 2D records of the draw kernels as a caller-held buffer: gsplatcu.splat packs them ONCE, draws from them
// (egs_splat_draw_rec*) and keeps them for the splatB call that follows with the same tensors (egs_splat_bwd_rec) --
// the seven-op surface otherwise packs twice per training step (2 x 20 us at 1 M Gaussians).
extern "C" int egs_pack_records(int n, int width, int height, const float* us, const float* cinv2ds,
                                const float* alphas, const float* colors, const int32_t* areas, const EgsPolicy* pol,
                                void* rec, void* stream) {
  EGS_CHECK_ARG(n >= 0 && width > 0 && height > 0 && pol);
  if (n == 0) return 0;
  EGS_CHECK_ARG(us && cinv2ds && alphas && colors && rec && (areas || pol->footprint != 1));
  EGS_CHECK_ARG(((uintptr_t)rec & 15) == 0);
  return pack_records(n, width, height, pol->footprint, pol->alpha_skip, us, cinv2ds, alphas, colors, areas, (float4*)rec,
                      nullptr, nullptr, nullptr, (hipStream_t)stream);
}

// words of a content-stamp array for n Gaussians: two per workgroup of 256, + ceil(workgroups / 4) for the byte table
// egs_pack_records_validate keeps behind its own stamps
extern "C" size_t egs_pair_stamp_words(int n) {
  const size_t nwg = (size_t)div_up(n > 0 ? n : 1, 256);
  return 2 * nwg + (nwg + 3) / 4 + 4;
}

// splatB's half of the content-validated pairing: pack the records from the tensors splatB was given, stamp
// them (stamp_b), and make the kept list (the one the forward draw walked, with masks) agree with them: every entry
// that is not the caller's own entry (plain) or whose Gaussian sits in a block of 256 whose stamps differ from the
// forward pass's (stamp_a, egs_splat_bin_pack) is replaced by the caller's entry with all four blocks set.
extern "C" int egs_pack_records_validate(int n, int width, int height, const float* us, const float* cinv2ds,
                                         const float* alphas, const float* colors, const EgsPolicy* pol, void* rec,
                                         const uint32_t* stamp_a, uint32_t* stamp_b, int64_t patches, void* kept,
                                         const int32_t* plain, void* stream) {
  EGS_CHECK_ARG(n > 0 && width > 0 && height > 0 && pol && pol->footprint == 0 && n < (1 << EGS_GSID_BITS));
  EGS_CHECK_ARG(us && cinv2ds && alphas && colors && rec && stamp_a && stamp_b && patches >= 0);
  EGS_CHECK_ARG(patches == 0 || (kept && plain && (((uintptr_t)kept | (uintptr_t)plain) & 15) == 0));
  EGS_CHECK_ARG(((uintptr_t)rec & 15) == 0 && (((uintptr_t)stamp_a | (uintptr_t)stamp_b) & 7) == 0);
  hipStream_t s = (hipStream_t)stream;
  uint8_t* same = (uint8_t*)(stamp_b + 2 * (size_t)div_up(n, 256));     // one byte per block of 256 Gaussians
  EGS_LAUNCH("k_pack_records", k_pack_records, dim3(div_up(n, 256)), dim3(256), s, n, width, height, 0,
             pol->alpha_skip, us, cinv2ds, alphas, colors, (const int32_t*)nullptr, (float4*)rec, stamp_b, stamp_a, same);
  if (patches > 0)
    EGS_LAUNCH("k_pair_fix", k_pair_fix, dim3(div_up(patches, 1024)), dim3(256), s, patches, (uint32_t*)kept, plain,
               (const uint8_t*)same, (uint32_t)n);
  EGS_LAUNCH_OK();
  return 0;
}
