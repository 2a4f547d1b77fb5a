// The per-tile alpha-blend kernels for gfx950 (CDNA4, wave64): k_draw (reference gsplatcu/kernel.cu:152-271) and
// k_draw_bwd (kernel.cu:809-950), the dispatch order of their tiles (k_tile_order) and the launchers that pick the
// template instance of a raster policy.  One wave64 per 16x16 tile; see egs_raster.h for how this differs from the
// reference by design.
#include "egs_draw_device.h"

#include <stdlib.h>

namespace egs {

// default dispatch order of the tiles for the forward / backward draw kernel (k_tile_order modes)
#ifndef EGS_TILE_ORDER_F_DEFAULT
#define EGS_TILE_ORDER_F_DEFAULT 1
#endif
#ifndef EGS_TILE_ORDER_B_DEFAULT
#define EGS_TILE_ORDER_B_DEFAULT 1
#endif
#ifndef EGS_DRAWB_RED_DEFAULT
#define EGS_DRAWB_RED_DEFAULT 7
#endif

// Longest-list-first dispatch order of the tiles for the two draw kernels.  A tile is one wave whose run
// time is proportional to its list length (0 ... ~2x the mean on the 1 M scene); workgroups are handed to
// the SIMDs in index order, so with tiles in IMAGE order a launch ends with whichever SIMD drew the longest
// lists while the others idle.  Sorted by length (descending) the long tiles start first and the short ones
// fill the gaps (LPT scheduling); when every tile is resident at once (k_draw: 8 waves per SIMD) the
// sorted order is dealt out in a serpentine of `period` slots so that every SIMD receives one tile of each
// length stratum, alternately from its top and its bottom.
//   mode 1: one global order           mode 2: global, serpentine
//   mode 3: per XCD (tile row % 8 stays on XCD b % 8: horizontal neighbours share one L2), sorted
//   mode 4: per XCD, serpentine
// One workgroup: counting sort on (class, length) in LDS -- 8160 tiles take a few microseconds.
constexpr int TO_BINS = 1024;
constexpr int TO_REGS = 16;    // tiles per thread whose (bin, rank) stay in registers between the two passes
// sort key of tile t: its list length, or -- `work` given -- the work the forward draw kernel measured for it
__device__ __forceinline__ int tile_len(const int32_t* __restrict__ ranges, const int32_t* __restrict__ work, int t) {
  if (work) return work[t];
  const int2 r = reinterpret_cast<const int2*>(ranges)[t];
  return r.y - r.x;
}
__global__ __launch_bounds__(1024) void k_tile_order(const int32_t* __restrict__ ranges,
                                                     const int32_t* __restrict__ work, int T, int gx, int mode,
                                                     int period, int32_t* __restrict__ order, int ngrid,
                                                     const int32_t* __restrict__ walk = nullptr,
                                                     uint32_t* __restrict__ hint = nullptr) {
  // walk / hint (nullable): hint[1] receives the longest WALK of the camera's previous render (walk[T], next to its work),
  // hint[0] the longest list when the tiles are sorted by length -- page-locked words the host steers by (fused.py: long
  // walks take the segment path)
  // 8192 bins in all: one class of 8192 (global modes) or eight of 1024 (per-XCD modes).
  // ONE LDS atomic per tile: the returning add that counts a bin also hands the tile its rank inside the bin
  // (arrival order -- any order inside a bin will do); after the scan of the bins its slot is start + rank.
  // LDS atomics retire about one lane per clock whatever the conflicts, so the kernel costs ~T cycles per pass:
  // the first version's two passes took 9 us at 1080p and 45 us at 4K (32400 tiles).
  constexpr int NB = 8 * TO_BINS;
  __shared__ uint32_t bins[NB];
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t cbase[9];
  const int tid = threadIdx.x;
  const bool per_xcd = mode >= 3;
  const int cbins = per_xcd ? TO_BINS : NB;                       // bins per class
  // key -> bin: list lengths 1:1 (1:4 per XCD); the forward kernel's work measure is ~6x a length
  const int shift = (per_xcd ? 2 : 0) + (work ? 2 : 0);
  int lenr[TO_REGS];     // all loads in flight at once: the kernel is a chain of latencies, not of bytes
#pragma unroll
  for (int r = 0; r < TO_REGS; ++r) {
    const int t = tid + r * 1024;
    lenr[r] = t < T ? tile_len(ranges, work, t) : 0;
  }
  for (int i = tid; i < NB; i += 1024) bins[i] = 0u;
  if (per_xcd)   // classes are padded to the largest one: slots without a tile stay -1
    for (int i = tid; i < ngrid; i += 1024) order[i] = -1;
  if (hint) {
    int mx = 0;
    if (walk) { for (int t = tid; t < T; t += 1024) mx = max(mx, walk[t]); }
    else if (!work) {
#pragma unroll
      for (int r = 0; r < TO_REGS; ++r) mx = max(mx, lenr[r]);
      for (int t = tid + TO_REGS * 1024; t < T; t += 1024) mx = max(mx, tile_len(ranges, work, t));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d, 64));
    if ((tid & 63) == 0) wsum[tid >> 6] = (uint32_t)mx;
  }
  __syncthreads();
  if (hint && tid == 0 && (walk || !work)) {
    uint32_t mx = 0u;
    for (int w = 0; w < 16; ++w) mx = max(mx, wsum[w]);
    hint[walk ? 1 : 0] = mx;
  }
  __syncthreads();
  auto key_of = [&](int t, int len) {
    const int q = min(max(len, 0) >> shift, cbins - 1);
    const int cls = per_xcd ? ((t / gx) & 7) : 0;
    return cls * cbins + (cbins - 1 - q);
  };
  // pass 1: (bin, rank) per tile, packed 13 + 19 bits (T < 2^19: checked by the host)
  uint32_t kr[TO_REGS];
#pragma unroll
  for (int r = 0; r < TO_REGS; ++r) {
    const int t = tid + r * 1024;
    kr[r] = 0u;
    if (t < T) {
      const int key = key_of(t, lenr[r]);
      kr[r] = ((uint32_t)key << 19) | atomicAdd(&bins[key], 1u);
    }
  }
  // (tiles beyond TO_REGS * 1024 keep their (bin, rank) in the order buffer itself until pass 2)
  for (int t = tid + TO_REGS * 1024; t < T; t += 1024) {
    const int key = key_of(t, tile_len(ranges, work, t));
    order[t] = (int32_t)(((uint32_t)key << 19) | atomicAdd(&bins[key], 1u));
  }
  __syncthreads();
  {  // exclusive scan of the 8192 bins: thread t owns bins [8 t, 8 t + 8)
    uint32_t v[8], s = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = bins[8 * tid + k]; s += v[k]; }
    const uint32_t inc = wave_inclusive_scan(s);
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t pre = 0u;
    for (int w = 0; w < (tid >> 6); ++w) pre += wsum[w];
    uint32_t ex = pre + inc - s;
#pragma unroll
    for (int k = 0; k < 8; ++k) { bins[8 * tid + k] = ex; ex += v[k]; }
  }
  __syncthreads();
  if (tid < 8) cbase[tid] = per_xcd ? bins[tid * TO_BINS] : (tid == 0 ? 0u : (uint32_t)T);
  if (tid == 8) cbase[8] = (uint32_t)T;
  __syncthreads();
  const bool serp = (mode == 2 || mode == 4) && period > 0;
  auto slot_of = [&](uint32_t packed) {
    const int key = (int)(packed >> 19);
    const int cls = key / cbins;
    int r = (int)(bins[key] + (packed & 0x7FFFFu) - cbase[cls]);
    if (serp) {
      const int cnt = (int)(cbase[cls + 1] - cbase[cls]);
      const int st = r / period, ps = r - st * period;
      if (st & 1) r = st * period + (min(period, cnt - st * period) - 1 - ps);
    }
    return per_xcd ? 8 * r + cls : r;
  };
  // pass 2 for the tiles parked in the order buffer: read them ALL before any slot is written (a slot may be
  // another tile's parking place)
  constexpr int TO_TAIL = 24;      // up to (TO_REGS + TO_TAIL) * 1024 = 40960 tiles (a 4K image has 32400)
  uint32_t tail[TO_TAIL];
#pragma unroll
  for (int u = 0; u < TO_TAIL; ++u) {
    const int t = tid + (TO_REGS + u) * 1024;
    tail[u] = t < T ? (uint32_t)order[t] : 0u;
  }
  __syncthreads();
  if (per_xcd) {   // the parking places go back to "no tile" before the real slots are written
#pragma unroll
    for (int u = 0; u < TO_TAIL; ++u) {
      const int t = tid + (TO_REGS + u) * 1024;
      if (t < T) order[t] = -1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < TO_REGS; ++r) {
    const int t = tid + r * 1024;
    if (t < T) { const int slot = slot_of(kr[r]); if (slot < ngrid) order[slot] = t; }
  }
#pragma unroll
  for (int u = 0; u < TO_TAIL; ++u) {
    const int t = tid + (TO_REGS + u) * 1024;
    if (t < T) { const int slot = slot_of(tail[u]); if (slot < ngrid) order[slot] = t; }
  }
}
static_assert(TILE_ORDER_MAX_T == (TO_REGS + 24) * 1024, "what k_tile_order handles");

// The per-tile work measure of k_draw (sum of the four blocks' largest contributor index + twice the tile's)
// rebuilt from the `contrib` image, for a backward pass that was not handed the forward pass's record.
__global__ __launch_bounds__(64) void k_tile_work(int W, int H, int gx, const int32_t* __restrict__ contrib,
                                                  int32_t* __restrict__ work, int32_t* __restrict__ walk = nullptr) {
  const int tile = blockIdx.x, lane = threadIdx.x;
  const int tx0 = (tile % gx) * EGS_TILE, ty0 = (tile / gx) * EGS_TILE;
  int w = 0, wmax = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = tx0 + (lane & 7) + 8 * (k & 1), py = ty0 + (lane >> 3) + 8 * (k >> 1);
    int mx = (px < W && py < H) ? contrib[(size_t)py * W + px] : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d, 64));
    w += mx;
    wmax = max(wmax, mx);
  }
  if (lane == 0) { work[tile] = w + 2 * wmax; if (walk) walk[tile] = wmax; }
}
// ... and the tile's walk alone (its largest contributor index), for a splatB that rebuilds segment states
__global__ __launch_bounds__(64) void k_tile_walk(int W, int H, int gx, const int32_t* __restrict__ contrib,
                                                  int32_t* __restrict__ walk) {
  const int tile = blockIdx.x, lane = threadIdx.x;
  const int tx0 = (tile % gx) * EGS_TILE, ty0 = (tile / gx) * EGS_TILE;
  int mx = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = tx0 + (lane & 7) + 8 * (k & 1), py = ty0 + (lane >> 3) + 8 * (k >> 1);
    if (px < W && py < H) mx = max(mx, contrib[(size_t)py * W + px]);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d, 64));
  if (lane == 0) walk[tile] = mx;
}
// capacity of an order buffer: the per-XCD modes pad every class to the largest one
int tile_order_len(int gx, int gy) { return 8 * div_up(gy, 8) * gx; }

// ============================================================================
// draw: per-tile front-to-back blend                   (reference kernel.cu:152-271)
// ============================================================================
// Workgroup b runs on XCD b % 8 (observed dispatch order; speed only): give each
// XCD a contiguous band of tiles so that its private 4-MiB L2 serves 1/8 of the
// Gaussian records instead of all of them.  Bijective for any T.
__device__ __forceinline__ int xcd_tile(int b, const DrawParams& p) {
  if (p.order) {   // (a caller-held buffer: an index outside the image is treated as padding, never dereferenced)
    if (b >= p.ngrid) return -1;
    const int t = p.order[b];
    return (unsigned)t < (unsigned)p.T ? t : -1;
  }
  if (p.map_mode == 0) return b < p.T ? b : -1;
  const int xcd = b & 7, k = b >> 3;
  if (p.map_mode == 1) {
    if (b >= p.T) return -1;
    const int q = p.T >> 3, r = p.T & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  // mode 2: tile row ty belongs to XCD ty % 8 (balanced when list lengths vary smoothly
  // over the image, still row-coherent inside one L2); the grid is padded to
  // 8 * ceil(gy/8) * gx blocks and the surplus blocks exit.
  const int ty = xcd + 8 * (k / p.gx), tx = k % p.gx;
  return ty < p.gy ? ty * p.gx + tx : -1;
}
// Dynamic LDS requested only to CAP the number of resident tile-waves per CU (experiment knobs
// EGS_DRAW_LDS_PAD / EGS_DRAWB_LDS_PAD, bytes): fewer resident waves let the dispatcher hand the
// remaining tiles to whichever SIMD drains first (dynamic load balance).
static size_t draw_lds_pad(int which) {
  static const size_t pad[2] = {
      [] { const char* e = getenv("EGS_DRAW_LDS_PAD"); return e ? (size_t)atoi(e) : (size_t)0; }(),
      [] { const char* e = getenv("EGS_DRAWB_LDS_PAD"); return e ? (size_t)atoi(e) : (size_t)0; }()};
  return pad[which];
}
int draw_grid(const DrawParams& p) {
  if (p.order) return p.ngrid;
  return p.map_mode == 2 ? 8 * div_up(p.gy, 8) * p.gx : p.T;
}

// Policy is compiled in (BOX: pixel-box footprint; FLOOR: max(0,m); CLAMP: min(0.99,.));
// the two thresholds stay runtime scalars (SGPR operands of the compares).
//
// One wave64 per 16x16 tile.  The tile is walked as four 8x8 pixel blocks
// (k = 0..3, block (k&1, k>>1)); lane l owns pixel (l&7, l>>3) of each block.  Per
// list entry a block is skipped outright when the entry's certain-miss box (pack
// kernel) or pixel box does not reach it -- a wave-uniform branch.  The forward kernel
// evaluates the exponent as a polynomial about the tile centre (below), the backward kernel
// separably from the differences it also needs for the moments: cxx[bx] + cyy[by] +
// cxy[bx]*dy[by].  (Measured on gfx950,
// tools/ubench_valu.hip: v_pk_*_f32 costs exactly 2x a plain fp32 op, v_exp/v_rcp 3x,
// v_max/v_cmp->SGPR 1.6x -- so the kernels minimise instruction count, not pack.)
// A pixel that is finished or outside the image holds tau < tau_stop, so "still
// blending" is the one compare `tau >= stop`; the wave-uniform 4-bit `live` mask of
// blocks with an unfinished pixel is refreshed after every group of eight entries and gates
// the per-block scalar branches and the early exit.
template <bool BOX, bool FLOOR, bool CLAMP, bool SKIP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((!BOX && SKIP) ? 8 : 6, 8))) void k_draw(DrawParams p, int32_t* __restrict__ ranges,
                                             const int32_t* __restrict__ gsid,
                                             const float4* __restrict__ rec, float* __restrict__ image,
                                             int32_t* __restrict__ contrib, float* __restrict__ final_tau) {
  // staged entry: 12 floats (BOX: three b128 pieces) or 10 (two b128 + one b64: an entry's broadcast reads are
  // 50 of the ~130 SIMD cycles it costs, on an LDS pipe the CU's four SIMDs share; b64 is half a b128)
  // (the third piece keeps the 16-B slot stride: all three reads are immediate offsets from ONE address register)
  __shared__ float4 sA[64], sB[64], sC[64];
  const int lane = threadIdx.x;
  if (p.zero_buf) {   // every workgroup of the grid (padding ones included) clears its slice
    const uint32_t z0 = blockIdx.x * p.zero_per, z1 = min(p.zero_n4, z0 + p.zero_per);
    float4* __restrict__ zb = p.zero_buf;
    for (uint32_t i = z0 + lane; i < z1; i += 64) zb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int tile = xcd_tile(blockIdx.x, p);
  if (tile < 0) return;
  const int r0 = ranges[2 * (size_t)tile], r1 = ranges[2 * (size_t)tile + 1];
  const int n = r1 - r0;
  const int tx0 = (tile % p.gx) * EGS_TILE, ty0 = (tile / p.gx) * EGS_TILE;
  // pixel k = 2*by + bx of this lane: (tx0 + (lane&7) + 8 bx, ty0 + (lane>>3) + 8 by)
  const int pxb[2] = {tx0 + (lane & 7), tx0 + (lane & 7) + 8};
  const int pyb[2] = {ty0 + (lane >> 3), ty0 + (lane >> 3) + 8};
  if (n <= 0) {  // empty tile: image = 0, contrib = 0 and final_tau = 0 (NOT 1), exactly what the
                 // reference's early return leaves in its zero-filled outputs (kernel.cu:182)
    if (p.work_out && lane == 0) { p.work_out[tile] = 0; if (p.walk_out) p.walk_out[tile] = 0; walk_raise(p.walk_max, 0); }
    // a tile without patches still holds the (INT_MAX, 0) the binning initialised it with: (0, 0), as the reference
    if (lane == 0 && (r0 != 0 || r1 != 0)) { ranges[2 * (size_t)tile] = 0; ranges[2 * (size_t)tile + 1] = 0; }
    const size_t HW0 = (size_t)p.W * p.H;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int px = pxb[k & 1], py = pyb[k >> 1];
      if (px < p.W && py < p.H) {
        const size_t pix = (size_t)py * p.W + px;
        image[pix] = 0.f; image[HW0 + pix] = 0.f; image[2 * HW0 + pix] = 0.f;
        contrib[pix] = 0; final_tau[pix] = 0.f;
      }
    }
    return;
  }
  // The exponent of alpha' = exp2(e) is evaluated as a polynomial in the pixel's offset (X, Y) from the TILE
  // CENTRE:  e = c0 + c1 X + c2 Y + qxx XX + qxy XY + qyy YY  with the entry's  c0 = log2(alpha) + E(D),
  // (c1, c2) = grad E(D), D = tile centre - u, computed once per (tile, entry) by the lane that stages the
  // entry (64 entries in parallel), and the six monomials per-lane CONSTANTS (|X|, |Y| <= 7.5).  Five FMAs
  // per 8x8 block and no per-entry set-up (the separable form cxx[bx] + cyy[by] + cxy[bx] dy[by] cost 14
  // VALU instructions per entry before the first block); same accuracy as differences from u itself
  // (emulated in fp32 on the 1 M scene: mean |error| 6e-7, max 4e-5 in the log2 domain, either way).
  const float X[2] = {(float)(lane & 7) - 7.5f, (float)(lane & 7) + 0.5f};
  const float Y[2] = {(float)(lane >> 3) - 7.5f, (float)(lane >> 3) + 0.5f};
  const float XX[2] = {X[0] * X[0], X[1] * X[1]}, YY[2] = {Y[0] * Y[0], Y[1] * Y[1]};
  const float XY[4] = {X[0] * Y[0], X[1] * Y[0], X[0] * Y[1], X[1] * Y[1]};
  // A pixel is finished when its tau fell below tau_stop (kernel.cu:256-260): `tau >= stop` IS the
  // "still blending" test, so no separate done flag is kept.  Lanes outside the image start at -1.
  float tau[4], cr[4], cg[4], cb[4];
  int cont[4];
  int live = 0;  // wave-uniform: bit k set while block k still has an unfinished pixel
  const float stop = p.tau_stop, lskip = p.lskip;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    cont[k] = 0;
    tau[k] = ((pxb[k & 1] < p.W) && (pyb[k >> 1] < p.H)) ? 1.f : -1.f;
    cr[k] = 0.f; cg[k] = 0.f; cb[k] = 0.f;
    if (__any(tau[k] >= stop)) live |= 1 << k;
  }
  constexpr float L99 = -0.014499569695115089f;  // log2(0.99): min(0.99, a) == exp2(min(log2 a, L99))
  const float cx0 = (float)tx0 + 7.5f, cy0 = (float)ty0 + 7.5f;
  // alpha' >= alpha_skip (kernel.cu:246) in the exponent domain: e >= log2(skip), a kernel constant (SKIP =
  // the policy has a skip threshold, compiled in); without one only a NaN exponent fails the compare
  const float lthr = SKIP ? lskip : -INFINITY;
  // the list value of the NEXT chunk is fetched one chunk ahead: the staging of a chunk then pays one global
  // latency (the record gather), not two dependent ones
  int gnext = (lane < n) ? gsid[r0 + lane] : 0;
  for (int base = 0; base < n && live != 0; base += 64) {
    __syncthreads();  // single-wave workgroup: orders the LDS reads of the previous chunk
    int mymask = 0;   // reach mask of the entry THIS lane staged (lane j <-> entry base + j)
    const int gm = gnext;
    const int g = p.masked ? (int)((uint32_t)gm & EGS_GSID_MASK) : gm;
    if (base + 64 + lane < n) gnext = gsid[r0 + base + 64 + lane];
    if (base + lane < n) {
      float4 A = rec[3 * (size_t)g], B = rec[3 * (size_t)g + 1];
      const float4 C = rec[3 * (size_t)g + 2];
      const bool nanfix = p.nan_blend && nan_entry_fix(A, B);
      // the record's thr = log2(skip / alpha), +inf for an entry that never blends (alpha < skip, or
      // alpha < 0 when there is no skip test): such an entry reaches nothing
      if (C.w < INFINITY) mymask = p.masked ? (int)((uint32_t)gm >> EGS_GSID_BITS) : reach_mask<BOX>(A, C, tx0, ty0);
      if (nanfix && !BOX && !p.masked && C.w < INFINITY) mymask = 0xF;
      // alpha' = exp2(e), e = log2(alpha) + log2 exp(-maha/2) (F.5.1, common.cuh:85-88, pre-scaled conic):
      // no multiply by alpha; the floor (maha >= 0) and the 0.99 clamp are ONE min against `cap`
      const float la = SKIP ? lskip - C.w : __builtin_amdgcn_logf(B.y);
      float cap = 3.0e38f;
      if (FLOOR) cap = CLAMP ? fminf(la, L99) : la;
      else if (CLAMP) cap = L99;
      const float Dx = cx0 - A.x, Dy = cy0 - A.y;
      const float c0 = la + (A.z * Dx * Dx + A.w * Dx * Dy + B.x * Dy * Dy);
      const float c1 = 2.f * A.z * Dx + A.w * Dy, c2 = 2.f * B.x * Dy + A.w * Dx;
      sA[lane] = make_float4(A.z, A.w, B.x, cap);   // qxx, qxy, qyy, cap
      if constexpr (BOX) {
        sB[lane] = make_float4(c0, c1, c2, C.y);      // polynomial about the tile centre; x pixel box
        sC[lane] = make_float4(B.z, B.w, C.x, C.z);   // colour; y pixel box
      } else {
        sB[lane] = make_float4(c0, c1, c2, B.z);      // polynomial about the tile centre; red
        *reinterpret_cast<float2*>(&sC[lane]) = make_float2(B.w, C.x);   // green, blue
      }
    }
    __syncthreads();
    // The reach masks of eight consecutive entries packed into one dword (4 bits each), gathered into the
    // group's first lane through the LDS permute path (ds_bpermute: no VALU issue slot): the blend loop
    // reads ONE SGPR per group of eight entries, skips the whole group when none of them reaches a live
    // block, and is fully unrolled over the group -- LDS addresses are an immediate offset from one base,
    // no per-entry v_readlane / v_mov / loop counter.  (Entries past the end of the list staged mask 0.)
    int pk = mymask;
    pk |= __shfl_down(pk, 1, 64) << 4;
    pk |= __shfl_down(pk, 2, 64) << 8;
    pk |= __shfl_down(pk, 4, 64) << 16;
    const int m = __builtin_amdgcn_readfirstlane(min(64, n - base));
    for (int j0 = 0; j0 < m && live != 0; j0 += 8) {  // eight entries, then the live-mask refresh
    const uint32_t act = (uint32_t)__builtin_amdgcn_readlane(pk, j0) & ((uint32_t)live * 0x11111111u);
    if (act != 0u) {
    const int vidx0 = base + j0 + 1;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int reach = (int)((act >> (4 * t)) & 0xFu);
      if (reach != 0) {  // scalar branch: some live block is within reach of this entry
        const int j = j0 + t;
        const float4 Q = sA[j], P = sB[j];            // wave-uniform address: LDS broadcast
        float4 K;
        if constexpr (BOX) K = sC[j];
        else { const float2 gb = *reinterpret_cast<const float2*>(&sC[j]); K = make_float4(P.w, gb.x, gb.y, 0.f); }
        bool inx[2] = {true, true}, iny[2] = {true, true};
        if (BOX) {
          const uint32_t bx = __float_as_uint(P.w), by = __float_as_uint(K.w);
          const int x0 = bx & 0xFFFF, x1 = bx >> 16, y0 = by & 0xFFFF, y1 = by >> 16;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            inx[b] = (pxb[b] >= x0) && (pxb[b] < x1);
            iny[b] = (pyb[b] >= y0) && (pyb[b] < y1);
          }
        }
        const int idx = vidx0 + t;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int bx = k & 1, by = k >> 1;
          if (reach & (1 << k)) {  // scalar branch: the whole 8x8 block is live and in reach
            float e = fmaf(P.z, Y[by], P.x);
            e = fmaf(P.y, X[bx], e);
            e = fmaf(Q.z, YY[by], e);
            e = fmaf(Q.y, XY[k], e);
            e = fmaf(Q.x, XX[bx], e);
            // unfinished and alpha' >= alpha_skip; the cap cannot change the outcome of the skip test
            // (cap >= log2(skip) for every entry that blends at all), so it is applied to the hits only
            bool hit = (tau[k] >= stop) && (e >= lthr);
            if (BOX) hit = hit && inx[bx] && iny[by];
            if (hit) {
              if (FLOOR || CLAMP) e = min_hi(e, Q.w);
              const float w = tau[k] * __builtin_amdgcn_exp2f(e);  // F.5: tau alpha'
              cr[k] += w * K.x; cg[k] += w * K.y; cb[k] += w * K.z;
              tau[k] -= w;  // F.5.2: tau (1 - alpha')
              cont[k] = idx;
            }
          }
        }
      }
    }
    // Finished pixels fail `tau >= stop` on their own, so the live-block mask only saves work: it is
    // refreshed after a group that blended something instead of tracking "some pixel just finished" per
    // block; when it empties, every pixel of the tile is finished and both loops end (scalar exit).
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((live & (1 << k)) && !__any(tau[k] >= stop)) live &= ~(1 << k);
    }
    }
  }
  if (p.work_out) {   // what k_draw_bwd will walk: the largest contributor index of the tile and of its blocks
    int w = 0, wmax = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int mx = cont[k];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d, 64));
      w += mx;
      wmax = max(wmax, mx);
    }
    if (lane == 0) {
      p.work_out[tile] = w + 2 * wmax;
      if (p.walk_out) p.walk_out[tile] = wmax;
      walk_raise(p.walk_max, wmax);
      if (p.walk_max) walk_raise(p.walk_max + 1, n);      // ... and the longest list of the same render
    }
  }
  const size_t HW = (size_t)p.W * p.H;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = pxb[k & 1], py = pyb[k >> 1];
    if (px < p.W && py < p.H) {
      const size_t pix = (size_t)py * p.W + px;
      image[pix] = cr[k];
      image[HW + pix] = cg[k];
      image[2 * HW + pix] = cb[k];
      contrib[pix] = cont[k];
      final_tau[pix] = tau[k];
    }
  }
}

// ============================================================================
// draw backward: per-tile back-to-front gradients        (reference kernel.cu:809-950)
// ============================================================================
// gfx950 cross-half / cross-row swaps (v_permlane32_swap_b32, v_permlane16_swap_b32)
__device__ __forceinline__ void swap32(float& a, float& b) {  // a[32..63] <-> b[0..31]
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& a, float& b) {  // odd rows of a <-> even rows of b
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
// ---- transposing wave reduction ------------------------------------------------------------------
// 4 entries x 9 quantities = 36 per-lane partials have to become 36 wave totals.  Every step pairs two
// registers, sends half of each to the partner lanes and adds: one output register per input pair, so the
// register count halves with the lane span (36 -> 18 -> 9 across the 16-lane rows with
// v_permlane32_swap / v_permlane16_swap, then 9 -> 5 -> 3 -> 2 -> 1 inside the rows with DPP mirrors).
// 54 + 27 instructions instead of 36 x 6 DPP adds, and the nine totals of an entry land in nine
// different lanes of its row -- exactly where the one-instruction atomic wants them.
__device__ __forceinline__ float rows_of4(float e0, float e1, float e2, float e3) {
  swap32(e0, e1);
  const float s01 = e0 + e1;  // lanes 0-31: e0 halves, lanes 32-63: e1 halves
  swap32(e2, e3);
  const float s23 = e2 + e3;
  float a = s01, b = s23;
  swap16(a, b);               // rows of a: [e0, e2, e1, e3]; rows of b: the other halves
  return a + b;               // row r: 16 partial sums of entry {0,2,1,3}[r]
}
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// `hi` lanes reduce b, the others reduce a; partner lane through the mirror CTRL (a bijection between
// the two lane classes)
template <int CTRL>
__device__ __forceinline__ float merge2(float a, float b, bool hi) {
  const float own = hi ? b : a, other = hi ? a : b;
  return own + dpp_get<CTRL>(other);
}
// the nine row-wise totals of (q0..q8) in lanes {0, 8, 4, 12, 2, 10, 6, 14, odd} of every row
__device__ __forceinline__ float rows_to_lanes9(const float (&q)[9], int c16) {
  const bool h8 = (c16 & 8) != 0, h4 = (c16 & 4) != 0, h2 = (c16 & 2) != 0, h1 = (c16 & 1) != 0;
  constexpr int M8 = 0x140, M4 = 0x141, M2 = 0x4E, M1 = 0xB1;  // row_mirror, row_half_mirror, quad [2,3,0,1], [1,0,3,2]
  const float p01 = merge2<M8>(q[0], q[1], h8), p23 = merge2<M8>(q[2], q[3], h8);
  const float p45 = merge2<M8>(q[4], q[5], h8), p67 = merge2<M8>(q[6], q[7], h8);
  float s8 = q[8] + dpp_get<M8>(q[8]);
  const float a = merge2<M4>(p01, p23, h4), b = merge2<M4>(p45, p67, h4);
  s8 += dpp_get<M4>(s8);
  const float r = merge2<M2>(a, b, h2);
  s8 += dpp_get<M2>(s8);
  return merge2<M1>(r, s8, h1);
}

// ---- the in-row stage without selects ----------------------------------------------------------------------
// Measured on gfx950 (tools/ubench_calib.hip, cycles per wave instruction per SIMD): add / mul / fma 2.5 (full
// rate); DPP, v_cndmask, v_med3, v_min/max, v_cmp, v_readlane, v_mov_b64 4.3 (half rate); v_permlane{32,16}_swap,
// v_exp, v_rcp 8.4 (quarter rate); ds_swizzle 8.2 and ds_bpermute 24 (the LDS crossbar is shared by the four SIMDs
// of a CU: moving the cross-row exchanges there was measured 10 % SLOWER, so they stay v_permlane swaps).
// merge2 above costs two v_cndmask and a DPP add.  The first two levels split the row by lane bits 3 and 2 --
// exactly what DPP's bank mask addresses (a bank = four consecutive lanes of a row): one DPP add for everybody,
// one bank-masked DPP add for the lanes that reduce the second register; no select.
// out = a + a[mirror] everywhere, then b + b[mirror] on the banks of `bank_hi`
#define EGS_MERGE_BANK(out, a, b, ctrl, bank_hi)                                                              \
  do {                                                                                                        \
    asm("v_add_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(a));                  \
    asm("v_add_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:" bank_hi : "+v"(out) : "v"(b));            \
  } while (0)
// same result layout as rows_to_lanes9: totals in lanes {0, 8, 4, 12, 2, 10, 6, 14, odd} of every row
__device__ __forceinline__ float rows_to_lanes9_bank(const float (&q)[9], int c16) {
  const bool h2 = (c16 & 2) != 0, h1 = (c16 & 1) != 0;
  constexpr int M8 = 0x140, M4 = 0x141, M2 = 0x4E, M1 = 0xB1;
  float p01, p23, p45, p67, a, b;
  EGS_MERGE_BANK(p01, q[0], q[1], "row_mirror", "0xc");        // lanes 8..15 (banks 2, 3) reduce the second one
  EGS_MERGE_BANK(p23, q[2], q[3], "row_mirror", "0xc");
  EGS_MERGE_BANK(p45, q[4], q[5], "row_mirror", "0xc");
  EGS_MERGE_BANK(p67, q[6], q[7], "row_mirror", "0xc");
  float s8 = q[8] + dpp_get<M8>(q[8]);
  EGS_MERGE_BANK(a, p01, p23, "row_half_mirror", "0xa");       // lanes 4..7, 12..15 (banks 1, 3)
  EGS_MERGE_BANK(b, p45, p67, "row_half_mirror", "0xa");
  s8 += dpp_get<M4>(s8);
  const float r = merge2<M2>(a, b, h2);
  s8 += dpp_get<M2>(s8);
  return merge2<M1>(r, s8, h1);
}

// Per-tile back-to-front gradient pass.  One wave64 per 16x16 tile walked as four 8x8
// pixel blocks exactly like k_draw (same block cull, same exponent-domain skip test).
// Entries are visited in descending list order in groups of four.  Per entry each lane
// sums over its 4 pixels nine partials:
//   S0 = sum dL/dalpha' g                      -> dalpha          (B.5.1a)
//   S1..S3 = sum dL/dgamma_c alpha' tau        -> dcolor          (B.5b)
//   with w = dL/dalpha' alpha':  M1x = sum w dx, M1y = sum w dy,
//   M2xx = sum w dx dx, M2xy = sum w dx dy, M2yy = sum w dy dy    (B.5.2b / B.5.2c as moments:
//   du = -cinv (M1x, M1y), dcinv = -(M2xx/2, M2xy, M2yy/2), applied once per entry)
// The 9 partials are reduced across the wave 4 entries at a time (transposing reduction below) and nine
// lanes per entry issue the 9 atomics as one instruction: one atomic set per (tile, Gaussian).
// SEG: the launch runs over the forward pass's work items (items1) instead of tiles: DIRECT(tile) is the kernel
// as it always was; SPEC(tile, s) walks entries [s L, (s + 1) L) of a split tile only, and a pixel whose last
// contributor lies BEHIND the segment starts from the state the forward pass's COMPOSE item left for the segment's end
// -- the transmittance there and G, the colour of everything behind it (lq = dL/dgamma . G) -- where the unsplit kernel
// starts every pixel from (final_tau, 0) at its last contributor.
template <bool BOX, bool FLOOR, bool CLAMP, int RED, bool SEG = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_draw_bwd(DrawParams p, const int32_t* __restrict__ ranges,
                                                 const int32_t* __restrict__ gsid,
                                                 const float4* __restrict__ rec,
                                                 const float* __restrict__ final_tau,
                                                 const int32_t* __restrict__ contrib,
                                                 const float* __restrict__ dLdg,
                                                 float* __restrict__ gpack, SegArgs sg) {
  __shared__ float4 sA[64], sB[64], sC[64], sD[64];  // sD = {cinv.x, cinv.y, cinv.z, gsid}
  __shared__ float4 szero[3];                        // a line of zeros (see the accumulator reset below)
  constexpr bool ZLDS = (RED & 2) != 0, LAZY = (RED & 4) != 0;
  if (ZLDS && threadIdx.x < 3) szero[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t zaddr = (uint32_t)(uintptr_t)szero;   // LDS byte offset of the zero line
  int tile, seg_lo = 0, seg_hi = 0x7fffffff;   // SEG: the entries [seg_lo, seg_hi) of the tile's list are this wave's
  size_t seg_state = 0;
  bool seg_item = false;
  if constexpr (SEG) {
    if ((int)blockIdx.x >= min(sg.hdr[SH_ITEMS1], sg.item_cap)) return;
    const uint32_t item = (uint32_t)sg.items1[blockIdx.x];
    tile = (int)(item & SEG_TILE_MASK);
    if (tile >= p.T) return;
    if ((item >> 30) == (uint32_t)SEG_SPEC) {
      const int L = sg.hdr[SH_L], sidx = (int)((item >> 19) & SEG_SEG_MASK);
      seg_item = true;
      seg_lo = sidx * L; seg_hi = seg_lo + L;
      seg_state = ((size_t)(sg.seg_base[tile] + sidx)) * 256 + threadIdx.x;
    }
  } else {
    tile = xcd_tile(blockIdx.x, p);
    if (tile < 0) return;
  }
  const int r0 = ranges[2 * (size_t)tile], r1 = ranges[2 * (size_t)tile + 1];
  const int n = r1 - r0;
  if (n <= 0) return;
  if (SEG) seg_hi = min(seg_hi, n);
  const int lane = threadIdx.x;
  const int tx0 = (tile % p.gx) * EGS_TILE, ty0 = (tile / p.gx) * EGS_TILE;
  const int pxb[2] = {tx0 + (lane & 7), tx0 + (lane & 7) + 8};
  const int pyb[2] = {ty0 + (lane >> 3), ty0 + (lane >> 3) + 8};
  const float fpx[2] = {(float)pxb[0], (float)pxb[1]};
  const float fpy[2] = {(float)pyb[0], (float)pyb[1]};
  const size_t HW = (size_t)p.W * p.H;
  // lq = dL/dgamma . gamma_cur2last: the only combination of gamma_cur2last (kernel.cu:854,948)
  // the gradient needs, so the 3-vector recurrence q += a'(c - q) is carried as one scalar
  float tau[4], lr[4], lg[4], lb[4], lq[4];
  int cont[4];
  int bmax[4];  // wave-uniform: largest contrib of block k -> entries >= bmax[k] are inert for it
  int maxcont = 0;
  // (all twenty loads requested first, from clamped addresses: guarded and inside the loop below, every block's
  // five waited for their own round trip before the next block's were issued)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = pxb[k & 1], py = pyb[k >> 1];
    const size_t pix = (size_t)min(py, p.H - 1) * p.W + min(px, p.W - 1);
    tau[k] = final_tau[pix];
    cont[k] = contrib[pix];
    lr[k] = dLdg[pix]; lg[k] = dLdg[HW + pix]; lb[k] = dLdg[2 * HW + pix];
    lq[k] = 0.f;
  }
  float4 segE[4];
  if constexpr (SEG) {   // (requested with the loads above; a DIRECT item or the last segment never uses them)
#pragma unroll
    for (int k = 0; k < 4; ++k) segE[k] = seg_item ? sg.st4[seg_state + 64 * k] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = pxb[k & 1], py = pyb[k >> 1];
    if (!(px < p.W && py < p.H)) { tau[k] = 0.f; cont[k] = 0; lr[k] = 0.f; lg[k] = 0.f; lb[k] = 0.f; }
    if constexpr (SEG) {
      if (seg_item) {
        if (cont[k] > seg_hi) {          // contributors behind this segment: start from the state at its end
          tau[k] = segE[k].w;
          lq[k] = lr[k] * segE[k].x + lg[k] * segE[k].y + lb[k] * segE[k].z;
          cont[k] = seg_hi;
        } else if (cont[k] <= seg_lo) {  // the pixel never got this far
          cont[k] = 0;
        }
      }
    }
    int mx = cont[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d, 64));
    bmax[k] = __builtin_amdgcn_readfirstlane(min(mx, n));
    maxcont = max(maxcont, bmax[k]);
  }
  if (maxcont <= 0) return;
  // The loads above must be WAITED FOR here, not at their first use inside the loop: gfx9 counts stores and
  // atomics in the same in-order vmcnt as loads, so a wait the compiler places at the first use (inside the
  // hit body) would, from the second group on, also wait for the previous group's gradient atomics --
  // a round trip to L2 per group of four entries on the critical path of the wave.
#pragma unroll
  for (int k = 0; k < 4; ++k)
    asm volatile("" ::"v"(tau[k]), "v"(lr[k]), "v"(lg[k]), "v"(lb[k]), "v"(cont[k]));
  // where the transposing reduction leaves the nine totals inside a row of 16 lanes, and what each of
  // those lanes adds to the packed gradient record {dalpha, dcolor[3], du[2], dcinv[3]}
  const int c16 = lane & 15;
  int qoff = -1, kind = 0;
  float kscale = 1.f;
  if (c16 & 1) { if (c16 == 1) { qoff = 8; kscale = -0.5f; } }          // M2yy -> dcinv.z
  else if (c16 == 0) { qoff = 4; kind = 1; }                            // M1x  -> du.x
  else if (c16 == 2) { qoff = 5; kind = 2; }                            // M1y  -> du.y
  else if (c16 == 4) qoff = 0;                                          // dalpha
  else if (c16 == 6) qoff = 1;                                          // dcolor.r
  else if (c16 == 8) qoff = 2;                                          // dcolor.g
  else if (c16 == 10) qoff = 3;                                         // dcolor.b
  else if (c16 == 12) { qoff = 6; kscale = -0.5f; }                     // M2xx -> dcinv.x
  else { qoff = 7; kscale = -1.f; }                                     // M2xy -> dcinv.y  (lane 14)

  const int c_first = (maxcont - 1) >> 6;
  int gnext = (c_first * 64 + lane < n) ? gsid[r0 + c_first * 64 + lane] : 0;   // one chunk ahead, as in k_draw
  const int c_last = SEG ? (seg_lo >> 6) : 0;
  for (int c = c_first; c >= c_last; --c) {
    __syncthreads();
    const int idx = c * 64 + lane;
    int mymask = 0;  // reach mask of the entry THIS lane staged (lane j <-> entry c*64 + j)
    const int gm = gnext;
    const int g = p.masked ? (int)((uint32_t)gm & EGS_GSID_MASK) : gm;
    if (c > c_last) gnext = gsid[r0 + idx - 64];
    if (idx < n) {
      float4 A = rec[3 * (size_t)g], B = rec[3 * (size_t)g + 1];
      const float4 C = rec[3 * (size_t)g + 2];
      constexpr float INVQ = 1.f / EGS_NHL2E;
      // (cinv from the record as it is: an entry with a NaN conic hands NaN to du = -cinv M1, as kernel.cu:926-933 does)
      const float4 Dc = make_float4(A.z * INVQ, A.w * (0.5f * INVQ), B.x * INVQ, __int_as_float(g));
      const bool nanfix = p.nan_blend && nan_entry_fix(A, B);
      mymask = p.masked ? (int)((uint32_t)gm >> EGS_GSID_BITS) : reach_mask<BOX>(A, C, tx0, ty0);
      if (nanfix && !BOX && !p.masked) mymask = 0xF;
      sA[lane] = A;
      sB[lane] = B;
      sC[lane] = C;
      // cinv back out of the pre-scaled conic of the record (q = -0.5 log2(e) (cinv.x, 2 cinv.y, cinv.z)):
      // no second 12-B gather per patch (131 MB of sector traffic at P = 4.1 M)
      sD[lane] = Dc;
    }
    __syncthreads();
    // Which entries of this chunk can contribute at all?  Every lane answers for the entry it staged: its
    // reach mask minus the blocks no pixel of which ever got this far (entry index >= the block's largest
    // contrib, kernel.cu:899); a scalar bit scan then walks the reachable entries in descending list order.
    // Groups of four: each of the four accumulator slots takes entries until one of them HITS (a quarter
    // of the entries that reach a live block hit no pixel: they leave the slot zero and cost neither a
    // re-zeroing nor a share of a wave reduction).
    int rl = mymask;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (idx >= bmax[k]) rl &= ~(1 << k);
    unsigned long long todo = __ballot(rl != 0);
    while (todo != 0ull) {
      int je[4] = {-1, -1, -1, -1};   // chunk-local entry index held by slot e
      float acc[4][9];
      bool any = false;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ZLDS) {
          // The nine zeros come out of LDS: broadcast reads of a zero line cost the VALU nothing (nine v_mov_b32 or
          // five v_mov_b64 are 21 issue cycles per slot in a kernel that is VALU-issue bound; the LDS pipe idles).
          // Inline asm, because the compiler would hoist a plain load and hand out register copies again.  The
          // wait is part of the statement: the compiler does not see these loads in its lgkmcnt bookkeeping and
          // may copy the results anywhere afterwards.  (The wave parks for one LDS latency; its four neighbours
          // on the SIMD issue meanwhile.)
          typedef float f4v __attribute__((ext_vector_type(4)));
          f4v z0, z1;
          float z2;
          asm volatile("ds_read_b128 %0, %3\n ds_read_b128 %1, %3 offset:16\n ds_read_b32 %2, %3 offset:32\n"
                       " s_waitcnt lgkmcnt(0)"
                       : "=v"(z0), "=v"(z1), "=v"(z2) : "v"(zaddr));
          acc[e][0] = z0.x; acc[e][1] = z0.y; acc[e][2] = z0.z; acc[e][3] = z0.w;
          acc[e][4] = z1.x; acc[e][5] = z1.y; acc[e][6] = z1.z; acc[e][7] = z1.w;
          acc[e][8] = z2;
        } else {  // nine zeros from five 64-bit moves (v_mov_b64 on gfx940+)
#pragma unroll
          for (int q = 0; q < 8; q += 2) {
            unsigned long long z = 0ull;
            asm volatile("" : "+v"(z));   // materialise the pair in VGPRs, keep it from being split into two constants
            acc[e][q] = __uint_as_float((unsigned)z);
            acc[e][q + 1] = __uint_as_float((unsigned)(z >> 32));
          }
          acc[e][8] = 0.f;
        }
        while (todo != 0ull) {
        const int j = 63 - __clzll((long long)todo);
        todo &= ~(1ull << j);
        bool any_e = false;
        const int i = c * 64 + j;  // forward index of this entry in the tile list
        const int reach = __builtin_amdgcn_readlane(rl, j);  // lane j's register: no LDS round trip
        const float4 A = sA[j], B = sB[j], C = sC[j];
        bool inx[2] = {true, true}, iny[2] = {true, true};
        if (BOX) {
          const uint32_t bx = __float_as_uint(C.y), by = __float_as_uint(C.z);
          const int x0 = bx & 0xFFFF, x1 = bx >> 16, y0 = by & 0xFFFF, y1 = by >> 16;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            inx[b] = (pxb[b] >= x0) && (pxb[b] < x1);
            iny[b] = (pyb[b] >= y0) && (pyb[b] < y1);
          }
        }
        // LAZY: the exponent from scratch per evaluated block (7 full-rate instructions) instead of the separable
        // form (14 per entry up front + 2 per block): most entries reach one or two of the four blocks.
        // (Measured and dropped: skipping the floor / clamp v_med3 for entries with a positive-definite conic and
        // alpha <= 0.989 behind a wave-uniform flag -- the two scalar branches cost more than the two half-rate
        // instructions they save: +1.5 %.)
        float dx[2], dy[2], cxx[2], cxy[2], cyy[2];
        if (!LAZY) {
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            dx[b] = A.x - fpx[b];
            cxx[b] = A.z * dx[b] * dx[b];
            cxy[b] = A.w * dx[b];
            dy[b] = A.y - fpy[b];
            cyy[b] = B.x * dy[b] * dy[b];
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int bx = k & 1, by = k >> 1;
          if (!(reach & (1 << k))) continue;  // scalar branch: block culled or past its last contributor
          float pw;
          if (LAZY) {
            dx[bx] = A.x - fpx[bx];
            dy[by] = A.y - fpy[by];
            float t = A.z * dx[bx];
            t = fmaf(A.w, dy[by], t);
            pw = t * dx[bx];
            pw = fmaf(B.x * dy[by], dy[by], pw);
          } else {
            pw = cxx[bx] + cyy[by] + cxy[bx] * dy[by];
          }
          bool hit = (i < cont[k]) && (pw >= C.w);  // kernel.cu:899,913
          if (BOX) hit = hit && inx[bx] && iny[by];
          if (hit) {
            const float g = __builtin_amdgcn_exp2f(FLOOR ? min_hi(pw, 0.f) : pw);
            float ap = B.y * g;
            if (CLAMP) ap = min_hi(ap, 0.99f);
            const float tk = tau[k] * __builtin_amdgcn_rcpf(1.f - ap);  // undo F.5.2
            tau[k] = tk;
            const float dq = (lr[k] * B.z + lg[k] * B.w + lb[k] * C.x) - lq[k];  // dL/dgamma . (color - gamma_cur2last)
            const float dl_dap = tk * dq;  // B.5a
            acc[e][0] += dl_dap * g;  // dalpha'/dalpha = g, also where the clamp binds (kernel.cu:921)
            const float wgt = ap * tk;
            acc[e][1] += lr[k] * wgt; acc[e][2] += lg[k] * wgt; acc[e][3] += lb[k] * wgt;
            const float w = dl_dap * ap;
            const float wx = w * dx[bx], wy = w * dy[by];
            acc[e][4] += wx; acc[e][5] += wy;
            acc[e][6] += wx * dx[bx]; acc[e][7] += wx * dy[by]; acc[e][8] += wy * dy[by];
            lq[k] += ap * dq;  // gamma_cur2last <- a' color + (1 - a') gamma_cur2last, dotted with dL/dgamma
            any_e = true;
          }
        }
        if (__any(any_e)) {  // wave-uniform: the entry contributed, the slot is taken
          je[e] = j;
          any = true;
          break;
        }
        }  // next reachable entry into the same (still zero) slot
      }
      if (any) {  // wave-uniform
        // quantity order chosen so that the two first moments meet in one quad (lanes 0 and 2):
        //   lane 0: M1x  2: M1y  4: dalpha  6,8,10: dcolor  12: M2xx  14: M2xy  odd: M2yy
        float rows[9];
        constexpr int ORDER[9] = {4, 2, 0, 6, 5, 3, 1, 7, 8};   // acc index feeding leaf q0..q8
#pragma unroll
        for (int q = 0; q < 9; ++q)
          rows[q] = rows_of4(acc[0][ORDER[q]], acc[1][ORDER[q]], acc[2][ORDER[q]], acc[3][ORDER[q]]);
        const float v = (RED & 1) == 0 ? rows_to_lanes9(rows, c16) : rows_to_lanes9_bank(rows, c16);
        // row r of the wave holds the totals of slot e = {0,2,1,3}[r]
        const int row = lane >> 4;
        const int e = ((row & 1) << 1) | (row >> 1);
        const int j = (e == 0) ? je[0] : (e == 1) ? je[1] : (e == 2) ? je[2] : je[3];
        // an empty slot (the chunk ran out of entries) holds zeros and no entry: it must not touch memory
        const bool rowact = j >= 0;
        const float4 D = sD[j & 63];
        // B.5.2b / B.5.2c from the moments: du = -cinv (M1x, M1y) needs both first moments -> the partner
        // comes from the other lane of the pair (quad_perm [2,3,0,1]); dcinv = -(M2xx/2, M2xy, M2yy/2).
        // The 9 atomics of an entry are ONE instruction on ONE 48-byte gradient record.
        const float nb = dpp_get<0x4E>(v);
        const float c_own = (kind == 1) ? -D.x : ((kind == 2) ? -D.z : kscale);
        float val = v * c_own;
        if (kind != 0) val = fmaf(nb, -D.y, val);
        if (rowact && qoff >= 0 && val != 0.f)
          unsafeAtomicAdd(gpack + 12 * (size_t)__float_as_int(D.w) + qoff, val);
      }
    }
  }
}

// packed [N][12] gradient records -> the four output tensors of splatB
__global__ __launch_bounds__(256) void k_unpack_grads(int n, const float4* __restrict__ gpack,
                                                      float* __restrict__ dus, float* __restrict__ dcinv,
                                                      float* __restrict__ dalpha, float* __restrict__ dcolor) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 a = gpack[3 * (size_t)i], b = gpack[3 * (size_t)i + 1], c = gpack[3 * (size_t)i + 2];
  dalpha[i] = a.x;
  dcolor[3 * (size_t)i] = a.y; dcolor[3 * (size_t)i + 1] = a.z; dcolor[3 * (size_t)i + 2] = a.w;
  dus[2 * (size_t)i] = b.x; dus[2 * (size_t)i + 1] = b.y;
  dcinv[3 * (size_t)i] = b.z; dcinv[3 * (size_t)i + 1] = b.w; dcinv[3 * (size_t)i + 2] = c.x;
}

// ============================================================================
// host side
// ============================================================================
DrawParams make_draw_params(int W, int H, const EgsPolicy* pol, bool backward) {
  DrawParams p;
  p.W = W; p.H = H;
  p.gx = div_up(W, EGS_TILE);
  p.gy = div_up(H, EGS_TILE);
  p.T = p.gx * p.gy;
  // tile -> workgroup map, chosen by measurement (same-box A/B, 1 M Gaussians at 1080p): the forward kernel
  // is 2 % faster with tile rows interleaved over the XCDs (223 vs 227 us), the backward kernel 2.5 % faster
  // with the plain map (580 vs 595 us).  EGS_TILE_MAP=0|1|2 overrides both (tuning knob).
  static const int forced = [] {
    const char* e = getenv("EGS_TILE_MAP");
    return e ? atoi(e) : -1;
  }();
  p.map_mode = forced >= 0 ? forced : (backward ? 0 : 2);
  p.order = nullptr;
  p.ngrid = 0;
  p.zero_buf = nullptr;
  p.zero_n4 = 0;
  p.zero_per = 0;
  p.work_out = nullptr;
  p.walk_out = nullptr;
  p.walk_max = nullptr;
  p.masked = 0;
  p.alpha_skip = pol->alpha_skip; p.tau_stop = pol->tau_stop;
  p.lskip = pol->alpha_skip > 0.f ? log2f(pol->alpha_skip) : -INFINITY;
  p.maha_floor = pol->maha_floor; p.alpha_clamp = pol->alpha_clamp;
  p.nan_blend = pol->nan_maha == 0 && pol->maha_floor;
  return p;
}

// Longest-list-first dispatch (k_tile_order) for one of the draw kernels: which = 0 forward, 1 backward.
// Mode by measurement (same-box A/B at 1 M / 1080p); EGS_TILE_ORDER_F / _B = 0..4 and
// EGS_TILE_SERP override (tuning knobs).
int tile_order_mode(int which) {
  static const int mode[2] = {
      [] { const char* e = getenv("EGS_TILE_ORDER_F"); return e ? atoi(e) : EGS_TILE_ORDER_F_DEFAULT; }(),
      [] { const char* e = getenv("EGS_TILE_ORDER_B"); return e ? atoi(e) : EGS_TILE_ORDER_B_DEFAULT; }()};
  return mode[which];
}
int tile_order_enqueue(DrawParams& p, int which, int32_t* buf, size_t buf_len, const int32_t* ranges, hipStream_t s,
                       const int32_t* work, const int32_t* walk, uint32_t* hint) {
  const int mode = tile_order_mode(which);
  if (mode <= 0 || !buf) return 0;
  const bool per_xcd = mode >= 3;
  const int ngrid = per_xcd ? tile_order_len(p.gx, p.gy) : p.T;
  if ((size_t)ngrid > buf_len || p.T > TILE_ORDER_MAX_T) return 0;   // (larger images keep the plain map)
  static const int serp = [] { const char* e = getenv("EGS_TILE_SERP"); return e ? atoi(e) : 0; }();
  const int period = serp > 0 ? serp : (per_xcd ? 128 : 1024);   // SIMDs per XCD / per chip
  EGS_LAUNCH("k_tile_order", k_tile_order, dim3(1), dim3(1024), s, ranges, work, p.T, p.gx, mode, period, buf, ngrid,
             walk, hint);
  EGS_LAUNCH_OK();
  p.order = buf;
  p.ngrid = ngrid;
  return 0;
}

int tile_work_from_contrib(const DrawParams& p, const int32_t* contrib, int32_t* work, int32_t* walk, hipStream_t s) {
  if (work) EGS_LAUNCH("k_tile_work", k_tile_work, dim3(p.T), dim3(64), s, p.W, p.H, p.gx, contrib, work, walk);
  else EGS_LAUNCH("k_tile_walk", k_tile_walk, dim3(p.T), dim3(64), s, p.W, p.H, p.gx, contrib, walk);
  EGS_LAUNCH_OK();
  return 0;
}

// policy -> template instance (compile-time footprint / floor / clamp)
int launch_draw(const DrawParams& dp, const EgsPolicy* pol, int32_t* ranges, const int32_t* gsid, const float4* rec,
                float* image, int32_t* contrib, float* final_tau, hipStream_t s) {
#define EGS_DRAW(BOX, FLOOR, CLAMP)                                                                                \
  do {                                                                                                             \
    if (pol->alpha_skip > 0.f)                                                                                     \
      EGS_LAUNCH_LDS("k_draw", (k_draw<BOX, FLOOR, CLAMP, true>), dim3(draw_grid(dp)), dim3(64), draw_lds_pad(0), s, \
                     dp, ranges, gsid, rec, image, contrib, final_tau);                                            \
    else                                                                                                           \
      EGS_LAUNCH_LDS("k_draw", (k_draw<BOX, FLOOR, CLAMP, false>), dim3(draw_grid(dp)), dim3(64), draw_lds_pad(0), s, \
                     dp, ranges, gsid, rec, image, contrib, final_tau);                                            \
  } while (0)
  const int sel = (pol->footprint == 1 ? 4 : 0) | (pol->maha_floor ? 2 : 0) | (pol->alpha_clamp ? 1 : 0);
  switch (sel) {
    case 0: EGS_DRAW(false, false, false); break;
    case 1: EGS_DRAW(false, false, true); break;
    case 2: EGS_DRAW(false, true, false); break;
    case 3: EGS_DRAW(false, true, true); break;
    case 4: EGS_DRAW(true, false, false); break;
    case 5: EGS_DRAW(true, false, true); break;
    case 6: EGS_DRAW(true, true, false); break;
    default: EGS_DRAW(true, true, true); break;
  }
#undef EGS_DRAW
  EGS_LAUNCH_OK();
  return 0;
}

int launch_draw_bwd(const DrawParams& dp, const EgsPolicy* pol, const int32_t* ranges, const int32_t* gsid,
                    const float4* rec, const float* final_tau, const int32_t* contrib, const float* dLdg, float* gpack,
                    hipStream_t s) {
  // variants of the backward kernel (bit 0: in-row merges of the wave reduction with bank-masked DPP adds instead
  // of selects; bit 1: accumulator zeros loaded from LDS instead of moved; bit 2: exponent per evaluated block);
  // EGS_DRAWB_RED = 0 | 3 | 7 overrides
  static const int red = [] { const char* e = getenv("EGS_DRAWB_RED"); return e ? atoi(e) : EGS_DRAWB_RED_DEFAULT; }();
  const SegArgs nosg = {};
#define EGS_DRAWB(BOX, FLOOR, CLAMP)                                                                               \
  do {                                                                                                             \
    if (red == 0)                                                                                                  \
      EGS_LAUNCH_LDS("k_draw_bwd", (k_draw_bwd<BOX, FLOOR, CLAMP, 0>), dim3(draw_grid(dp)), dim3(64), draw_lds_pad(1), \
                     s, dp, ranges, gsid, rec, final_tau, contrib, dLdg, gpack, nosg);                             \
    else if (red == 3)                                                                                             \
      EGS_LAUNCH_LDS("k_draw_bwd", (k_draw_bwd<BOX, FLOOR, CLAMP, 3>), dim3(draw_grid(dp)), dim3(64), draw_lds_pad(1), \
                     s, dp, ranges, gsid, rec, final_tau, contrib, dLdg, gpack, nosg);                             \
    else                                                                                                           \
      EGS_LAUNCH_LDS("k_draw_bwd", (k_draw_bwd<BOX, FLOOR, CLAMP, 7>), dim3(draw_grid(dp)), dim3(64), draw_lds_pad(1), \
                     s, dp, ranges, gsid, rec, final_tau, contrib, dLdg, gpack, nosg);                             \
  } while (0)
  const int sel = (pol->footprint == 1 ? 4 : 0) | (pol->maha_floor ? 2 : 0) | (pol->alpha_clamp ? 1 : 0);
  switch (sel) {
    case 0: EGS_DRAWB(false, false, false); break;
    case 1: EGS_DRAWB(false, false, true); break;
    case 2: EGS_DRAWB(false, true, false); break;
    case 3: EGS_DRAWB(false, true, true); break;
    case 4: EGS_DRAWB(true, false, false); break;
    case 5: EGS_DRAWB(true, false, true); break;
    case 6: EGS_DRAWB(true, true, false); break;
    default: EGS_DRAWB(true, true, true); break;
  }
#undef EGS_DRAWB
  EGS_LAUNCH_OK();
  return 0;
}

int launch_draw_bwd_seg(const DrawParams& dp, const EgsPolicy* pol, const int32_t* ranges, const int32_t* gsid,
                        const float4* rec, const float* final_tau, const int32_t* contrib, const float* dLdg,
                        float* gpack, const SegArgs& sga, int grid, hipStream_t s) {
#define EGS_DRAWBS(FLOOR, CLAMP)                                                                                   \
  EGS_LAUNCH("k_draw_bwd_seg", (k_draw_bwd<false, FLOOR, CLAMP, 7, true>), dim3(grid), dim3(64), s, dp, ranges, gsid, \
             rec, final_tau, contrib, dLdg, gpack, sga)
  switch ((pol->maha_floor ? 2 : 0) | (pol->alpha_clamp ? 1 : 0)) {
    case 0: EGS_DRAWBS(false, false); break;
    case 1: EGS_DRAWBS(false, true); break;
    case 2: EGS_DRAWBS(true, false); break;
    default: EGS_DRAWBS(true, true); break;
  }
#undef EGS_DRAWBS
  EGS_LAUNCH_OK();
  return 0;
}

int unpack_grads(int n, const float* gpack, float* dus, float* dcinv, float* dalpha, float* dcolor, hipStream_t s) {
  EGS_LAUNCH("k_unpack_grads", k_unpack_grads, dim3(div_up(n, 256)), dim3(256), s, n, (const float4*)gpack, dus, dcinv,
             dalpha, dcolor);
  EGS_LAUNCH_OK();
  return 0;
}

}  // namespace egs

// a caller-held tile_order buffer: [dispatch order of the forward draw | per-tile work it measured | walk (T ints each)]
extern "C" size_t egs_tile_order_len(int width, int height) {
  const int gx = egs::div_up(width, EGS_TILE), gy = egs::div_up(height, EGS_TILE);
  return (size_t)egs::tile_order_len(gx, gy) + 2 * (size_t)gx * gy;
}
