// Long tile lists split over several waves: the plan kernel, the three forward launches over work items
// (k_draw_seg<ROLE>), the segment workspace and its configuration.  The backward kernel over the same work items is
// k_draw_bwd<.., SEG = true> in egs_draw.hip.  Reference: one 256-thread workgroup per tile, gsplatcu/kernel.cu:152-271.
#include "egs_draw_device.h"

#include <stdlib.h>

#include <algorithm>
#include <atomic>

namespace egs {

// ============================================================================
// long lists: a tile's list split over several waves
// ============================================================================
// k_draw / k_draw_bwd spend ONE wave64 on a tile, and a lone wave walks about five entries per microsecond forward, 2.5
// backward: both kernels end when the longest walk ends.  On the iid scene (lists <= 830) that is the throughput time;
// on a heavy-tailed scene right after reset_alpha (every opacity 0.01: nothing saturates, scene.skewed_scene) one tile
// walks 8 325 entries and the two kernels take 1.6 + 3.4 ms for 0.25 + 0.65 ms of work (profiles/r5_skewed_baseline.json).
// The reference spends 256 threads on a tile (kernel.cu:152-271, launched (16, 16) at gausplat.cu:94) -- one pixel per
// thread, every thread walks the whole list: that splits the PIXELS, which buys at most 2.5x here (the block masks
// already skip the blocks an entry cannot reach).  This splits the LIST, in segments of L entries (L a multiple of the
// 64-entry chunk), which scales with the list:
//
//   front-to-back blending is associative on (colour, tau) pairs:  (C1, t1) o (C2, t2) = (C1 + t1 C2, t1 t2),
//
// so a segment can be blended from tau = 1 ("local frame") by its own wave and composed afterwards.  What is NOT
// associative is the early stop (a pixel is finished once tau < tau_stop, kernel.cu:256-260): it depends on the
// transmittance in front of the segment.  A segment wave therefore stops a pixel only when its LOCAL tau falls below
// tau_stop (the true tau is smaller still: conservative), and the composing wave -- which knows the true transmittance
// T in front of every segment -- re-walks a segment for exactly the pixels that finish inside it (T tau_local <
// tau_stop), with the per-pixel threshold tau_stop / T in the local frame.  Every pixel finishes once, so this costs at
// most one extra segment walk per segment that holds a finishing pixel, restricted to the 8x8 blocks of those pixels.
//
// Work items (one wave64 each; k_seg_plan writes them longest first):
//   DIRECT(tile)          a tile of at most `split_min` entries: exactly k_draw
//   SPEC(tile, s)         segment s of a split tile, blended from tau = 1 into the tile's state slot s
//   COMPOSE(tile, nspec)  composes the tile's first nspec segments (re-walking where a pixel finishes), CONTINUES
//                         sequentially from there while a pixel is still alive -- segment by segment in the local
//                         frame, each leaving its state -- and writes the tile's pixels; then turns the slots into what
//                         the BACKWARD pass needs at the end of each segment: the transmittance there and the colour
//                         of everything behind it, G_s = C_(s+1) + tau_(s+1) G_(s+1) (no cancellation, no division).
// With those, the backward pass has no sequential dependence left at all: k_draw_bwd<SEG> walks segment s of a tile
// from (T_s, dL/dgamma . G_s) exactly as the unsplit kernel walks a tile from (final_tau, 0), one wave per segment.
// SPEC items need a prediction of how far the tile will be walked -- the walk length this camera's previous render
// measured (a trainer meets every view again; without one the COMPOSE item does the whole tile, exactly, and only the
// backward pass is split).  Nothing depends on the prediction but the balance.
// The process-wide default (segment length, shortest split list), one atomic word: a render reads it ONCE, when its
// forward pass plans (the plan kernel writes L into the workspace header, every later launch -- the backward pass
// included -- takes it from there and sizes its grid by the workspace), so a change between two calls cannot tear a render.
static std::atomic<uint64_t> g_seg_cfg{((uint64_t)1024 << 32) | 256u};
static void seg_config_env() {
  static const bool once = [] {
    const char* a = getenv("EGS_SEG_L");
    const char* b = getenv("EGS_SEG_MIN");
    int L = 256, mn = 1024;
    if (a && atoi(a) >= 64) { L = 64; while (2 * L <= atoi(a) && L < 65536) L *= 2; }
    if (b && atoi(b) > 0) mn = atoi(b);
    if (mn < L) mn = L;
    g_seg_cfg.store(((uint64_t)(uint32_t)mn << 32) | (uint32_t)L);
    return true;
  }();
  (void)once;
}
SegConfig seg_config() {
  seg_config_env();
  const uint64_t v = g_seg_cfg.load();
  return SegConfig{(int)(uint32_t)v, (int)(v >> 32)};
}
int64_t seg_item_bound(int T, int64_t patches, const SegConfig& c) {
  return (int64_t)T + patches / c.L + patches / c.split_min + 2;
}
static size_t seg_fixed_words(int T) { return (size_t)SEG_HDR + 48 + SEG_PLAN_BINS + 5 * (size_t)align_up((size_t)T, 64); }
size_t seg_ws_bytes_for(int64_t slots, int T) {
  return 4 * (seg_fixed_words(T) + ((size_t)T + (size_t)slots + 64)) + (size_t)slots * SEG_SLOT_FLOATS * 4 + 1024;
}
bool seg_carve(void* ws, size_t bytes, int T, SegArgs* a) {
  if (!ws || bytes < seg_ws_bytes_for(16, T)) return false;
  const size_t per_slot = SEG_SLOT_FLOATS * 4 + 4;
  const int64_t slots = (int64_t)((bytes - seg_ws_bytes_for(0, T)) / per_slot);
  if (slots < 16) return false;
  const size_t Tp = align_up((size_t)T, 64);
  int32_t* w = (int32_t*)ws;
  a->hdr = w; w += SEG_HDR + 48 + SEG_PLAN_BINS;   // header, spare words, the plan's global bins
  a->seg_base = w; w += Tp;
  a->walk = w; w += Tp;
  a->items3 = w; w += Tp;
  a->tmp = w; w += Tp;
  a->tmp2 = w; w += Tp;
  a->item_cap = (int)std::min<int64_t>((int64_t)T + slots, (int64_t)1 << 20);
  a->slot_cap = (int)std::min<int64_t>(slots, (int64_t)a->item_cap - T);
  a->items1 = w; w += (size_t)T + (size_t)slots + 64;
  a->st4 = (float4*)(((uintptr_t)w + 255) & ~(uintptr_t)255);
  a->st1 = (float*)(a->st4 + (size_t)a->slot_cap * 256);
  a->st2 = a->st1 + (size_t)a->slot_cap * 256;
  a->hist_walk = nullptr;
  a->rebuild = 0;
  a->walk_max = nullptr;
  return (char*)(a->st2 + (size_t)a->slot_cap * 256) <= (char*)ws + bytes;
}

// Planning a render: which tiles are split (state slots are handed out here), the work items, longest first (counting
// sort on the estimated walk, as k_tile_order), and the list statistics the host steers by:
//   ranges (+ hist: the walks this camera's previous render measured) -> seg_base, items1, items3, hdr
// Round 5 did this in ONE workgroup: 3 900 instructions per thread on one CU, 27-32 us whatever was tried inside it
// (LAB r5 note 4).  Round 6: two launches of T / 256 workgroups around GLOBAL bins -- k_seg_plan_count hands out the
// state slots and the compose positions (one returning atomic per wave: any disjoint ranges will do, the order is
// irrelevant) and ranks every tile's items inside its bin (one returning atomic per tile); k_seg_plan_place scans the
// 4096 bins in every workgroup (16 KB out of L2) and writes each tile's run of items itself.  The bins and counters
// (SEG_PLAN_WORDS words behind the header) are zeroed on the side by the kernel in front (k_tile_ranges).
// The BACKWARD launch runs over the same items1 (a second plan from this render's walks cost 31 us for a marginally
// better order): a segment the pixels never reached returns after its first loads.
// Segment 0 of a split tile is ALWAYS a SPEC item: it starts from tau = 1 like the unsplit walk, so it is exact and
// never wasted; further SPEC items follow the prediction (walk + a quarter), the COMPOSE item walks on where they end.
// (L is a power of two: a segment index is a shift -- an integer division is ~40 instructions on this part)
__device__ __forceinline__ int seg_nspec(const int32_t* __restrict__ hist, int h, int n, int nseg, int L, int Ls,
                                         int speculate) {
  // no walk on record for this camera: segment 0 only -- or, when the host knows the scene's tiles to be walked to
  // (nearly) their ends (EGS_DRAW_SEG_SPECULATE: nothing saturates, e.g. right after reset_alpha), the whole list
  if (!hist) return speculate ? nseg : 1;
  const int w = min(max(h, 0), n);
  // a walk on record that is far shorter than the list while the scene's recent renders walk most of theirs (`speculate`:
  // the host's hint words): a STALE record -- the render right after reset_alpha (gsmodel.py:320-324) meets the walks of
  // the opaque scene -- and the whole list is speculated, as at first sight
  if (speculate && 4 * w < n) return nseg;
  return max(1, min(nseg, (w + (w >> 2) + L) >> Ls));
}
// longest first: one bin per entry up to 3072, one per 32 beyond (to ~36 k).  (One bin per 8 entries put the 7 000 DIRECT
// tiles of a 1080p render on ~130 addresses: several hundred same-address atomics each, 12 ns apiece.)
__device__ __forceinline__ int seg_plan_bin(int est) {
  const int e = max(est, 0);
  return SEG_PLAN_BINS - 1 - (e < 3072 ? e : 3072 + min((e - 3072) >> 5, SEG_PLAN_BINS - 3073));
}
// the SPEC items of all split tiles are equal work: spread them over a few bins (they would all meet in one)
__device__ __forceinline__ int seg_plan_jitter(int t) { return (int)(((uint32_t)t * 2654435761u) >> 25) - 64; }

__global__ __launch_bounds__(256) void k_seg_plan_count(int T, const int32_t* __restrict__ ranges,
                                                        const int32_t* __restrict__ hist, int L, int split_min,
                                                        SegArgs a, int speculate) {
  const int t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const bool valid = t < T;
  const int tc = min(t, T - 1);
  int2 rr = reinterpret_cast<const int2*>(ranges)[tc];
  const int hh = hist ? hist[tc] : 0;
  if (a.rebuild) rr.y = rr.x + min(max(rr.y - rr.x, 0), max(hh, 0));   // a tile's list ends at its (given) walk
  const int Ls = 31 - __clz(L);
  const int n = valid ? max(rr.y - rr.x, 0) : 0, nseg = (n + L - 1) >> Ls;
  const bool split = valid && n > split_min && nseg <= (int)SEG_SEG_MASK;
  // state slots and compose positions of the wave's split tiles: ONE wave scan over (slots << 10 | tiles), one returning
  // atomic per wave on each of the two global counters
  const uint32_t both = split ? (((uint32_t)nseg << 10) | 1u) : 0u;
  const uint32_t inc = wave_inclusive_scan(both);
  int32_t* plan = a.hdr + SEG_HDR + 48;        // [SEG_PLAN_BINS] bins
  uint32_t wb = 0u, w3 = 0u;
  if (lane == 63 && inc) {      // (both counters in ONE 64-bit atomic: SH_P_SLOTS is 8-byte aligned, SH_P_N3 follows it)
    const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(&a.hdr[SH_P_SLOTS]),
                                             ((unsigned long long)(inc & 1023u) << 32) | (unsigned long long)(inc >> 10));
    wb = (uint32_t)old; w3 = (uint32_t)(old >> 32);
  }
  wb = (uint32_t)__builtin_amdgcn_readlane((int)wb, 63);
  w3 = (uint32_t)__builtin_amdgcn_readlane((int)w3, 63);
  const uint32_t sb = wb + ((inc - both) >> 10), s3 = w3 + ((inc - both) & 1023u);
  int mx = n, mw = (hist && valid) ? hh : 0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { mx = max(mx, __shfl_xor(mx, d, 64)); mw = max(mw, __shfl_xor(mw, d, 64)); }
  if (lane == 0) { if (mx > 0) atomicMax(&a.hdr[SH_P_MAXLEN], mx); if (mw > 0) atomicMax(&a.hdr[SH_P_MAXWALK], mw); }
  int base = -1;
  if (split) {
    // (a workspace of egs_seg_ws_bytes cannot run out of slots; if a caller's does, the tile stays unsplit)
    if ((int)sb + nseg <= a.slot_cap) {
      base = (int)sb;
      a.items3[s3] = (int32_t)((uint32_t)t | ((uint32_t)seg_nspec(hist, hh, n, nseg, L, Ls, speculate) << 19) |
                               ((uint32_t)SEG_COMPOSE << 30));
    } else {
      a.items3[s3] = t;      // (no COMPOSE kind: the per-tile launches skip it)
    }
  }
  if (valid) {
    a.seg_base[t] = base;
    int cnt = 1, est = n;
    if (base >= 0) { cnt = seg_nspec(hist, hh, n, nseg, L, Ls, speculate); est = L + seg_plan_jitter(t); }
    else if (hist) est = min(max(hh, 0), n);
    const int b = seg_plan_bin(est);
    a.tmp[t] = (int32_t)(((uint32_t)b << 20) | (uint32_t)atomicAdd(&plan[b], cnt));     // (bin, rank inside the bin)
    a.tmp2[t] = base >= 0 ? cnt : -1;
  }
}

__global__ __launch_bounds__(256) void k_seg_plan_place(int T, const int32_t* __restrict__ hist, int L, int split_min,
                                                        SegArgs a, uint32_t* __restrict__ hint_host) {
  __shared__ uint32_t start[SEG_PLAN_BINS];
  __shared__ uint32_t wsum[4];
  const int tid = threadIdx.x;
  const int32_t* plan = a.hdr + SEG_HDR + 48;
  {  // exclusive scan of the bins, in every workgroup: thread t owns bins [16 t, 16 t + 16)
    constexpr int PER = SEG_PLAN_BINS / 256;
    uint32_t v[PER], sum = 0u;
    const uint4* p4 = reinterpret_cast<const uint4*>(plan) + tid * (PER / 4);
#pragma unroll
    for (int k = 0; k < PER / 4; ++k) {
      const uint4 q = p4[k];
      v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
      sum += q.x + q.y + q.z + q.w;
    }
    const uint32_t inc = wave_inclusive_scan(sum);
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t ex = inc - sum;
    for (int w = 0; w < (tid >> 6); ++w) ex += wsum[w];
#pragma unroll
    for (int k = 0; k < PER; ++k) { start[PER * tid + k] = ex; ex += v[k]; }
  }
  __syncthreads();
  const int total = (int)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
  const int ntot = min(total, a.item_cap);
  const int t = blockIdx.x * 256 + tid;
  if (t < T) {
    const uint32_t pk = (uint32_t)a.tmp[t];
    const int cnt = a.tmp2[t];
    const int slot = (int)(start[pk >> 20] + (pk & 0xFFFFFu));
    // DIRECT: the bare tile index; split: its SPEC items, segment 0 first
    if (cnt < 0) { if (slot < ntot) a.items1[slot] = t; }
    else {
      const int32_t head = (int32_t)((uint32_t)t | ((uint32_t)SEG_SPEC << 30));
      for (int j = 0; j < cnt && slot + j < ntot; ++j) a.items1[slot + j] = head + (j << 19);
    }
  }
  if (blockIdx.x == 0 && tid == 0) {
    a.hdr[SH_ITEMS1] = ntot; a.hdr[SH_ITEMS3] = a.hdr[SH_P_N3]; a.hdr[SH_SLOTS] = a.hdr[SH_P_SLOTS];
    a.hdr[SH_MAXLEN] = a.hdr[SH_P_MAXLEN]; a.hdr[SH_SPLIT] = a.hdr[SH_P_N3]; a.hdr[SH_L] = L; a.hdr[SH_MIN] = split_min;
    a.hdr[SH_MAXWALK] = 0;
    // page-locked words the host peeks at before a LATER render
    // (a rebuild -- the public splatB -- is handed this render's own walks: both words at once, a consistent pair; a
    // forward render's pair comes from the range kernel of the next render on the stream, walk_raise)
    if (hint_host && a.rebuild) { hint_host[0] = (uint32_t)a.hdr[SH_P_MAXLEN]; hint_host[1] = (uint32_t)a.hdr[SH_P_MAXWALK]; }
  }
}

// The forward kernels over work items (see above), tile-footprint policies with a skip threshold only (the pixel-box
// policy of forward_cpu.py has no early stop to speak of and is not a training path).  Three launches, ROLE:
//   0  items1: DIRECT tiles (== k_draw) and SPEC segments, blended from tau = 1 into their state slot
//   1  items1 again, SPEC items with s > 0 only: the wave forms the transmittance in FRONT of its segment, T_s = tau_0
//      ... tau_(s-1) (dense copies of the taus in st2, s KB per item: a per-tile prefix launch in between cost 19 us
//      of dependent loads); the pixels that FINISH inside this segment (T_s >= tau_stop > T_s tau_s:
//      the wave of launch 0 could not know) are blended again from tau = T_s, which stops them exactly where the
//      unsplit kernel does, and their state is replaced (last contributor stored NEGATIVE: "finished here").  Every
//      pixel finishes once, and only the 8x8 blocks that hold such a pixel are live: a fraction of one more segment
//      walk, all segments at once
//   2  items3 (COMPOSE): composes the SPEC segments in order, walks on from there while a pixel is alive, writes the
//      tile's pixels and turns the slots into the backward pass's segment-end states
// The blend loop is k_draw's, unchanged (one stop threshold for the whole wave: a re-walk or a continuation starts
// from the TRUE transmittance, not from 1).
template <bool FLOOR, bool CLAMP, int ROLE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ROLE == 0 ? 8 : (ROLE == 2 ? 2 : 5), 8))) void k_draw_seg(
    DrawParams p, SegArgs sg, int32_t* __restrict__ ranges, const int32_t* __restrict__ gsid,
    const float4* __restrict__ rec, float* __restrict__ image, int32_t* __restrict__ contrib,
    float* __restrict__ final_tau) {
  __shared__ float4 sA[64], sB[64], sC[64];
  const int lane = threadIdx.x;
  if (ROLE == 0 && p.zero_buf) {   // every workgroup of the grid clears its slice of the gradient records
    const uint32_t z0 = blockIdx.x * p.zero_per, z1 = min(p.zero_n4, z0 + p.zero_per);
    float4* __restrict__ zb = p.zero_buf;
    for (uint32_t i = z0 + lane; i < z1; i += 64) zb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  constexpr bool PER_TILE = ROLE == 2;
  // launch 1 runs FOUR waves per item, one per 8x8 block of the tile: what it costs is the latency of ONE lone wave
  // walking one segment (it executes 3 % of launch 0's instructions), and a wave that blends one block instead of up
  // to four walks an entry in a fraction of the time
  const uint32_t bix = ROLE == 1 ? blockIdx.x >> 2 : blockIdx.x;
  const int quad = ROLE == 1 ? (int)(blockIdx.x & 3u) : -1;
  if ((int)bix >= sg.hdr[PER_TILE ? SH_ITEMS3 : SH_ITEMS1]) return;
  const uint32_t item = (uint32_t)(PER_TILE ? sg.items3 : sg.items1)[bix];
  const int tile = (int)(item & SEG_TILE_MASK), iseg = (int)((item >> 19) & SEG_SEG_MASK), kind = (int)(item >> 30);
  if (tile >= p.T) return;
  if (ROLE == 1 && (kind != SEG_SPEC || iseg == 0)) return;   // (segment 0 starts from T = 1: launch 0 was exact)
  if (ROLE == 0 && sg.rebuild && kind != SEG_SPEC) return;    // (an unsplit tile has no state to rebuild; its item is
                                                              // the backward launch's)
  if (PER_TILE && kind != SEG_COMPOSE) return;
  const int L = sg.hdr[SH_L];
  const int r0 = ranges[2 * (size_t)tile], r1 = ranges[2 * (size_t)tile + 1];
  const int n = sg.rebuild ? min(r1 - r0, sg.walk[tile]) : r1 - r0;
  const int tx0 = (tile % p.gx) * EGS_TILE, ty0 = (tile / p.gx) * EGS_TILE;
  const int pxb[2] = {tx0 + (lane & 7), tx0 + (lane & 7) + 8};
  const int pyb[2] = {ty0 + (lane >> 3), ty0 + (lane >> 3) + 8};
  const size_t HW = (size_t)p.W * p.H;
  // (recomputed where they are used: nothing of this stays in registers across the blend loop)
  auto inside_px = [&](int k) { return (pxb[k & 1] < p.W) && (pyb[k >> 1] < p.H); };
  auto pix_of = [&](int k) { return (size_t)min(pyb[k >> 1], p.H - 1) * p.W + min(pxb[k & 1], p.W - 1); };
  if (n <= 0) {  // (DIRECT only) empty tile: zeros, final_tau = 0, ranges (0, 0) -- as k_draw
    if (ROLE != 0) return;
    if (p.work_out && lane == 0) p.work_out[tile] = 0;
    if (lane == 0) { sg.walk[tile] = 0; if (sg.hist_walk) sg.hist_walk[tile] = 0; }
    if (lane == 0 && (r0 != 0 || r1 != 0)) { ranges[2 * (size_t)tile] = 0; ranges[2 * (size_t)tile + 1] = 0; }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (inside_px(k)) {
        image[pix_of(k)] = 0.f; image[HW + pix_of(k)] = 0.f; image[2 * HW + pix_of(k)] = 0.f;
        contrib[pix_of(k)] = 0; final_tau[pix_of(k)] = 0.f;
      }
    return;
  }
  const int slot0 = sg.seg_base[tile];
  const float stop = p.tau_stop, lthr = p.lskip;
  const float X[2] = {(float)(lane & 7) - 7.5f, (float)(lane & 7) + 0.5f};
  const float Y[2] = {(float)(lane >> 3) - 7.5f, (float)(lane >> 3) + 0.5f};
  const float XX[2] = {X[0] * X[0], X[1] * X[1]}, YY[2] = {Y[0] * Y[0], Y[1] * Y[1]};
  const float XY[4] = {X[0] * Y[0], X[1] * Y[0], X[0] * Y[1], X[1] * Y[1]};
  constexpr float L99 = -0.014499569695115089f;
  const float cx0 = (float)tx0 + 7.5f, cy0 = (float)ty0 + 7.5f;
  float tau[4], cr[4], cg[4], cb[4];   // blend state of one walk; a pixel that does not take part holds tau = -1
  int cont[4];
  // ROLE 2: the transmittance in front of the current segment; negative: the pixel is finished (or outside the image),
  // |Tf| its final transmittance.  ROLE 1: the transmittance in front of segment iseg for the pixels to blend again.
  float Tf[4];
  int nseg = 1, nspec = 0, sdone = 0;
  if (ROLE == 1) {
    // the transmittance in FRONT of this segment, T_s = tau_0 tau_1 ... tau_(s-1) in that order (as COMPOSE forms it),
    // from the dense copies launch 0 left in st2: s x 1 KB per item, requested eight segments at a time
    const int qo = 64 * quad;            // this wave's block of the tile
    const bool qin = (pxb[quad & 1] < p.W) && (pyb[quad >> 1] < p.H);
    float T = qin ? 1.f : -1.f;
    constexpr int AHEAD = 8;
    for (int s0 = 0; s0 < iseg; s0 += AHEAD) {
      float tl[AHEAD];
#pragma unroll
      for (int u = 0; u < AHEAD; ++u) tl[u] = sg.st2[((size_t)(slot0 + min(s0 + u, iseg - 1))) * 256 + lane + qo];
#pragma unroll
      for (int u = 0; u < AHEAD; ++u)
        if (s0 + u < iseg) T *= tl[u];
      if (!__any(T >= stop)) return;       // every pixel of the block finished in front of this segment
    }
    const float tls = sg.st2[((size_t)(slot0 + iseg)) * 256 + lane + qo];
    const bool ev = (T >= stop) && (T * tls < stop);
#pragma unroll
    for (int k = 0; k < 4; ++k) Tf[k] = (k == quad && ev) ? T : -1.f;
    if (!__any(ev)) return;
  }
  if (ROLE == 2) {
    // ---- compose the SPEC segments: colour and last contributor in registers, the states requested ahead ----
    nseg = (n + L - 1) >> (31 - __clz(L)); nspec = iseg;
    float ca[4][3];
    int cc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { Tf[k] = inside_px(k) ? 1.f : -1.f; ca[k][0] = 0.f; ca[k][1] = 0.f; ca[k][2] = 0.f; cc[k] = 0; }
    // (a chain of dependent round trips to the slots: four segments are requested together)
    constexpr int CA = 4;
    bool done = false;
    for (int s0 = 0; s0 < nspec && !done; s0 += CA) {
      float4 v[CA][4];
      int c[CA][4];
#pragma unroll
      for (int u = 0; u < CA; ++u) {
        const size_t so = ((size_t)(slot0 + min(s0 + u, nspec - 1))) * 256 + lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[u][k] = sg.st4[so + 64 * k]; c[u][k] = __float_as_int(sg.st1[so + 64 * k]); }
      }
#pragma unroll
      for (int u = 0; u < CA; ++u) {
        if (done || s0 + u >= nspec) continue;
        bool alive = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) alive = alive || (Tf[k] >= stop);
        if (!__any(alive)) { done = true; continue; }
        sdone = s0 + u + 1;
        const size_t so = ((size_t)(slot0 + s0 + u)) * 256 + lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (!(Tf[k] >= stop)) continue;
          ca[k][0] = fmaf(Tf[k], v[u][k].x, ca[k][0]); ca[k][1] = fmaf(Tf[k], v[u][k].y, ca[k][1]);
          ca[k][2] = fmaf(Tf[k], v[u][k].z, ca[k][2]);
          float tn = Tf[k] * v[u][k].w;
          // (a negative contributor: launch 1 blended this pixel to its end inside the segment -- not decided again here
          // from a product that may round the other way)
          if (c[u][k] < 0 || tn < stop) tn = -fmaxf(tn, 1.0e-30f);
          if (c[u][k] != 0) cc[k] = abs(c[u][k]);
          Tf[k] = tn;
          sg.st1[so + 64 * k] = fabsf(tn);     // transmittance at the END of segment s
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (inside_px(k)) {
        image[pix_of(k)] = ca[k][0]; image[HW + pix_of(k)] = ca[k][1]; image[2 * HW + pix_of(k)] = ca[k][2];
        contrib[pix_of(k)] = cc[k];
      }
  }
  // ---- walks: launch 0 (a tile or one segment from tau = 1), launch 1 (one segment again, the finishing pixels from
  // their true transmittance), launch 2 (the segments behind the SPEC ones, one after the other, while a pixel is alive)
  for (int s = (ROLE == 2 ? nspec : 0); s < nseg; ++s) {
    int e0 = 0, e1 = n;
    if (ROLE != 2 && kind == SEG_SPEC) { e0 = iseg * L; e1 = min(n, e0 + L); }
    if (ROLE == 2) {
      if (sdone < s) break;          // (the composition above ended early: every pixel is finished)
      bool alive = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) alive = alive || (Tf[k] >= stop);
      if (!__any(alive)) break;
      e0 = s * L; e1 = min(n, e0 + L);
      sdone = s + 1;
    }
    int live = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (ROLE == 0) tau[k] = inside_px(k) ? 1.f : -1.f;
      else tau[k] = (Tf[k] >= stop) ? Tf[k] : -1.f;      // the TRUE transmittance in front of the segment
      cr[k] = 0.f; cg[k] = 0.f; cb[k] = 0.f; cont[k] = 0;
      if (__any(tau[k] >= stop)) live |= 1 << k;
    }
    // ---- the blend loop of k_draw over entries [e0, e1) ----
    int gnext = (e0 + lane < e1) ? gsid[r0 + e0 + lane] : 0;
    for (int base = e0; base < e1 && live != 0; base += 64) {
      __syncthreads();
      int mymask = 0;
      const int gm = gnext;
      const int g = p.masked ? (int)((uint32_t)gm & EGS_GSID_MASK) : gm;
      if (base + 64 + lane < e1) gnext = gsid[r0 + base + 64 + lane];
      if (base + lane < e1) {
        float4 A = rec[3 * (size_t)g], B = rec[3 * (size_t)g + 1];
        const float4 C = rec[3 * (size_t)g + 2];
        const bool nanfix = p.nan_blend && nan_entry_fix(A, B);
        if (C.w < INFINITY) mymask = p.masked ? (int)((uint32_t)gm >> EGS_GSID_BITS) : reach_mask<false>(A, C, tx0, ty0);
        // (as k_draw: a NaN entry blends everywhere, its certain-miss box says nothing -- unmasked lists: the public
        // splatB's rebuild, EGS_CULL_LISTS=0)
        if (nanfix && !p.masked && C.w < INFINITY) mymask = 0xF;
        const float la = lthr - C.w;
        float cap = 3.0e38f;
        if (FLOOR) cap = CLAMP ? fminf(la, L99) : la;
        else if (CLAMP) cap = L99;
        const float Dx = cx0 - A.x, Dy = cy0 - A.y;
        const float c0 = la + (A.z * Dx * Dx + A.w * Dx * Dy + B.x * Dy * Dy);
        const float c1 = 2.f * A.z * Dx + A.w * Dy, c2 = 2.f * B.x * Dy + A.w * Dx;
        sA[lane] = make_float4(A.z, A.w, B.x, cap);
        sB[lane] = make_float4(c0, c1, c2, B.z);
        *reinterpret_cast<float2*>(&sC[lane]) = make_float2(B.w, C.x);
      }
      __syncthreads();
      int pk = mymask;
      pk |= __shfl_down(pk, 1, 64) << 4;
      pk |= __shfl_down(pk, 2, 64) << 8;
      pk |= __shfl_down(pk, 4, 64) << 16;
      const int m = __builtin_amdgcn_readfirstlane(min(64, e1 - base));
      for (int j0 = 0; j0 < m && live != 0; j0 += 8) {
        const uint32_t act = (uint32_t)__builtin_amdgcn_readlane(pk, j0) & ((uint32_t)live * 0x11111111u);
        if (act != 0u) {
          const int vidx0 = base + j0 + 1;
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const int reach = (int)((act >> (4 * t)) & 0xFu);
            if (reach != 0) {
              const int j = j0 + t;
              const float4 Q = sA[j], Pq = sB[j];
              const float2 gb = *reinterpret_cast<const float2*>(&sC[j]);
              const int idx = vidx0 + t;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const int bx = k & 1, by = k >> 1;
                if (reach & (1 << k)) {
                  float e = fmaf(Pq.z, Y[by], Pq.x);
                  e = fmaf(Pq.y, X[bx], e);
                  e = fmaf(Q.z, YY[by], e);
                  e = fmaf(Q.y, XY[k], e);
                  e = fmaf(Q.x, XX[bx], e);
                  if ((tau[k] >= stop) && (e >= lthr)) {
                    if (FLOOR || CLAMP) e = min_hi(e, Q.w);
                    const float w = tau[k] * __builtin_amdgcn_exp2f(e);
                    cr[k] += w * Pq.w; cg[k] += w * gb.x; cb[k] += w * gb.y;
                    tau[k] -= w;
                    cont[k] = idx;
                  }
                }
              }
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if ((live & (1 << k)) && !__any(tau[k] >= stop)) live &= ~(1 << k);
        }
      }
    }
    if (ROLE == 0 && kind == SEG_SPEC) {     // the segment's local state: (colour, tau) and its last contributor
      const size_t so = ((size_t)(slot0 + iseg)) * 256 + lane;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sg.st4[so + 64 * k] = make_float4(cr[k], cg[k], cb[k], tau[k]);
        sg.st1[so + 64 * k] = __int_as_float(cont[k]);
        // (dense copy: launch 1 multiplies the taus in front of a segment; st2 follows st1)
        sg.st1[so + 64 * k + (size_t)sg.slot_cap * 256] = tau[k];
      }
      return;
    }
    if (ROLE == 1) {     // the pixels that finish inside this segment, blended from their true transmittance
      const size_t so = ((size_t)(slot0 + iseg)) * 256 + lane;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (Tf[k] >= stop) {
          const float rt = 1.f / Tf[k];      // back into the segment's local frame (Tf >= tau_stop: no blow-up)
          sg.st4[so + 64 * k] = make_float4(cr[k] * rt, cg[k] * rt, cb[k] * rt, tau[k] * rt);
          // negative: "finished here"; a pixel the walk did not finish after all stays an ordinary one
          sg.st1[so + 64 * k] = __int_as_float(tau[k] < stop ? -cont[k] : cont[k]);
        }
      return;
    }
    if (ROLE == 2) {     // a continuation segment, walked right here from Tf: absolute colour, exact stop
      if (lane == 0) {   // the backward launch walks it with a wave of its own: one more SPEC item
        const int at = atomicAdd(&sg.hdr[SH_ITEMS1], 1);
        if (at < sg.item_cap) sg.items1[at] = (int32_t)((uint32_t)tile | ((uint32_t)s << 19) | ((uint32_t)SEG_SPEC << 30));
      }
      const size_t so = ((size_t)(slot0 + s)) * 256 + lane;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!(Tf[k] >= stop)) {   // (the scan below wants finite numbers in every slot it reads)
          sg.st4[so + 64 * k] = make_float4(0.f, 0.f, 0.f, 1.f);
          continue;
        }
        const float rt = 1.f / Tf[k];
        sg.st4[so + 64 * k] = make_float4(cr[k] * rt, cg[k] * rt, cb[k] * rt, tau[k] * rt);
        image[pix_of(k)] += cr[k]; image[HW + pix_of(k)] += cg[k]; image[2 * HW + pix_of(k)] += cb[k];
        if (cont[k] > 0) contrib[pix_of(k)] = cont[k];
        Tf[k] = (tau[k] < stop) ? -fmaxf(tau[k], 1.0e-30f) : tau[k];
        sg.st1[so + 64 * k] = fabsf(Tf[k]);     // transmittance at the END of segment s
      }
    }
  }
  // ---- the tile's pixels -------------------------------------------------------------------------------------
  int cfin[4];
  if (ROLE == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      cfin[k] = cont[k];
      if (inside_px(k)) {
        image[pix_of(k)] = cr[k]; image[HW + pix_of(k)] = cg[k]; image[2 * HW + pix_of(k)] = cb[k];
        contrib[pix_of(k)] = cont[k]; final_tau[pix_of(k)] = tau[k];
      }
    }
  } else {
    // COMPOSE: what the backward pass needs at the end of segment s -- the transmittance there (already in st1) and
    // G_s, the colour of everything behind it seen from there: G_last = 0, G_(s-1) = C_s + tau_s G_s over the
    // segments the pixel was alive in (it finished in the segment of its last contributor, if it finished).
    float G[4][3];
    int sstar[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      cfin[k] = inside_px(k) ? contrib[pix_of(k)] : 0;
      if (inside_px(k)) final_tau[pix_of(k)] = fabsf(Tf[k]);
      sstar[k] = (Tf[k] < stop) ? (max(cfin[k] - 1, 0) >> (31 - __clz(L))) : sdone - 1;
      G[k][0] = 0.f; G[k][1] = 0.f; G[k][2] = 0.f;
    }
    constexpr int GA = 4;       // (again four slots per round trip; a slot is read before this loop overwrites it)
    for (int s1 = sdone - 1; s1 >= 0; s1 -= GA) {
      float4 vv[GA][4];
      float te[GA][4];
#pragma unroll
      for (int u = 0; u < GA; ++u) {
        const size_t sp = ((size_t)(slot0 + max(s1 - u, 0))) * 256 + lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) { vv[u][k] = sg.st4[sp + 64 * k]; te[u][k] = sg.st1[sp + 64 * k]; }
      }
#pragma unroll
      for (int u = 0; u < GA; ++u) {
        const int s = s1 - u;
        if (s < 0) continue;
        const size_t so = ((size_t)(slot0 + s)) * 256 + lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          sg.st4[so + 64 * k] = make_float4(G[k][0], G[k][1], G[k][2], te[u][k]);
          if (s <= sstar[k]) {
            G[k][0] = fmaf(vv[u][k].w, G[k][0], vv[u][k].x); G[k][1] = fmaf(vv[u][k].w, G[k][1], vv[u][k].y);
            G[k][2] = fmaf(vv[u][k].w, G[k][2], vv[u][k].z);
          }
        }
      }
    }
  }
  {   // how far the tile was walked: the work measure of the dispatch orders and the backward pass's segment count
    int w = 0, wmax = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int mx = cfin[k];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d, 64));
      w += mx;
      wmax = max(wmax, mx);
    }
    if (lane == 0) {
      if (p.work_out) p.work_out[tile] = w + 2 * wmax;
      if (!sg.rebuild) sg.walk[tile] = wmax;      // (rebuilding: the walk of the pass whose `contrib` was handed in stays)
      if (sg.hist_walk) sg.hist_walk[tile] = wmax;
      walk_raise(&sg.hdr[SH_MAXWALK], wmax);
      walk_raise(sg.walk_max, wmax);     // (-> the host's hint words, by the next render's range kernel)
      if (sg.walk_max) walk_raise(sg.walk_max + 1, r1 - r0);     // ... with the longest list of the render (DIRECT and COMPOSE items)
    }
  }
}

// plan + the forward launches over a carved workspace (see the ROLE comment above k_draw_seg)
int draw_segments_forward(DrawParams& dp, const EgsPolicy* pol, SegArgs& sga, const SegConfig& cfg, int64_t patches,
                          const int32_t* hist, int speculate, bool fix_pass, int32_t* walk_word, uint32_t* seg_hint,
                          bool plan_zeroed, int32_t* ranges, const int32_t* gsid, const float4* rec, float* image,
                          int32_t* contrib, float* final_tau, hipStream_t s) {
  if (!plan_zeroed) EGS_HIP(hipMemsetAsync(sga.hdr, 0, (size_t)SEG_PLAN_WORDS * 4, s));
  sga.walk_max = walk_word;
  const int pg = div_up(dp.T, 256);
  EGS_LAUNCH("k_seg_plan_count", k_seg_plan_count, dim3(pg), dim3(256), s, dp.T, ranges, hist, cfg.L, cfg.split_min, sga, speculate);
  EGS_LAUNCH("k_seg_plan_place", k_seg_plan_place, dim3(pg), dim3(256), s, dp.T, hist, cfg.L, cfg.split_min, sga, seg_hint);
  // items <= tiles + segments <= T + P / L + P / split_min: the launch covers the bound, surplus workgroups exit
  const int grid1 = (int)std::min<int64_t>(seg_item_bound(dp.T, patches, cfg), sga.item_cap);
  if (dp.zero_buf) dp.zero_per = (dp.zero_n4 + (uint32_t)grid1 - 1) / (uint32_t)grid1;
#define EGS_DRAWS(FLOOR, CLAMP, ROLE, NAME, GRID)                                                                 \
  EGS_LAUNCH(NAME, (k_draw_seg<FLOOR, CLAMP, ROLE>), dim3(GRID), dim3(64), s, dp, sga, ranges, gsid, rec, image, contrib, \
             final_tau)
#define EGS_DRAWS3(FLOOR, CLAMP)                                                                                  \
  do {                                                                                                            \
    EGS_DRAWS(FLOOR, CLAMP, 0, "k_draw_seg", grid1);                                                              \
    if (fix_pass) {   /* (else segment 0 is the only SPEC item of a tile, and it is exact) */                        \
      EGS_DRAWS(FLOOR, CLAMP, 1, "k_draw_seg_fix", 4 * grid1);                                                    \
    }                                                                                                             \
    EGS_DRAWS(FLOOR, CLAMP, 2, "k_draw_seg_compose", dp.T);                                                       \
  } while (0)
  switch ((pol->maha_floor ? 2 : 0) | (pol->alpha_clamp ? 1 : 0)) {
    case 0: EGS_DRAWS3(false, false); break;
    case 1: EGS_DRAWS3(false, true); break;
    case 2: EGS_DRAWS3(true, false); break;
    default: EGS_DRAWS3(true, true); break;
  }
#undef EGS_DRAWS3
#undef EGS_DRAWS
  EGS_LAUNCH_OK();
  return 0;
}

}  // namespace egs

using namespace egs;

extern "C" size_t egs_seg_ws_bytes(int64_t patch_capacity, int width, int height) {
  const SegConfig c = seg_config();
  const int T = div_up(width, EGS_TILE) * div_up(height, EGS_TILE);
  const int64_t P = patch_capacity > 0 ? patch_capacity : 1;
  return seg_ws_bytes_for(P / c.L + P / c.split_min + 64, T);
}
extern "C" int egs_seg_config(int segment_len, int split_min, int* out2) {
  SegConfig c = seg_config();
  if (out2) { out2[0] = c.L; out2[1] = c.split_min; }
  if (segment_len > 0) {
    EGS_CHECK_ARG(segment_len >= 64 && segment_len <= 65536 && (segment_len & (segment_len - 1)) == 0);
    c.L = segment_len;
  }
  if (split_min > 0) c.split_min = split_min;
  if (c.split_min < c.L) c.split_min = c.L;
  g_seg_cfg.store(((uint64_t)(uint32_t)c.split_min << 32) | (uint32_t)c.L);
  return 0;
}
