// Shared host/device helpers of libegs_hip.so (gfx950 / CDNA4 only: wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/egs_hip.h"

#define EGS_MIN_DEPTH 0.2f   // reference kernel.cu:10
#define EGS_BAD_MARKER (-1.f) // reference kernel.cu:11
#define EGS_TILE 16          // reference common.cuh:12 (BLOCK)
#define EGS_WAVE 64

namespace egs {

// ---- error reporting --------------------------------------------------------
void set_error(int code, const char* what, const char* file, int line);

#define EGS_CHECK_ARG(cond)                                                      \
  do {                                                                           \
    if (!(cond)) {                                                               \
      ::egs::set_error(EGS_ERR_BAD_ARG, "bad argument: " #cond, __FILE__, __LINE__); \
      return EGS_ERR_BAD_ARG;                                                    \
    }                                                                            \
  } while (0)

#define EGS_HIP(expr)                                                            \
  do {                                                                           \
    hipError_t e__ = (expr);                                                     \
    if (e__ != hipSuccess) {                                                     \
      ::egs::set_error((int)e__, hipGetErrorString(e__), __FILE__, __LINE__);    \
      return (int)e__;                                                           \
    }                                                                            \
  } while (0)

// after a <<<>>> launch: catches launch-configuration errors without syncing
#define EGS_LAUNCH_OK() EGS_HIP(hipGetLastError())

// ---- optional per-kernel timing (egs_prof_* in the C ABI) -------------------
bool prof_on(const char* name);
void prof_begin(const char* name, hipStream_t s);
void prof_end(hipStream_t s);
struct ProfScope {
  hipStream_t s;
  bool on;
  ProfScope(const char* name, hipStream_t st) : s(st), on(prof_on(name)) {
    if (on) prof_begin(name, s);
  }
  ~ProfScope() {
    if (on) prof_end(s);
  }
};
// launch `kern` on `stream`, bracketed by events when profiling is enabled
#define EGS_LAUNCH(name, kern, grid, block, stream, ...)                 \
  do {                                                                   \
    ::egs::ProfScope ps__(name, stream);                                 \
    hipLaunchKernelGGL(kern, grid, block, 0, stream, __VA_ARGS__);       \
  } while (0)
// same with `lds` bytes of dynamic LDS (used only to cap residency: see k_draw launch)
#define EGS_LAUNCH_LDS(name, kern, grid, block, lds, stream, ...)        \
  do {                                                                   \
    ::egs::ProfScope ps__(name, stream);                                 \
    hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);     \
  } while (0)

static inline int div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// bump allocator over a caller-supplied workspace (256-B aligned pieces)
struct Carver {
  char* base;
  size_t off;
  size_t cap;
  Carver(void* p, size_t bytes) : base((char*)p), off(0), cap(bytes) {}
  template <typename T>
  T* take(size_t count) {
    size_t start = align_up(off, 256);
    off = start + count * sizeof(T);
    return (T*)(base + start);
  }
  bool ok() const { return off <= cap; }
};

// ---- binning pieces shared by egs_bin.hip (egs_splat_bin) and egs_preprocess.hip (the fused forward
// kernel does getRects + the depth key itself) ---------------------------------------------------------
struct BinParams {
  int W, H, gx, gy;
  int footprint, far_cull, depth_key, mutate;
  int cull_lists;   // lists may drop the tiles of a rect that the footprint alpha' >= alpha_skip cannot reach (fused path)
};
// What the binning stage keeps of a Gaussian (32 bytes, gathered ONCE into depth order by the depth sort's last
// scatter pass): its footprint -- centre u, the conic in "power form" and the threshold m:
//     the Gaussian blends into a pixel  <=>  A dx^2 + 2 Bh dx dy + C dy^2 <= m        (d = pixel - u)
// (A, 2 Bh, C) = -(qxx, qxy, qyy) of the draw record, m = -thr = log2(alpha / alpha_skip): exactly the skip test of
// the draw kernels (kernel.cu:246) -- and its tile rect (getRects, kernel.cu:82-122).  m = +inf: no footprint
// culling, the Gaussian is emitted for every tile of its rect (the reference's lists).
struct BinRec {
  float ux, uy, A, Bh;
  float C, m;
  uint32_t xy;   // x0 | y0 << 16   (tiles)
  uint32_t wh;   // w | h << 16
};
static_assert(sizeof(BinRec) == 32, "BinRec is two dwordx4");
// The compact form the binning stage works with (16 bytes, gathered ONCE into depth order by the depth sort's last
// scatter pass):  {x0 | y0 << 16,  w | h << 16 | flags,  b_lo, b_hi}  (tiles).
//   rect of at most 4 x 4 tiles:  b = 64-bit bitmap of the 8x8 pixel blocks of the rect the footprint reaches, bit
//     8 by + bx relative to the rect's first block (ALL blocks of the rect when the Gaussian is not cullable).  The
//     Gaussian is emitted for the tiles with a block set, the list value's mask is the tile's four bits: emission is
//     bit arithmetic, the square roots of the footprint are taken once per Gaussian, in k_preprocess_fwd.
//   rect of at most 8 x 8 tiles (EGS_CR_TILEMAP, cullable Gaussians only):  b = 64-bit bitmap of the TILES of the rect
//     the footprint reaches, bit 8 ty + tx; emission finds slot r's tile by bit arithmetic and evaluates the two
//     8-row slabs of THAT tile with the full footprint record br[gaussian] for its block mask (11 % of the patches of
//     a ring view of the 1 M scene sit in rects larger than 4 x 4 tiles, 0.3 % of view 0's).
//   larger rect (EGS_CR_BIG):  b_lo = patch count, b_hi != 0: cullable -- k_bin_emit walks the rows of the rect with
//     br[gaussian]; b_hi == 0: every tile of the rect.
//   EGS_CR_ALLTILES (seven-op surface, k_pack_bin): the lists are the REFERENCE's -- every tile of the rect, row-major --
//     and the footprint only supplies the block mask of each tile (possibly empty: the entry stays in the list and is
//     never evaluated).  Rect of at most 4 x 4 tiles: b = the block bitmap as above; EGS_CR_BIG with b_hi == 2: every
//     tile of the rect, masks from br[gaussian].
#define EGS_CR_BIG 0x80000000u
#define EGS_CR_TILEMAP 0x40000000u
#define EGS_CR_ALLTILES 0x20000000u
#define EGS_CR_WH_MASK 0x1FFFFFFFu
struct BinCountOut {  // where k_bin_count's results live inside the bin workspace
  uint4* cr;                       // compact bin record per Gaussian
  BinRec* br;                      // footprint record, written for the cullable Gaussians with a rect larger than 4 x 4 tiles only
  uint32_t *dkeys, *ids, *maxkey;  // maxkey[1 + workgroup] = per-workgroup maximum of the depth keys
  uint32_t* sort_sup;              // the depth sort's superblock sums: the producing kernel ZEROES them on the side
  uint32_t sort_sup_words;
};
// list values of the culled lists (fused path): Gaussian index in the low 28 bits, in the high 4 the 8x8 pixel blocks
// of the tile the footprint reaches (bit k = block (k&1, k>>1)) -- computed ONCE per (tile, Gaussian) at emission
// instead of per entry and draw kernel, and exact where the draw kernels' own box test is conservative
#define EGS_GSID_BITS 28
#define EGS_GSID_MASK 0x0FFFFFFFu
BinParams make_bin_params(int width, int height, const EgsPolicy* pol, bool cull_lists = false);
bool bin_count_outputs(void* ws_bin, size_t ws_bin_bytes, int n, BinCountOut* out);
// everything of egs_splat_bin after k_bin_count (max reduce, depth sort, offsets scan)
// host_totals (nullable): page-locked host uint32[2] the kernels ALSO write {P, max depth key} into (mailbox slot)
int splat_bin_after_count(int n, int key_bits_hint, void* ws_bin, size_t ws_bin_bytes, uint32_t* total_patches,
                          void* stream, uint32_t* host_totals);

// splatB's draw pass into the packed [N][12] gradient records (egs_splat.hip); *gpack
// points into `ws`.  Shared by egs_splat_bwd (+unpack) and egs_fused_backward.
int splat_bwd_packed(int n, int64_t patches, int width, int height, const float* us, const float* cinv2ds,
                     const float* alphas, const float* colors, const int32_t* areas, const EgsPolicy* pol,
                     const int32_t* contrib, const float* final_tau, const int32_t* patch_range_per_tile,
                     const int32_t* gsid_per_patch, const float* dloss_dgammas, void* ws, size_t ws_bytes,
                     float** gpack, void* stream, const void* rec_in /* packed records or NULL */,
                     const int32_t* tile_order /* dispatch order left by the forward pass, or NULL */,
                     float* grad_records /* [N][12] records ALREADY ZEROED (by the forward draw kernel), or NULL */,
                     bool keep_forward_order = false /* dispatch the tiles exactly as tile_order says */,
                     bool masked_lists = false /* gsid_per_patch carries block masks (culled lists, fused path) */,
                     void* seg_ws = nullptr /* the segment workspace the forward draw filled (egs_splat_draw_rec_seg) */,
                     size_t seg_ws_bytes = 0, int rebuild = 0 /* seg_ws is fresh: rebuild the states from contrib first */,
                     uint32_t* seg_hint = nullptr /* page-locked words that learn the longest walk */);

// ---- device helpers ---------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive scan over a 256-thread block; `total` receives the block sum.
// `smem` must hold >= 4 uint32 and is reused on return after a barrier.
__device__ __forceinline__ uint32_t block256_exclusive_scan(uint32_t v, uint32_t* smem, uint32_t* total) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t inc = wave_inclusive_scan(v);
  if (lane == 63) smem[wave] = inc;
  __syncthreads();
  uint32_t s0 = smem[0], s1 = smem[1], s2 = smem[2], s3 = smem[3];
  uint32_t off = (wave > 0 ? s0 : 0u) + (wave > 1 ? s1 : 0u) + (wave > 2 ? s2 : 0u);
  if (total) *total = s0 + s1 + s2 + s3;
  __syncthreads();
  return off + inc - v;
}

// sum over the 64 lanes of a wave; result valid in every lane
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
#endif

}  // namespace egs
