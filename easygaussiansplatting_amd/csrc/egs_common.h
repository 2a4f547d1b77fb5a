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

// ---- binning pieces shared by egs_raster.hip (egs_splat_bin) and egs_preprocess.hip (the fused forward
// kernel does getRects + the depth key itself) ---------------------------------------------------------
struct BinParams {
  int W, H, gx, gy;
  int footprint, far_cull, depth_key, mutate;
};
// tile rect of one Gaussian packed in 8 bytes: {x0 | y0 << 16, w | h << 16} (tiles; w * h = its patch count)
__host__ __device__ inline uint2 pack_rect(uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
  return make_uint2(x0 | (y0 << 16), (x1 - x0) | ((y1 - y0) << 16));
}
struct BinCountOut {  // where k_bin_count's results live inside the bin workspace
  uint2* rc;                       // packed rect (+ implied count) per Gaussian
  uint32_t *dkeys, *ids, *maxkey;  // maxkey[1 + workgroup] = per-workgroup maximum of the depth keys
};
BinParams make_bin_params(int width, int height, const EgsPolicy* pol);
bool bin_count_outputs(void* ws_bin, size_t ws_bin_bytes, int n, BinCountOut* out);
// everything of egs_splat_bin after k_bin_count (max reduce, depth sort, offsets scan)
// host_totals (nullable): page-locked host uint32[2] the kernels ALSO write {P, max depth key} into (mailbox slot)
int splat_bin_after_count(int n, int key_bits_hint, void* ws_bin, size_t ws_bin_bytes, uint32_t* total_patches,
                          void* stream, uint32_t* host_totals);

// splatB's draw pass into the packed [N][12] gradient records (egs_raster.hip); *gpack
// points into `ws`.  Shared by egs_splat_bwd (+unpack) and egs_fused_backward.
int splat_bwd_packed(int n, int64_t patches, int width, int height, const float* us, const float* cinv2ds,
                     const float* alphas, const float* colors, const int32_t* areas, const EgsPolicy* pol,
                     const int32_t* contrib, const float* final_tau, const int32_t* patch_range_per_tile,
                     const int32_t* gsid_per_patch, const float* dloss_dgammas, void* ws, size_t ws_bytes,
                     float** gpack, void* stream, const void* rec_in /* packed records or NULL */,
                     const int32_t* tile_order /* dispatch order left by the forward pass, or NULL */,
                     float* grad_records /* [N][12] records ALREADY ZEROED (by the forward draw kernel), or NULL */,
                     bool keep_forward_order = false /* dispatch the tiles exactly as tile_order says */);

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
