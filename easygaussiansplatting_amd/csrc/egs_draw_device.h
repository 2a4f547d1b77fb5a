// Device helpers shared by the blend loops of egs_draw.hip (k_draw, k_draw_bwd) and egs_segments.hip (k_draw_seg).
#pragma once
#include "egs_raster.h"

namespace egs {

// 4-bit reach mask of one list entry over the four 8x8 blocks of a tile (bit k = block
// (k&1, k>>1)).  Computed ONCE per entry by the lane that stages it (64 entries in
// parallel) instead of by all 64 lanes of the blend loop.
template <bool BOX>
__device__ __forceinline__ int reach_mask(const float4& A, const float4& C, int tx0, int ty0) {
  bool okx[2], oky[2];
  if (BOX) {  // overlap of the pixel box (gausplat.py:212-215) with the block
    const uint32_t bx = __float_as_uint(C.y), by = __float_as_uint(C.z);
    const int x0 = bx & 0xFFFF, x1 = bx >> 16, y0 = by & 0xFFFF, y1 = by >> 16;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      okx[b] = (x0 < tx0 + 8 * b + 8) && (x1 > tx0 + 8 * b);
      oky[b] = (y0 < ty0 + 8 * b + 8) && (y1 > ty0 + 8 * b);
    }
  } else {    // certain-miss box (ex, ey) of the pack kernel vs the block (half size 3.5 px)
    const float rx = C.y + 3.5f, ry = C.z + 3.5f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      okx[b] = fabsf(A.x - ((float)tx0 + 3.5f + 8.f * b)) <= rx;
      oky[b] = fabsf(A.y - ((float)ty0 + 3.5f + 8.f * b)) <= ry;
    }
  }
  return (int)(okx[0] && oky[0]) | ((int)(okx[1] && oky[0]) << 1) | ((int)(okx[0] && oky[1]) << 2) |
         ((int)(okx[1] && oky[1]) << 3);
}

// CUDA's max(0.0f, NaN) == 0 (kernel.cu:243-246, 909-913): a Mahalanobis term that is NaN counts as 0 and the Gaussian
// blends at min(0.99, alpha).  A NaN in the conic or the centre of an entry makes EVERY pixel's term NaN, so the lane that
// stages the entry decides it once: conic := 0, centre := 0 -- the exponent is then log2(alpha) everywhere (forward) and
// the power 0 (backward).  Per entry, 64 entries in parallel, nothing in the blend loops.  (A NaN that arises at single
// pixels from inf * 0 is not covered: EgsPolicy.nan_maha.)
__device__ __forceinline__ bool nan_entry_fix(float4& A, float4& B) {
  if (A.x != A.x || A.y != A.y || A.z != A.z || A.w != A.w || B.x != B.x) {
    A = make_float4(0.f, 0.f, 0.f, 0.f);
    B.x = 0.f;
    return true;
  }
  return false;
}

// The longest walk of a render -> the host's hint word, without a launch of its own.  The host reads that word WITHOUT
// waiting, while the GPU may be in the middle of a render, so it must only ever hold the maximum of a COMPLETED render.
// Two things were tried first in round 6 and measured: a running maximum mirrored into the host word by whichever wave
// raises it (the host saw 300 of an 8 000-entry walk, took the unsplit kernels, and the path flip-flopped), and tickets
// -- every wave counts its arrival, the last one publishes (k_draw 152 -> 452 us: 8 160 returning device-scope atomics
// on one cache line serialise at ~40 ns each, sharded or not).  What is left costs nothing: the waves gather the
// maximum in a PERSISTENT device word the host keeps per (problem size, stream) -- the conditional load keeps all but a
// few dozen waves off the atomic -- and the range kernel of the NEXT render on that stream publishes it (stream order:
// the render it belongs to is complete) and resets it to -1 = "nothing gathered yet".
// (Round 5: a k_seg_report launch behind the compose launch, 6 us; nothing on the unsplit path, where the word was
// refreshed every 4th render of a camera by k_tile_order -- after reset_alpha a trainer kept walking 8 000-entry lists
// with one wave per tile for an epoch.)
__device__ __forceinline__ void walk_raise(int32_t* __restrict__ word, int wmax) {
  if (word && wmax > *word) atomicMax(word, wmax);
}

// min(x, hi) as ONE v_med3_f32 (fminf() costs a canonicalising v_max + v_min in IEEE mode; the
// low bound is finite so the compiler cannot fold the median back into a min)
__device__ __forceinline__ float min_hi(float x, float hi) {
  return __builtin_amdgcn_fmed3f(x, hi, -3.0e38f);
}

}  // namespace egs
