// Exact nearest-neighbour squared distance of every point of a cloud to the rest of the cloud:
// out[i] = min_{j != i} |p_i - p_j|^2.  This is what the reference takes from
// faiss.IndexFlatL2(3).search(pws, 2)[0][:, 1] (gsplat/read_write_model.py:216-220) to
// initialise the Gaussian scales from a COLMAP point cloud (faiss is an exact brute-force index;
// duplicates give 0 exactly like the second column of the faiss result).
//
// Brute force on the VALU: K = 3, so there is no contraction worth an MFMA; the work is
// 7 fp32 ops per (query, candidate).  Candidates are wave-uniform -> they are fetched with scalar
// loads (s_load_dwordx4, one per candidate per WAVE, not per lane) and never touch LDS or the
// vector memory path; each lane keeps QPT queries in registers.  N = 1 M: 1e12 pairs.
#include <float.h>

#include "egs_common.h"

namespace egs {

constexpr int KNN_QPT = 4;               // queries per thread
constexpr int KNN_BLOCK = 256;
constexpr int KNN_QPB = KNN_QPT * KNN_BLOCK;

__global__ __launch_bounds__(256) void k_knn_pack(int n, const float* __restrict__ pts, float4* __restrict__ out,
                                                  float* __restrict__ best) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], 0.f);
  best[i] = FLT_MAX;
}

template <bool SELF>
__device__ __forceinline__ void knn_range(const float4* __restrict__ cand, int j0, int j1, const float (&qx)[KNN_QPT],
                                          const float (&qy)[KNN_QPT], const float (&qz)[KNN_QPT],
                                          const int (&qi)[KNN_QPT], float (&best)[KNN_QPT]) {
#pragma unroll 4
  for (int j = j0; j < j1; ++j) {
    const float4 c = cand[j];            // uniform address: scalar load
#pragma unroll
    for (int k = 0; k < KNN_QPT; ++k) {
      const float dx = qx[k] - c.x, dy = qy[k] - c.y, dz = qz[k] - c.z;
      float d = dx * dx + dy * dy + dz * dz;
      if (SELF) d = (j == qi[k]) ? FLT_MAX : d;
      best[k] = fminf(best[k], d);
    }
  }
}

// grid = (query blocks, candidate slices): small clouds would otherwise launch fewer workgroups than
// there are CUs (200 k points = 196 query blocks).  Slices merge with atomicMin on the bit pattern
// (non-negative floats order like unsigned integers).
__global__ __launch_bounds__(KNN_BLOCK) void k_knn_sqdist(int n, int slice, const float4* __restrict__ cand,
                                                          float* __restrict__ out) {
  const int base = blockIdx.x * KNN_QPB;
  float qx[KNN_QPT], qy[KNN_QPT], qz[KNN_QPT], best[KNN_QPT];
  int qi[KNN_QPT];
#pragma unroll
  for (int k = 0; k < KNN_QPT; ++k) {
    qi[k] = base + k * KNN_BLOCK + threadIdx.x;
    const float4 q = cand[min(qi[k], n - 1)];
    qx[k] = q.x; qy[k] = q.y; qz[k] = q.z;
    best[k] = FLT_MAX;
  }
  const int c0 = min((int)blockIdx.y * slice, n), c1 = min(c0 + slice, n);
  const int own0 = min(max(base, c0), c1), own1 = min(max(base + KNN_QPB, c0), c1);
  knn_range<false>(cand, c0, own0, qx, qy, qz, qi, best);      // the self test is only needed in the
  knn_range<true>(cand, own0, own1, qx, qy, qz, qi, best);     // block's own index range
  knn_range<false>(cand, own1, c1, qx, qy, qz, qi, best);
#pragma unroll
  for (int k = 0; k < KNN_QPT; ++k)
    if (qi[k] < n) atomicMin((unsigned int*)out + qi[k], __float_as_uint(best[k]));
}

}  // namespace egs

using namespace egs;

extern "C" size_t egs_nn_sqdist_ws_bytes(int n) { return align_up((size_t)(n > 0 ? n : 1) * sizeof(float4), 256) + 256; }

extern "C" int egs_nn_sqdist(int n, const float* points, void* ws, size_t ws_bytes, float* out_sqdist, void* stream) {
  EGS_CHECK_ARG(n >= 0);
  if (n == 0) return 0;
  EGS_CHECK_ARG(points && ws && out_sqdist);
  EGS_CHECK_ARG(ws_bytes >= egs_nn_sqdist_ws_bytes(n));
  hipStream_t s = (hipStream_t)stream;
  Carver cv(ws, ws_bytes);
  float4* cand = cv.take<float4>(n);
  EGS_LAUNCH("k_knn_pack", k_knn_pack, dim3(div_up(n, 256)), dim3(256), s, n, points, cand, out_sqdist);
  EGS_LAUNCH_OK();
  const int qblocks = div_up(n, KNN_QPB);
  int slices = div_up(4096, qblocks);                          // >= 4096 workgroups (256 CUs x 4 SIMDs x 4)
  if (slices > div_up(n, 4096)) slices = div_up(n, 4096);      // but at least 4096 candidates per slice
  const int slice = div_up(n, slices);
  EGS_LAUNCH("k_knn_sqdist", k_knn_sqdist, dim3(qblocks, div_up(n, slice)), dim3(KNN_BLOCK), s, n, slice,
             (const float4*)cand, out_sqdist);
  EGS_LAUNCH_OK();
  return 0;
}
