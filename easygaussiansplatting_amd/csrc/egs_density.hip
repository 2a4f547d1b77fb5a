// Adaptive density control + optimizer surgery + fused Adam for gfx950.
//
// Replaces, on the device and without host round trips other than ONE 16-byte count read-back:
//   GSModel.update_density_info      reference gsplat/gsmodel.py:214-230
//   GSModel.update_gaussian_density  reference gsplat/gsmodel.py:232-317
//     + prune_params (151-166) and update_params (132-148): the reference builds ~40 masked
//       copies / torch.cat's of the 6 parameter tensors and their 12 Adam moment tensors;
//       here one classify pass, one 3-way scan and ONE compaction kernel move every row once.
//   GSModel.reset_alpha              reference gsplat/gsmodel.py:319-330
//   torch.optim.Adam.step            as configured by reference train.py:32 (6 groups, eps 1e-15):
//       one launch over all groups (7 x 236 B per Gaussian of HBM traffic, nothing else).
//
// Output row order == the reference's: [survivors in order | clones in order | split children in
// order].  The split offsets come either from a caller-supplied unit-normal table (parity tests)
// or from a counter-based generator keyed by (seed, round, ORIGINAL row index) -- every data-parallel
// replica computes bit-identical new Gaussians without communicating (the reference draws from the
// device RNG stream, gsmodel.py:274).
#include "egs_common.h"
#include "egs_gaussian_math.h"

namespace egs {

constexpr int DB = 256;                 // rows per workgroup
constexpr int NT = 6;                   // pws, low_shs, high_shs, alphas_raw, scales_raw, rots_raw

struct ParamSet { float* t[NT]; };      // same order as EgsGaussianParams
struct Widths { int w[NT]; };

// ---- statistics ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_density_accum(int n, const float2* __restrict__ dus,
                                                       const uint8_t* __restrict__ visible, int first,
                                                       float* __restrict__ acc, int32_t* __restrict__ cnt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float2 d = dus[i];
  const float g = sqrtf(d.x * d.x + d.y * d.y);     // torch.norm(dloss_dus, dim=-1)
  const int vis = visible[i] != 0;
  if (first) {                                      // gsmodel.py:222-224: every row, mask as counts
    acc[i] = g;
    cnt[i] = vis;
  } else if (vis) {                                 // gsmodel.py:226-227
    acc[i] += g;
    cnt[i] += 1;
  }
}

// ---- classification -----------------------------------------------------------------------------
// cls: 0 pruned, 1 survivor, 2 survivor + clone, 3 survivor + split
__device__ __forceinline__ uint32_t pack3(int cls) {   // (keep, clone, split) counters in 10-bit fields
  return (cls >= 1 ? 1u : 0u) | (cls == 2 ? 1u << 10 : 0u) | (cls == 3 ? 1u << 20 : 0u);
}

__global__ __launch_bounds__(DB) void k_densify_classify(int n, const float* __restrict__ alphas_raw,
                                                         const float* __restrict__ scales_raw,
                                                         const float* __restrict__ acc,
                                                         const int32_t* __restrict__ cnt, float alpha_thr_raw,
                                                         float big_thr_raw, float grad_thr, float scale_thr,
                                                         uint8_t* __restrict__ cls_out,
                                                         uint32_t* __restrict__ blocksum /* [3][nblocks] */,
                                                         int nblocks) {
  __shared__ uint32_t sm[4];
  const int i = blockIdx.x * DB + threadIdx.x;
  int cls = 0;
  if (i < n) {
    const float a = alphas_raw[i];
    const float smax = fmaxf(scales_raw[3 * i], fmaxf(scales_raw[3 * i + 1], scales_raw[3 * i + 2]));
    const bool prune = (a < alpha_thr_raw) || (smax > big_thr_raw);      // gsmodel.py:234-236
    if (!prune) {
      float g = acc[i] / (float)cnt[i];                                  // gsmodel.py:241-242
      if (g != g) g = 0.f;
      const bool by_grad = g >= grad_thr;
      const bool small = expf(smax) <= scale_thr;                        // gsmodel.py:252
      cls = by_grad ? (small ? 2 : 3) : 1;
    }
    cls_out[i] = (uint8_t)cls;
  }
  uint32_t total;
  (void)block256_exclusive_scan(pack3(cls), sm, &total);
  if (threadIdx.x == 0) {
    blocksum[blockIdx.x] = total & 1023u;
    blocksum[nblocks + blockIdx.x] = (total >> 10) & 1023u;
    blocksum[2 * nblocks + blockIdx.x] = total >> 20;
  }
}

// exclusive scan of the three per-block count arrays (one workgroup; nblocks <= a few thousand)
__global__ __launch_bounds__(256) void k_densify_scan(int nblocks, uint32_t* __restrict__ blocksum,
                                                      int32_t* __restrict__ totals, int n) {
  __shared__ uint32_t sm[4];
  uint32_t sums[3];
  for (int a = 0; a < 3; ++a) {
    uint32_t* arr = blocksum + (size_t)a * nblocks;
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += 256) {
      const int i = base + threadIdx.x;
      const uint32_t v = i < nblocks ? arr[i] : 0u;
      uint32_t tot;
      const uint32_t ex = block256_exclusive_scan(v, sm, &tot);
      if (i < nblocks) arr[i] = carry + ex;
      carry += tot;
    }
    sums[a] = carry;
  }
  if (threadIdx.x == 0) {
    totals[0] = (int32_t)sums[0];
    totals[1] = (int32_t)sums[1];
    totals[2] = (int32_t)sums[2];
    totals[3] = n - (int32_t)sums[0];
  }
}

// ---- counter-based normals: bit-compatible with easygaussiansplatting_amd/scene.py ------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double uniform01(uint64_t seed, uint64_t stream, uint64_t e) {
  const uint64_t key = splitmix64(seed * 0x100000001B3ull + stream);
  uint64_t bits = splitmix64(e ^ key);
  bits = splitmix64(bits + key);
  return (double)(bits >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ float unit_normal(uint64_t seed, uint64_t stream, uint64_t e) {
  double u1 = uniform01(seed, 2 * stream + 1000, e);
  const double u2 = uniform01(seed, 2 * stream + 1001, e);
  u1 = u1 > 1e-300 ? u1 : 1e-300;
  return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2));
}

// ---- compaction ---------------------------------------------------------------------------------
struct ApplyArgs {
  ParamSet in, in_m, in_v, out, out_m, out_v;   // in_m.t[0] == nullptr: optimizer has no state yet
  Widths w;
  int n, nblocks, n_keep, n_clone;
  const uint8_t* cls;
  const uint32_t* blockoff;                     // [3][nblocks] exclusive
  const float* unit_noise;                      // [n][3] or nullptr -> generator
  uint64_t seed, round;
};

__global__ __launch_bounds__(DB) void k_densify_apply(ApplyArgs A) {
  __shared__ uint32_t sm[4];
  __shared__ int s_keep[DB], s_new[DB];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int i = b * DB + tid;
  const int cls = i < A.n ? (int)A.cls[i] : 0;
  const uint32_t ex = block256_exclusive_scan(pack3(cls), sm, nullptr);
  const int keep_dst = cls >= 1 ? (int)(A.blockoff[b] + (ex & 1023u)) : -1;
  int new_dst = -1;
  if (cls == 2) new_dst = A.n_keep + (int)(A.blockoff[A.nblocks + b] + ((ex >> 10) & 1023u));
  if (cls == 3) new_dst = A.n_keep + A.n_clone + (int)(A.blockoff[2 * A.nblocks + b] + (ex >> 20));
  s_keep[tid] = keep_dst;
  s_new[tid] = new_dst;

  // appended row: activated -> (perturbed) -> back to raw, gsmodel.py:257-288
  if (new_dst >= 0) {
    const float ar = A.in.t[3][i];
    const float al = 1.f / (1.f + expf(-ar));                                    // get_alphas
    A.out.t[3][new_dst] = logf(al / (1.f - al));                                 // get_alphas_raw
    float sc[3], q[4], pw[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { sc[c] = expf(A.in.t[4][3 * i + c]); pw[c] = A.in.t[0][3 * i + c]; }
#pragma unroll
    for (int c = 0; c < 4; ++c) q[c] = A.in.t[5][4 * i + c];
    const float nq = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
#pragma unroll
    for (int c = 0; c < 4; ++c) q[c] /= nq;                                      // get_rots
    if (cls == 3) {
      float v[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float z = A.unit_noise ? A.unit_noise[3 * i + c] : unit_normal(A.seed, A.round, 3ull * i + c);
        v[c] = sc[c] * z;                                                        // torch.normal(0, scales)
      }
      // rotate_vector_by_quaternion (utils.py:46-54); it normalises q once more
      const float n2 = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
      const float s = q[0] / n2, ux = q[1] / n2, uy = q[2] / n2, uz = q[3] / n2;
      const float uv = ux * v[0] + uy * v[1] + uz * v[2], uu = ux * ux + uy * uy + uz * uz;
      const float cx = uy * v[2] - uz * v[1], cy = uz * v[0] - ux * v[2], cz = ux * v[1] - uy * v[0];
      const float k = s * s - uu;
      pw[0] += 2.f * ux * uv + v[0] * k + 2.f * cx * s;
      pw[1] += 2.f * uy * uv + v[1] * k + 2.f * cy * s;
      pw[2] += 2.f * uz * uv + v[2] * k + 2.f * cz * s;
#pragma unroll
      for (int c = 0; c < 3; ++c) sc[c] *= 0.6f;                                 // gsmodel.py:279
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      A.out.t[0][3 * (size_t)new_dst + c] = pw[c];
      A.out.t[4][3 * (size_t)new_dst + c] = logf(sc[c]);                         // get_scales_raw
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) A.out.t[5][4 * (size_t)new_dst + c] = q[c];
  }
  __syncthreads();

  // cooperative row moves: reads are one contiguous span of DB*w floats per tensor, writes are
  // contiguous runs (survivors stay in order)
  const bool has_state = A.in_m.t[0] != nullptr;
  const int rows = min(DB, A.n - b * DB);
#pragma unroll 1
  for (int t = 0; t < NT; ++t) {
    const int w = A.w.w[t];
    const size_t base = (size_t)b * DB * w;
    const bool copied = (t == 1 || t == 2);       // SH rows are copied to the appended row as they are
    const float* __restrict__ src = A.in.t[t] + base;
    const float* __restrict__ srcm = has_state ? A.in_m.t[t] + base : nullptr;
    const float* __restrict__ srcv = has_state ? A.in_v.t[t] + base : nullptr;
    for (int e = tid; e < rows * w; e += DB) {
      const int r = e / w, c = e - r * w;
      const int kd = s_keep[r], nd = s_new[r];
      if (kd < 0) continue;
      const float x = src[e];
      A.out.t[t][(size_t)kd * w + c] = x;
      if (has_state) {
        A.out_m.t[t][(size_t)kd * w + c] = srcm[e];
        A.out_v.t[t][(size_t)kd * w + c] = srcv[e];
      }
      if (nd >= 0) {
        if (copied) A.out.t[t][(size_t)nd * w + c] = x;
        if (has_state) {                          // update_params: zero moments for appended rows
          A.out_m.t[t][(size_t)nd * w + c] = 0.f;
          A.out_v.t[t][(size_t)nd * w + c] = 0.f;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_reset_alpha(int n, float raw_val, float* __restrict__ alphas_raw,
                                                     float* __restrict__ m, float* __restrict__ v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float a = alphas_raw[i];
  if (a > raw_val) alphas_raw[i] = raw_val;       // gsmodel.py:320-323
  if (m) m[i] = 0.f;                              // gsmodel.py:327-328
  if (v) v[i] = 0.f;
}

// ---- fused Adam ---------------------------------------------------------------------------------
constexpr int ADAM_MAX_GROUPS = 8;
constexpr int ADAM_PER_BLOCK = 256 * 4;           // floats per workgroup
struct AdamGroupDev {
  float* p; const float* g; float* m; float* v;
  int64_t count;
  float step_size, sqrt_bc2;                      // lr / (1 - b1^t), sqrt(1 - b2^t)
  int first_block;
};
struct AdamArgs {
  AdamGroupDev g[ADAM_MAX_GROUPS];
  int n_groups;
  float beta1, beta2, omb1, omb2, eps;            // omb = 1 - beta rounded from double, as torch does
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& A, float step_size,
                                      float sqrt_bc2) {
  // (no contraction: torch's kernels round every product, and k_adam / k_adam_sh_factored must agree to the bit
  // wherever the compiler inlines this)
#pragma clang fp contract(off)
  m = m + (g - m) * A.omb1;                       // exp_avg.lerp_(grad, 1 - beta1)
  v = v * A.beta2 + g * g * A.omb2;               // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
  const float denom = sqrtf(v) / sqrt_bc2 + A.eps;   // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
  p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void k_adam(AdamArgs A) {
  int gi = 0;
#pragma unroll
  for (int k = 1; k < ADAM_MAX_GROUPS; ++k)
    if (k < A.n_groups && (int)blockIdx.x >= A.g[k].first_block) gi = k;
  const AdamGroupDev G = A.g[gi];
  const int64_t e0 = ((int64_t)(blockIdx.x - G.first_block) * 256 + threadIdx.x) * 4;
  if (e0 >= G.count) return;
  if (e0 + 4 <= G.count) {                        // tensors come from the torch allocator: 16-B aligned
    float4 p = *(float4*)(G.p + e0), m = *(float4*)(G.m + e0), v = *(float4*)(G.v + e0);
    const float4 g = *(const float4*)(G.g + e0);
    adam1(p.x, g.x, m.x, v.x, A, G.step_size, G.sqrt_bc2);
    adam1(p.y, g.y, m.y, v.y, A, G.step_size, G.sqrt_bc2);
    adam1(p.z, g.z, m.z, v.z, A, G.step_size, G.sqrt_bc2);
    adam1(p.w, g.w, m.w, v.w, A, G.step_size, G.sqrt_bc2);
    *(float4*)(G.p + e0) = p; *(float4*)(G.m + e0) = m; *(float4*)(G.v + e0) = v;
  } else {
    for (int64_t e = e0; e < G.count; ++e) {
      float p = G.p[e], m = G.m[e], v = G.v[e];
      adam1(p, G.g[e], m, v, A, G.step_size, G.sqrt_bc2);
      G.p[e] = p; G.m[e] = m; G.v[e] = v;
    }
  }
}

// ---- Adam on the SH coefficients straight from the factored gradient -------------------------------------------
// A step that kept its SH gradient factored (EGS_BWD_FACTORED_SH: dL/dcolour [N][3] per view, egs_hip.h) never needs
// the 4 sh_dim-byte rows in HBM: one wave of the workgroup forms the rows of 64 Gaussians in LDS -- the arithmetic of
// k_sh_grad_views, scale * sum_v dL/dcolour_v (x) basis(pw - twc_v) -- and all four waves run torch's Adam update over
// the 64 x K contiguous elements of param / exp_avg / exp_avg_sq (7 x 4 B per element instead of 8 plus the rows'
// write: 1344 + 12 V bytes per Gaussian at degree 3 where k_sh_grad_views + k_adam move 1728 + 12 V).
constexpr int ASH_ROWS = 64;
struct AdamShTensor {
  float *p, *m, *v;                               // [N][width]
  float step_size, sqrt_bc2;
  int width;                                      // floats per Gaussian; 0: tensor absent
  int col0;                                       // first column of the K-wide row it holds (0: low / whole, 3: high)
};
template <int NC>
__global__ __launch_bounds__(256) void k_adam_sh_factored(int n, int views, const float* __restrict__ pws,
                                                          const float* __restrict__ rows, int64_t stride, float scale,
                                                          AdamShTensor T0, AdamShTensor T1, AdamArgs A) {
  constexpr int K = 3 * NC;
  __shared__ float g[ASH_ROWS * K];
  const int base = blockIdx.x * ASH_ROWS;
  const int here = min(ASH_ROWS, n - base);
  if (threadIdx.x < ASH_ROWS) {
    float gsh[K];
#pragma unroll
    for (int k = 0; k < K; ++k) gsh[k] = 0.f;
    const int i = base + threadIdx.x;
    if (i < n) {
      const f3 pw = ld3(pws + 3 * (size_t)i);
      for (int v = 0; v < views; ++v) {
        const float* row = rows + (size_t)v * stride;
        const f3 gc = ld3(row + 3 * (size_t)i);
        if (gc.x == 0.f && gc.y == 0.f && gc.z == 0.f) continue;
        const ShDir<NC> d = sh_basis_f<NC>(pw, row + 3 * (size_t)n);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          gsh[3 * c] = __builtin_fmaf(gc.x, d.B[c], gsh[3 * c]);
          gsh[3 * c + 1] = __builtin_fmaf(gc.y, d.B[c], gsh[3 * c + 1]);
          gsh[3 * c + 2] = __builtin_fmaf(gc.z, d.B[c], gsh[3 * c + 2]);
        }
      }
#pragma unroll
      for (int k = 0; k < K; ++k) gsh[k] *= scale;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) g[threadIdx.x * K + k] = gsh[k];
  }
  __syncthreads();
  auto update = [&](const AdamShTensor& T) {
    if (T.width == 0) return;
    const int cnt = here * T.width;                         // this workgroup's contiguous elements of the tensor
    const size_t off = (size_t)base * T.width;              // (64 rows: every offset is a multiple of 16 bytes)
    for (int e0 = threadIdx.x * 4; e0 < cnt; e0 += 1024) {
      float gg[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = min(e0 + q, cnt - 1);
        gg[q] = g[(e / T.width) * K + T.col0 + e % T.width];
      }
      if (e0 + 4 <= cnt) {
        float4 p = *(float4*)(T.p + off + e0), m = *(float4*)(T.m + off + e0), v = *(float4*)(T.v + off + e0);
        adam1(p.x, gg[0], m.x, v.x, A, T.step_size, T.sqrt_bc2);
        adam1(p.y, gg[1], m.y, v.y, A, T.step_size, T.sqrt_bc2);
        adam1(p.z, gg[2], m.z, v.z, A, T.step_size, T.sqrt_bc2);
        adam1(p.w, gg[3], m.w, v.w, A, T.step_size, T.sqrt_bc2);
        *(float4*)(T.p + off + e0) = p; *(float4*)(T.m + off + e0) = m; *(float4*)(T.v + off + e0) = v;
      } else {
        for (int q = 0; e0 + q < cnt; ++q) {
          float p = T.p[off + e0 + q], m = T.m[off + e0 + q], v = T.v[off + e0 + q];
          adam1(p, gg[q], m, v, A, T.step_size, T.sqrt_bc2);
          T.p[off + e0 + q] = p; T.m[off + e0 + q] = m; T.v[off + e0 + q] = v;
        }
      }
    }
  };
  update(T0);
  update(T1);
}

static ParamSet to_set(const EgsGaussianParams* p) {
  ParamSet s;
  if (!p) { for (int t = 0; t < NT; ++t) s.t[t] = nullptr; return s; }
  s.t[0] = p->pws; s.t[1] = p->low_shs; s.t[2] = p->high_shs;
  s.t[3] = p->alphas_raw; s.t[4] = p->scales_raw; s.t[5] = p->rots_raw;
  return s;
}
static bool all_set(const ParamSet& s) {
  for (int t = 0; t < NT; ++t) if (!s.t[t]) return false;
  return true;
}

}  // namespace egs

using namespace egs;

extern "C" int egs_density_accumulate(int n, const float* dloss_dus, const uint8_t* visible, int first,
                                      float* grad_accum, int32_t* count, void* stream) {
  EGS_CHECK_ARG(n >= 0);
  if (n == 0) return 0;
  EGS_CHECK_ARG(dloss_dus && visible && grad_accum && count);
  hipStream_t s = (hipStream_t)stream;
  EGS_LAUNCH("k_density_accum", k_density_accum, dim3(div_up(n, 256)), dim3(256), s, n, (const float2*)dloss_dus,
             visible, first, grad_accum, count);
  EGS_LAUNCH_OK();
  return 0;
}

extern "C" size_t egs_densify_ws_bytes(int n) {
  const size_t nb = (size_t)div_up(n > 0 ? n : 1, DB);
  return align_up(3 * nb * sizeof(uint32_t), 256) + 256;
}

extern "C" int egs_densify_plan(int n, const float* alphas_raw, const float* scales_raw, const float* grad_accum,
                                const int32_t* count, float alpha_thr_raw, float big_thr_raw, float grad_thr,
                                float scale_thr, uint8_t* cls, void* ws, size_t ws_bytes, int32_t* totals,
                                void* stream) {
  EGS_CHECK_ARG(n >= 0 && totals);
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    EGS_HIP(hipMemsetAsync(totals, 0, 4 * sizeof(int32_t), s));
    return 0;
  }
  EGS_CHECK_ARG(alphas_raw && scales_raw && grad_accum && count && cls && ws);
  EGS_CHECK_ARG(ws_bytes >= egs_densify_ws_bytes(n));
  const int nb = div_up(n, DB);
  uint32_t* blocksum = (uint32_t*)ws;
  EGS_LAUNCH("k_densify_classify", k_densify_classify, dim3(nb), dim3(DB), s, n, alphas_raw, scales_raw, grad_accum,
             count, alpha_thr_raw, big_thr_raw, grad_thr, scale_thr, cls, blocksum, nb);
  EGS_LAUNCH_OK();
  EGS_LAUNCH("k_densify_scan", k_densify_scan, dim3(1), dim3(256), s, nb, blocksum, totals, n);
  EGS_LAUNCH_OK();
  return 0;
}

extern "C" int egs_densify_apply(int n, int n_keep, int n_clone, int n_split, int high_sh_width, const uint8_t* cls,
                                 const void* ws, const EgsGaussianParams* in, const EgsGaussianParams* in_exp_avg,
                                 const EgsGaussianParams* in_exp_avg_sq, const EgsGaussianParams* out,
                                 const EgsGaussianParams* out_exp_avg, const EgsGaussianParams* out_exp_avg_sq,
                                 const float* unit_noise, uint64_t seed, uint64_t round, void* stream) {
  EGS_CHECK_ARG(n >= 0 && n_keep >= 0 && n_clone >= 0 && n_split >= 0 && n_keep <= n && n_clone + n_split <= n_keep);
  EGS_CHECK_ARG(high_sh_width >= 0);
  if (n == 0) return 0;
  EGS_CHECK_ARG(cls && ws && in && out);
  ApplyArgs A;
  A.in = to_set(in); A.out = to_set(out);
  A.in_m = to_set(in_exp_avg); A.in_v = to_set(in_exp_avg_sq);
  A.out_m = to_set(out_exp_avg); A.out_v = to_set(out_exp_avg_sq);
  const bool has_state = in_exp_avg != nullptr;
  EGS_CHECK_ARG(has_state == (in_exp_avg_sq != nullptr) && has_state == (out_exp_avg != nullptr) &&
                has_state == (out_exp_avg_sq != nullptr));
  const int widths[NT] = {3, 3, high_sh_width, 1, 3, 4};
  for (int t = 0; t < NT; ++t) {
    A.w.w[t] = widths[t];
    if (widths[t] == 0) continue;
    EGS_CHECK_ARG(A.in.t[t] && (A.out.t[t] || n_keep + n_clone + n_split == 0));
    if (has_state) EGS_CHECK_ARG(A.in_m.t[t] && A.in_v.t[t]);
  }
  if (!has_state) A.in_m.t[0] = nullptr;
  A.n = n; A.nblocks = div_up(n, DB); A.n_keep = n_keep; A.n_clone = n_clone;
  A.cls = cls; A.blockoff = (const uint32_t*)ws; A.unit_noise = unit_noise; A.seed = seed; A.round = round;
  hipStream_t s = (hipStream_t)stream;
  EGS_LAUNCH("k_densify_apply", k_densify_apply, dim3(A.nblocks), dim3(DB), s, A);
  EGS_LAUNCH_OK();
  return 0;
}

extern "C" int egs_reset_alpha(int n, float raw_val, float* alphas_raw, float* exp_avg, float* exp_avg_sq,
                               void* stream) {
  EGS_CHECK_ARG(n >= 0);
  if (n == 0) return 0;
  EGS_CHECK_ARG(alphas_raw);
  hipStream_t s = (hipStream_t)stream;
  EGS_LAUNCH("k_reset_alpha", k_reset_alpha, dim3(div_up(n, 256)), dim3(256), s, n, raw_val, alphas_raw, exp_avg,
             exp_avg_sq);
  EGS_LAUNCH_OK();
  return 0;
}

extern "C" int egs_adam_step(int n_groups, const EgsAdamGroup* groups, double beta1, double beta2, double eps,
                             void* stream) {
  EGS_CHECK_ARG(n_groups >= 0 && n_groups <= ADAM_MAX_GROUPS);
  EGS_CHECK_ARG(n_groups == 0 || groups);
  AdamArgs A;
  A.n_groups = 0; A.beta1 = (float)beta1; A.beta2 = (float)beta2; A.eps = (float)eps;
  A.omb1 = (float)(1.0 - beta1); A.omb2 = (float)(1.0 - beta2);
  int64_t blocks = 0;
  for (int k = 0; k < n_groups; ++k) {
    const EgsAdamGroup& g = groups[k];
    EGS_CHECK_ARG(g.count >= 0 && g.step >= 1);
    if (g.count == 0) continue;
    EGS_CHECK_ARG(g.param && g.grad && g.exp_avg && g.exp_avg_sq);
    AdamGroupDev& d = A.g[A.n_groups++];
    d.p = g.param; d.g = g.grad; d.m = g.exp_avg; d.v = g.exp_avg_sq; d.count = g.count;
    const double bc1 = 1.0 - pow(beta1, (double)g.step), bc2 = 1.0 - pow(beta2, (double)g.step);
    d.step_size = (float)((double)g.lr / bc1);
    d.sqrt_bc2 = (float)sqrt(bc2);
    d.first_block = (int)blocks;
    blocks += (g.count + ADAM_PER_BLOCK - 1) / ADAM_PER_BLOCK;
  }
  if (blocks == 0) return 0;
  EGS_CHECK_ARG(blocks < (int64_t)1 << 31);
  hipStream_t s = (hipStream_t)stream;
  EGS_LAUNCH("k_adam", k_adam, dim3((unsigned)blocks), dim3(256), s, A);
  EGS_LAUNCH_OK();
  return 0;
}

extern "C" int egs_adam_sh_factored(int n, int sh_dim, int views, const float* pws, const float* rows,
                                    int64_t row_stride, float scale, const EgsAdamGroup* low,
                                    const EgsAdamGroup* high, double beta1, double beta2, double eps, void* stream) {
  EGS_CHECK_ARG(n >= 0 && views >= 0 && low);
  EGS_CHECK_ARG(sh_dim == 3 || sh_dim == 12 || sh_dim == 27 || sh_dim == 48);
  if (n == 0) return 0;
  EGS_CHECK_ARG(pws && (views == 0 || rows) && row_stride >= 3 * (int64_t)n + 3);
  const bool raw = high != nullptr && sh_dim > 3;
  AdamArgs A;
  A.n_groups = 0; A.beta1 = (float)beta1; A.beta2 = (float)beta2; A.eps = (float)eps;
  A.omb1 = (float)(1.0 - beta1); A.omb2 = (float)(1.0 - beta2);
  auto tensor = [&](const EgsAdamGroup* g, int width, int col0, AdamShTensor* t) -> bool {
    t->width = 0; t->col0 = col0; t->p = t->m = t->v = nullptr; t->step_size = 0.f; t->sqrt_bc2 = 1.f;
    if (!g) return true;
    if (!(g->param && g->exp_avg && g->exp_avg_sq && g->step >= 1 && g->count == (int64_t)n * width)) return false;
    if ((((uintptr_t)g->param | (uintptr_t)g->exp_avg | (uintptr_t)g->exp_avg_sq) & 15) != 0) return false;
    t->p = g->param; t->m = g->exp_avg; t->v = g->exp_avg_sq; t->width = width;
    const double bc1 = 1.0 - pow(beta1, (double)g->step), bc2 = 1.0 - pow(beta2, (double)g->step);
    t->step_size = (float)((double)g->lr / bc1);
    t->sqrt_bc2 = (float)sqrt(bc2);
    return true;
  };
  AdamShTensor T0, T1;
  EGS_CHECK_ARG(tensor(low, raw ? 3 : sh_dim, 0, &T0));
  EGS_CHECK_ARG(tensor(raw ? high : nullptr, sh_dim - 3, 3, &T1));
  dim3 g(div_up(n, ASH_ROWS)), b(256);
  hipStream_t s = (hipStream_t)stream;
#define EGS_ASH(NC) \
  EGS_LAUNCH("k_adam_sh_factored", (k_adam_sh_factored<NC>), g, b, s, n, views, pws, rows, row_stride, scale, T0, T1, A)
  switch (sh_dim) {
    case 3: EGS_ASH(1); break;
    case 12: EGS_ASH(4); break;
    case 27: EGS_ASH(9); break;
    default: EGS_ASH(16); break;
  }
#undef EGS_ASH
  EGS_LAUNCH_OK();
  return 0;
}
