// Per-Gaussian stages of the splatting pipeline for gfx950 (wave64):
//   * the five ops of the reference surface -- project / cov3d / cov2d / sh2color /
//     inv_cov2d (+ optional Jacobians stored to HBM, as gsplatcu does) -- and the chain
//     rule kernel that consumes those Jacobians;
//   * the fused training path: k_preprocess_fwd (all five stages in
//     one pass, 236 B read + 44 B written per Gaussian, no Jacobians materialised) and
//     k_preprocess_bwd (re-derives the Jacobians in registers and applies the chain rule:
//     236 + 48 B read, 244 B written, instead of 436 B of Jacobians written then re-read).
//
// The math lives in egs_gaussian_math.h (one source of truth for both paths).
// All kernels: one Gaussian per lane, 256-thread workgroups (4 waves), >= 3900
// workgroups at N = 1 M so all 256 CUs are covered many times over.  They are
// HBM-bound streaming kernels; algorithmic bytes per Gaussian are listed in DESIGN.md.
#include "egs_gaussian_math.h"

namespace egs {

// ---- row stores of the seven-op kernels ---------------------------------------------------------------------
// The reference's ops hand back per-Gaussian rows of K floats (K = 3 ... 24: values and Jacobians, 832 B per
// Gaussian with calc_J).  A lane that stores its own row issues K dword (or K/4 dwordx4) stores whose 64 lanes are
// 4K bytes apart: every instruction touches 64 different lines and fills 4 or 16 bytes of each -- the AoS store
// pattern that kept these kernels at 3.0-3.7 TB/s.  The workgroup's 256 rows are ONE contiguous, 16-B aligned span
// of the output (256 * K * 4 bytes), so the rows are deposited in LDS -- row stride padded to an odd number of words
// (of 16-B units for K % 4 == 0): conflict-free -- and the span leaves as coalesced dwordx4 stores, full lines.
// EVERY row is written, culled Gaussians as zeros: the callers allocate with torch.empty, not torch.zeros (the
// reference's zero-filled outputs, gausplat.cu:170-178 etc., cost 528 B per Gaussian and step of pure fill here).
template <int K>
struct RowsOut {
  static constexpr bool V4 = (K % 4 == 0);
  static constexpr bool DIRECT = (K <= 2);
  static constexpr int STRIDE = V4 ? 4 * ((K / 4 + 1) | 1) : (K | 1);   // floats
  static constexpr int LDS_FLOATS = DIRECT ? 1 : 256 * STRIDE;
};
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// all 256 threads of the workgroup call this (barriers inside); `row` of lanes past n is ignored
template <int K>
__device__ __forceinline__ void rows_out(const float* row, float* __restrict__ dst, int n, int base, float* lds) {
  using RO = RowsOut<K>;
  const int tid = threadIdx.x;
  const int rows = min(256, n - base);
  if constexpr (RO::DIRECT) {
    if (tid < rows) {
      if constexpr (K == 2) reinterpret_cast<float2*>(dst)[base + tid] = make_float2(row[0], row[1]);
      else dst[base + tid] = row[0];
    }
  } else {
    __syncthreads();   // the previous user of `lds` is done with it
    if constexpr (RO::V4) {
#pragma unroll
      for (int j = 0; j < K / 4; ++j)
        *reinterpret_cast<float4*>(lds + tid * RO::STRIDE + 4 * j) =
            make_float4(row[4 * j], row[4 * j + 1], row[4 * j + 2], row[4 * j + 3]);
    } else {
#pragma unroll
      for (int j = 0; j < K; ++j) lds[tid * RO::STRIDE + j] = row[j];
    }
    __syncthreads();
    float4* __restrict__ d4 = reinterpret_cast<float4*>(dst + (size_t)K * base);
    const int total = rows * K, nq = total >> 2;
#pragma unroll
    for (int j = 0; j < (K + 3) / 4; ++j) {
      const int f = tid + 256 * j;
      if (f < nq) {
        if constexpr (RO::V4) {
          const int r = f / (K / 4), c = f - r * (K / 4);
          d4[f] = *reinterpret_cast<const float4*>(lds + r * RO::STRIDE + 4 * c);
        } else {
          int r = (4 * f) / K, c = 4 * f - r * K;
          float v[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            v[t] = lds[r * RO::STRIDE + c];
            if (++c == K) { c = 0; ++r; }
          }
          d4[f] = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    if constexpr (!RO::V4) {   // <= 3 floats when the last workgroup's row count is not a multiple of four
      const int f = 4 * nq + tid;
      if (f < total) {
        const int r = f / K, c = f - r * K;
        dst[(size_t)K * base + f] = lds[r * RO::STRIDE + c];
      }
    }
  }
}

// ---- project                                          (reference kernel.cu:553-617)
// (60 B per Gaussian: too little to pay for a trip through LDS -- measured 20 us direct, 23 us staged; every row is
// still written, culled Gaussians as zeros)
__global__ __launch_bounds__(256) void k_project(int n, const float* __restrict__ pws,
                                                 const float* __restrict__ Rcw,
                                                 const float* __restrict__ tcw, float fx, float fy,
                                                 float cx, float cy, int near_cull,
                                                 float* __restrict__ us, float* __restrict__ pcs,
                                                 float* __restrict__ depths,
                                                 float* __restrict__ du_dpcs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Proj P = project_f(ld3(pws + 3 * (size_t)i), Rcw, tcw, fx, fy, cx, cy);
  const bool cull = near_cull && P.pc.z < EGS_MIN_DEPTH;
  float2* J2 = du_dpcs ? reinterpret_cast<float2*>(du_dpcs + 6 * (size_t)i) : nullptr;   // 24-B rows: 8-B aligned
  if (cull) {   // everything but the marker reads 0, like the reference's zero-filled outputs
    reinterpret_cast<float2*>(us)[i] = make_float2(0.f, 0.f);
    st3(pcs + 3 * (size_t)i, {0.f, 0.f, 0.f});
    depths[i] = EGS_BAD_MARKER;
    if (J2) { J2[0] = make_float2(0.f, 0.f); J2[1] = make_float2(0.f, 0.f); J2[2] = make_float2(0.f, 0.f); }
    return;
  }
  reinterpret_cast<float2*>(us)[i] = make_float2(P.u0, P.u1);
  st3(pcs + 3 * (size_t)i, P.pc);
  depths[i] = P.pc.z;
  if (J2) {
    float j00, j02, j11, j12;
    project_jac(P, fx, fy, j00, j02, j11, j12);
    J2[0] = make_float2(j00, 0.f); J2[1] = make_float2(j02, 0.f); J2[2] = make_float2(j11, j12);
  }
}

// ---- cov3d                                            (reference kernel.cu:326-423)
__global__ __launch_bounds__(256) void k_cov3d(int n, const float* __restrict__ rots,
                                               const float* __restrict__ scales,
                                               const float* __restrict__ depths, int near_cull,
                                               float* __restrict__ cov3ds,
                                               float* __restrict__ dcov3d_drots,
                                               float* __restrict__ dcov3d_dscales) {
  __shared__ float stage[RowsOut<24>::LDS_FLOATS];
  const int base = blockIdx.x * 256, i = base + threadIdx.x;
  const bool jac = dcov3d_drots && dcov3d_dscales;
  float c6[6], dq[24], ds[18];
#pragma unroll
  for (int k = 0; k < 6; ++k) c6[k] = 0.f;
#pragma unroll
  for (int k = 0; k < 24; ++k) dq[k] = 0.f;
#pragma unroll
  for (int k = 0; k < 18; ++k) ds[k] = 0.f;
  if (i < n && !(near_cull && depths[i] < EGS_MIN_DEPTH)) {
    const float4 q = *reinterpret_cast<const float4*>(rots + 4 * (size_t)i);  // 16-B aligned rows
    const f3 s = ld3(scales + 3 * (size_t)i);
    const Cov3 c = cov3d_f(q, s);
#pragma unroll
    for (int k = 0; k < 6; ++k) c6[k] = c.c[k];
    if (jac) cov3d_jac(c, q, s, dq, ds);
  }
  rows_out<6>(c6, cov3ds, n, base, stage);
  if (jac) {
    rows_out<24>(dq, dcov3d_drots, n, base, stage);
    rows_out<18>(ds, dcov3d_dscales, n, base, stage);
  }
}

// ---- cov2d                                            (reference kernel.cu:425-551)
__global__ __launch_bounds__(256) void k_cov2d(int n, const float* __restrict__ cov3ds,
                                               const float* __restrict__ pcs,
                                               const float* __restrict__ Rcw,
                                               const float* __restrict__ depths, float fx, float fy,
                                               float limx, float limy, int clamp_fov, int near_cull,
                                               float* __restrict__ cov2ds,
                                               float* __restrict__ dcov2d_dcov3ds,
                                               float* __restrict__ dcov2d_dpcs) {
  __shared__ float stage[RowsOut<18>::LDS_FLOATS];
  const int base = blockIdx.x * 256, i = base + threadIdx.x;
  const bool jac = dcov2d_dcov3ds && dcov2d_dpcs;
  float c3[3] = {0.f, 0.f, 0.f}, J3[18], Jp[9];
#pragma unroll
  for (int k = 0; k < 18; ++k) J3[k] = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) Jp[k] = 0.f;
  if (i < n && !(near_cull && depths[i] < EGS_MIN_DEPTH)) {
    const f3 pc = ld3(pcs + 3 * (size_t)i);
    float cv[6];
    const float2* cr = reinterpret_cast<const float2*>(cov3ds + 6 * (size_t)i);   // 24-B rows: 8-B aligned
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float2 v = cr[k]; cv[2 * k] = v.x; cv[2 * k + 1] = v.y; }
    const Cov2 c = cov2d_f(cv, pc, Rcw, fx, fy, limx, limy, clamp_fov);
    c3[0] = c.c[0]; c3[1] = c.c[1]; c3[2] = c.c[2];
    if (jac) cov2d_jac(c, pc.z, Rcw, fx, fy, J3, Jp);
  }
  rows_out<3>(c3, cov2ds, n, base, stage);
  if (jac) {
    rows_out<18>(J3, dcov2d_dcov3ds, n, base, stage);
    rows_out<9>(Jp, dcov2d_dpcs, n, base, stage);
  }
}

// ---- sh2color                                         (reference kernel.cu:619-807)
// A/B knobs: colour and dcolor/dpw in ONE pass over the SH row (sh_color_and_jac_dpw) in k_sh2color / in
// k_preprocess_fwd<.., JW>.  Measured (round 4, same-box A/B): the training instance of k_preprocess_fwd drops from 82
// to 60 VGPRs (5 -> 8 waves per SIMD) -- and gets SLOWER, 99 -> 110 us: more resident waves mean more SH rows competing
// for the 32-KB L1 between the twelve row loads of a lane.  Off there.
#ifndef EGS_SH_FUSED_JAC
#define EGS_SH_FUSED_JAC 1
#endif
#ifndef EGS_SH_FUSED_JAC_PRE
#define EGS_SH_FUSED_JAC_PRE 0
#endif

#ifndef EGS_SH2COLOR_WAVES     // A/B knob: minimum waves per SIMD of k_sh2color (106 VGPRs = 4 as compiled freely)
#define EGS_SH2COLOR_WAVES 1
#endif
template <int NC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(EGS_SH2COLOR_WAVES, 8))) void k_sh2color(int n, const float* __restrict__ shs,
                                                  const float* __restrict__ pws,
                                                  const float* __restrict__ twc,
                                                  float* __restrict__ colors,
                                                  float* __restrict__ dcolor_dshs,
                                                  float* __restrict__ dcolor_dpws) {
  __shared__ float stage[cmax(RowsOut<NC>::LDS_FLOATS, RowsOut<9>::LDS_FLOATS)];
  const int base = blockIdx.x * 256, i = base + threadIdx.x;
  constexpr int K = 3 * NC;
  const bool jac = dcolor_dshs && dcolor_dpws;
  float col[3] = {0.f, 0.f, 0.f}, jp[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) jp[j] = 0.f;
#if EGS_SH_FUSED_JAC
  // Round 4: the basis row leaves FIRST (it depends on the direction alone), then ONE pass over the coefficients sums
  // colour and dcolor/ddir, re-evaluating each basis value where it is used: neither the 16 basis values nor the
  // consumed part of the SH row stay in registers (110 -> see DESIGN 3.1).
  float sh[K];
  ShDir<NC> d;
#pragma unroll
  for (int c = 0; c < NC; ++c) d.B[c] = 0.f;
  d.d0 = 0.f; d.d1 = 0.f; d.d2 = 0.f; d.ninv = 0.f; d.x = 0.f; d.y = 0.f; d.z = 0.f;
  if (i < n) {   // colour has no depth test in the reference (kernel.cu:619-725)
    load_sh_row<K>(shs + (size_t)K * i, sh);
    d = sh_basis_f<NC>(ld3(pws + 3 * (size_t)i), twc);
  }
  if (jac) {
    rows_out<NC>(d.B, dcolor_dshs, n, base, stage);
    // (the direction is laundered: the compiler must not keep the 16 stored values alive for the pass below)
    asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z));
    if (i < n) sh_color_and_jac_dpw<NC, true, true>(d, sh, col, jp);
  } else if (i < n) {
    sh_color_and_jac_dpw<NC, true, false>(d, sh, col, jp);
  }
  rows_out<3>(col, colors, n, base, stage);
  if (jac) rows_out<9>(jp, dcolor_dpws, n, base, stage);
#else
  float B[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) B[c] = 0.f;
  if (i < n) {   // colour has no depth test in the reference (kernel.cu:619-725)
    float sh[K];
    load_sh_row<K>(shs + (size_t)K * i, sh);
    const ShDir<NC> d = sh_basis_f<NC>(ld3(pws + 3 * (size_t)i), twc);
    sh_color_f<NC>(d, sh, col);
    if (jac) {
#pragma unroll
      for (int c = 0; c < NC; ++c) B[c] = d.B[c];
      sh_jac_dpw<NC>(d, sh, jp);
    }
  }
  rows_out<3>(col, colors, n, base, stage);
  if (jac) {
    rows_out<NC>(B, dcolor_dshs, n, base, stage);
    rows_out<9>(jp, dcolor_dpws, n, base, stage);
  }
#endif
}

// ---- inverse_cov2d                                    (reference kernel.cu:274-324)
__global__ __launch_bounds__(256) void k_inv_cov2d(int n, const float* __restrict__ cov2ds,
                                                   float* __restrict__ depths, float det_eps,
                                                   int near_cull, int nan_cull, int radius_mode,
                                                   float* __restrict__ cinv2ds,
                                                   int32_t* __restrict__ areas,
                                                   float* __restrict__ dcinv2d_dcov2ds) {
  __shared__ float stage[RowsOut<9>::LDS_FLOATS];
  const int base = blockIdx.x * 256, i = base + threadIdx.x;
  float ci[3] = {0.f, 0.f, 0.f}, ar[2] = {0.f, 0.f}, J[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) J[k] = 0.f;
  if (i < n && !(near_cull && depths[i] < EGS_MIN_DEPTH)) {
    const float c2[3] = {cov2ds[3 * (size_t)i], cov2ds[3 * (size_t)i + 1], cov2ds[3 * (size_t)i + 2]};
    float cc[3];
    const float det_inv = inv_cov2d_f(c2, det_eps, cc);
    if (nan_cull && isnan(det_inv)) {
      depths[i] = EGS_BAD_MARKER;  // in-place contract GSFunction relies on (gsmodel.py:50)
    } else {
      ci[0] = cc[0]; ci[1] = cc[1]; ci[2] = cc[2];
      int rx, ry;
      radius_f(c2, radius_mode, rx, ry);
      ar[0] = __int_as_float(rx); ar[1] = __int_as_float(ry);   // (bit patterns of the int32 radii)
      if (dcinv2d_dcov2ds) inv_cov2d_jac(c2, det_inv, J);
    }
  }
  rows_out<3>(ci, cinv2ds, n, base, stage);
  rows_out<2>(ar, reinterpret_cast<float*>(areas), n, base, stage);
  if (dcinv2d_dcov2ds) rows_out<9>(J, dcinv2d_dcov2ds, n, base, stage);
}

// row loads of the chain-rule kernel: the widest vector load the row's alignment allows (a dword load per float
// costs one L1 lookup per lane and float: 24 instructions x 64 lookups for a dcov3d/drot row, 6 x 64 as dwordx4)
template <int K>
__device__ __forceinline__ void load_row(const float* __restrict__ row, float* out) {
  if constexpr (K % 4 == 0) {
#pragma unroll
    for (int j = 0; j < K / 4; ++j) {
      const float4 v = reinterpret_cast<const float4*>(row)[j];
      out[4 * j] = v.x; out[4 * j + 1] = v.y; out[4 * j + 2] = v.z; out[4 * j + 3] = v.w;
    }
  } else if constexpr (K % 2 == 0) {
#pragma unroll
    for (int j = 0; j < K / 2; ++j) {
      const float2 v = reinterpret_cast<const float2*>(row)[j];
      out[2 * j] = v.x; out[2 * j + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) out[j] = row[j];
  }
}

// ----------------------------------------------------------------------------
// fused chain rule: reference gsplat/gsmodel.py:71-85 == backward_cpu.py:476-482
// (nine batched [N,1,a]@[N,a,b] matmuls there; one pass over the Jacobians here)
// ----------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void k_chain_rule(
    int n, const float* __restrict__ dL_du, const float* __restrict__ dL_dcinv,
    const float* __restrict__ dL_dcolor, const float* __restrict__ Rcw,
    const float* __restrict__ J_cinv_cov2, const float* __restrict__ J_cov2_cov3,
    const float* __restrict__ J_cov3_rot, const float* __restrict__ J_cov3_scale,
    const float* __restrict__ J_color_sh, const float* __restrict__ J_u_pc,
    const float* __restrict__ J_cov2_pc, const float* __restrict__ J_color_pw,
    float* __restrict__ dL_dpw, float* __restrict__ dL_dsh, float* __restrict__ dL_dscale,
    float* __restrict__ dL_drot) {
  constexpr int K = 3 * NC;
  __shared__ float stage[cmax(RowsOut<K>::LDS_FLOATS, RowsOut<3>::LDS_FLOATS)];
  const int base = blockIdx.x * 256, i = base + threadIdx.x;
  float grot[4] = {0.f, 0.f, 0.f, 0.f}, gsc[3] = {0.f, 0.f, 0.f}, gpw[3] = {0.f, 0.f, 0.f}, gsh[K];
#pragma unroll
  for (int k = 0; k < K; ++k) gsh[k] = 0.f;
  if (i < n) {
    const f3 gci = ld3(dL_dcinv + 3 * (size_t)i);
    float A[9];
    load_row<9>(J_cinv_cov2 + 9 * (size_t)i, A);
    // dL/dcov2d = dL/dcinv2d @ J (row vector times 3x3)
    const f3 gc2 = {gci.x * A[0] + gci.y * A[3] + gci.z * A[6], gci.x * A[1] + gci.y * A[4] + gci.z * A[7],
                    gci.x * A[2] + gci.y * A[5] + gci.z * A[8]};
    float Bm[18];
    load_row<18>(J_cov2_cov3 + 18 * (size_t)i, Bm);
    float gc3[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) gc3[k] = gc2.x * Bm[k] + gc2.y * Bm[6 + k] + gc2.z * Bm[12 + k];
    float Cq[24];
    load_row<24>(J_cov3_rot + 24 * (size_t)i, Cq);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 6; ++r) s += gc3[r] * Cq[4 * r + k];
      grot[k] = s;
    }
    float Cs[18];
    load_row<18>(J_cov3_scale + 18 * (size_t)i, Cs);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 6; ++r) s += gc3[r] * Cs[3 * r + k];
      gsc[k] = s;
    }
    const f3 gcol = ld3(dL_dcolor + 3 * (size_t)i);
    float Bs[NC];
    load_row<NC>(J_color_sh + (size_t)NC * i, Bs);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      gsh[3 * c] = gcol.x * Bs[c]; gsh[3 * c + 1] = gcol.y * Bs[c]; gsh[3 * c + 2] = gcol.z * Bs[c];
    }
    // dL/dpc = dL/du @ du_dpc + dL/dcov2d @ dcov2d_dpc ; dL/dpw = dL/dpc @ Rcw + dL/dcolor @ dcolor_dpw
    const float2 gu = reinterpret_cast<const float2*>(dL_du)[i];
    const float gu0 = gu.x, gu1 = gu.y;
    float U[6], Pc[9], W[9];
    load_row<6>(J_u_pc + 6 * (size_t)i, U);
    load_row<9>(J_cov2_pc + 9 * (size_t)i, Pc);
    load_row<9>(J_color_pw + 9 * (size_t)i, W);
    const f3 gpc = {gu0 * U[0] + gu1 * U[3] + gc2.x * Pc[0] + gc2.y * Pc[3] + gc2.z * Pc[6],
                    gu0 * U[1] + gu1 * U[4] + gc2.x * Pc[1] + gc2.y * Pc[4] + gc2.z * Pc[7],
                    gu0 * U[2] + gu1 * U[5] + gc2.x * Pc[2] + gc2.y * Pc[5] + gc2.z * Pc[8]};
#pragma unroll
    for (int k = 0; k < 3; ++k)
      gpw[k] = gpc.x * Rcw[k] + gpc.y * Rcw[3 + k] + gpc.z * Rcw[6 + k] + gcol.x * W[k] + gcol.y * W[3 + k] +
               gcol.z * W[6 + k];
    *reinterpret_cast<float4*>(dL_drot + 4 * (size_t)i) = make_float4(grot[0], grot[1], grot[2], grot[3]);
  }
  rows_out<3>(gsc, dL_dscale, n, base, stage);
  rows_out<3>(gpw, dL_dpw, n, base, stage);
  rows_out<K>(gsh, dL_dsh, n, base, stage);   // the 4K-byte dL/dsh rows leave as full lines
}

// ============================================================================
// fused training path
// ============================================================================
struct PreParams {
  float fx, fy, cx, cy, limx, limy, det_eps, alpha_skip;
  int clamp_fov, near_cull, nan_cull, radius_mode, footprint, W, H;
  int stage_in;   // experiment knob (EGS_PRE_STAGE_IN): k_preprocess_bwd also stages the SH rows it reads
};

// ---- activations of the raw training parameters (reference gsplat/utils.py:121-150) -------------
// The RAW variants of the fused kernels read the optimizer's own tensors -- alphas_raw, scales_raw,
// rots_raw, low_shs [N,3] and high_shs [N,K-3] -- and return gradients with respect to THEM, so the
// torch ops around GSFunction (sigmoid, exp, normalize, cat and their autograd twins: ~0.4 ms per
// step at 1 M Gaussians, 12 launches, 0.9 GB of HBM traffic) disappear.
__device__ __forceinline__ float act_alpha(float a) { return 1.f / (1.f + expf(-a)); }   // get_alphas
__device__ __forceinline__ f3 act_scale(const f3& s) { return {expf(s.x), expf(s.y), expf(s.z)}; }  // get_scales
__device__ __forceinline__ float4 act_rot(const float4& r, float& norm) {                // get_rots
  norm = fmaxf(sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w), 1e-12f);            // F.normalize eps
  return make_float4(r.x / norm, r.y / norm, r.z / norm, r.w / norm);
}

// ---- cooperative row staging through LDS --------------------------------------------------
// A lane-per-Gaussian kernel that reads its own K-float SH row issues K/4 dwordx4 loads whose 64
// lanes are 4K bytes apart (64 different cache lines per instruction).  Staged instead: the
// workgroup's 256 rows are one contiguous span, fetched with fully coalesced dwordx4 loads into LDS,
// then every lane reads its row from LDS (row stride padded to an odd number of 16-B units:
// conflict-free ds_read_b128).  Same for the dL/dsh rows on the way out.
template <int K>
struct RowStage {
  static constexpr bool V4 = (K % 4 == 0);
  static constexpr int Q = K / 4;                                      // float4 per row (V4)
  static constexpr int STRIDE = V4 ? 4 * ((Q + 1) | 1) : (K | 1);      // floats
  static constexpr int LDS_FLOATS = 256 * STRIDE;
};

template <int K>
__device__ __forceinline__ void stage_rows_in(const float* __restrict__ src, int n, int base, float* lds, float* row) {
  using RS = RowStage<K>;
  const int rows = min(256, n - base);
  const int tid = threadIdx.x;
  if constexpr (RS::V4) {
    const float4* __restrict__ s4 = reinterpret_cast<const float4*>(src + (size_t)K * base);
#pragma unroll
    for (int j = 0; j < RS::Q; ++j) {
      const int f = tid + 256 * j;
      const int r = f / RS::Q, c = f - r * RS::Q;
      if (r < rows) *reinterpret_cast<float4*>(lds + r * RS::STRIDE + 4 * c) = s4[f];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS::Q; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(lds + tid * RS::STRIDE + 4 * j);
      row[4 * j] = v.x; row[4 * j + 1] = v.y; row[4 * j + 2] = v.z; row[4 * j + 3] = v.w;
    }
  } else {
    const float* __restrict__ s1 = src + (size_t)K * base;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int f = tid + 256 * j;
      const int r = f / K, c = f - r * K;
      if (r < rows) lds[r * RS::STRIDE + c] = s1[f];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < K; ++j) row[j] = lds[tid * RS::STRIDE + j];
  }
}

// the inverse: every lane deposits its row, the workgroup stores the span coalesced
// accum: dst += rows (a read-modify-write of the same coalesced span) instead of dst = rows
template <int K>
__device__ __forceinline__ void stage_rows_out(const float* row, float* __restrict__ dst, int n, int base, float* lds,
                                               bool accum = false) {
  using RS = RowStage<K>;
  const int rows = min(256, n - base);
  const int tid = threadIdx.x;
  __syncthreads();   // everyone is done reading the staged input rows
  if constexpr (RS::V4) {
#pragma unroll
    for (int j = 0; j < RS::Q; ++j)
      *reinterpret_cast<float4*>(lds + tid * RS::STRIDE + 4 * j) =
          make_float4(row[4 * j], row[4 * j + 1], row[4 * j + 2], row[4 * j + 3]);
    __syncthreads();
    float4* __restrict__ d4 = reinterpret_cast<float4*>(dst + (size_t)K * base);
    // accum: the old values of all the thread's pieces are requested first (clamped addresses, no branches) -- read
    // at its use inside the loop below, each piece waited for its own load AND for the previous piece's store
    float4 o[RS::Q];
    if (accum) {
      const int flast = rows * RS::Q - 1;
#pragma unroll
      for (int j = 0; j < RS::Q; ++j) o[j] = d4[min(tid + 256 * j, flast)];
    }
#pragma unroll
    for (int j = 0; j < RS::Q; ++j) {
      const int f = tid + 256 * j;
      const int r = f / RS::Q, c = f - r * RS::Q;
      if (r < rows) {
        float4 v = *reinterpret_cast<const float4*>(lds + r * RS::STRIDE + 4 * c);
        if (accum) { v.x += o[j].x; v.y += o[j].y; v.z += o[j].z; v.w += o[j].w; }
        d4[f] = v;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) lds[tid * RS::STRIDE + j] = row[j];
    __syncthreads();
    float* __restrict__ d1 = dst + (size_t)K * base;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int f = tid + 256 * j;
      const int r = f / K, c = f - r * K;
      if (r < rows) d1[f] = lds[r * RS::STRIDE + c] + (accum ? d1[f] : 0.f);
    }
  }
}

// Rows whose width is not a multiple of 4 floats (the 45-float high_shs rows): the workgroup's 256 rows
// are still ONE contiguous, 16-B aligned span (256 * K * 4 bytes), so the span moves as dwordx4 and lands
// in LDS unpadded; lane t then owns lds[t*K .. t*K+K) -- an odd K makes the dword reads conflict-free.
template <int K>
__device__ __forceinline__ void stage_span_in(const float* __restrict__ src, int n, int base, float* lds, float* row) {
  static_assert(K % 2 == 1, "odd row width expected (conflict-free LDS rows)");
  const int rows = min(256, n - base);
  const int tid = threadIdx.x;
  const int total = rows * K, nq = total >> 2;
  const float4* __restrict__ s4 = reinterpret_cast<const float4*>(src + (size_t)K * base);
  for (int f = tid; f < nq; f += 256) reinterpret_cast<float4*>(lds)[f] = s4[f];
  for (int f = 4 * nq + tid; f < total; f += 256) lds[f] = src[(size_t)K * base + f];   // <= 3 tail floats
  __syncthreads();
#pragma unroll
  for (int j = 0; j < K; ++j) row[j] = lds[tid * K + j];
}

template <int K>
__device__ __forceinline__ void stage_span_out(const float* row, float* __restrict__ dst, int n, int base, float* lds,
                                               bool accum = false) {
  const int rows = min(256, n - base);
  const int tid = threadIdx.x;
  __syncthreads();   // everyone is done reading the staged input rows
#pragma unroll
  for (int j = 0; j < K; ++j) lds[tid * K + j] = row[j];
  __syncthreads();
  const int total = rows * K, nq = total >> 2;
  float4* __restrict__ d4 = reinterpret_cast<float4*>(dst + (size_t)K * base);
  constexpr int NI = (K + 3) / 4;            // pieces per thread: 256 K / 4 float4 over 256 threads
  float4 o[NI];
  if (accum && nq > 0) {                     // (old values first, as in stage_rows_out)
#pragma unroll
    for (int j = 0; j < NI; ++j) o[j] = d4[min(tid + 256 * j, nq - 1)];
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int f = tid + 256 * j;
    if (f < nq) {
      float4 v = reinterpret_cast<const float4*>(lds)[f];
      if (accum) { v.x += o[j].x; v.y += o[j].y; v.z += o[j].z; v.w += o[j].w; }
      d4[f] = v;
    }
  }
  for (int f = 4 * nq + tid; f < total; f += 256)
    dst[(size_t)K * base + f] = lds[f] + (accum ? dst[(size_t)K * base + f] : 0.f);
}

// Occupancy: 20 KB of LDS per workgroup = exactly 8 workgroups per CU (8 waves per SIMD, which needs <= 64 VGPRs).
// With 7 -- one VGPR or 16 bytes of LDS too many -- the 3906 workgroups of 1 M Gaussians run as 2.2 "rounds" of
// 1792, i.e. the launch ends with a third round that is 20 % full.
// forward.md steps 1-5 for one Gaussian in one pass (== gsmodel.py:21-35 minus splat):
// writes exactly what splat / splatB / the backward pass consume.
// RAW: rots/scales/alphas are the un-activated tensors, shs = low_shs [N,3], shs_high = high_shs [N,K-3]
// JW: also write dcolor/dpw (dcolor_dpws) for the backward pass -- a training render; the SH Jacobian keeps ~40 more
// registers alive, so these instances are not pinned to 8 waves per SIMD
// A/B knob: rotation / scale / opacity requested together with the position.  The unpinned JW instance then also
// hoists all twelve SH loads (101 VGPRs, 5 waves per SIMD, ONE round trip per row instead of six): measured equal
// to the 58-register form within the noise of three same-box pairs (0.860-0.868 ms per step either way) -- the
// kernel is bound by the memory system's queues, not by the latency of a row.  Off.
#ifndef EGS_PRE_EARLY_LOADS
#define EGS_PRE_EARLY_LOADS 0
#endif
template <int NC, bool RAW, bool JW>
#ifndef EGS_PRE_JW_WAVES       // A/B knob: minimum waves per SIMD of the JW instances (register cap 512 / waves)
#define EGS_PRE_JW_WAVES 1
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RAW ? 1 : (JW ? EGS_PRE_JW_WAVES : 8), 8))) void k_preprocess_fwd(int n, PreParams pp, const float* __restrict__ pws,
                                                        const float* __restrict__ rots,
                                                        const float* __restrict__ scales,
                                                        const float* __restrict__ shs,
                                                        const float* __restrict__ shs_high,
                                                        const float* __restrict__ alphas,
                                                        const float* __restrict__ Rcw,
                                                        const float* __restrict__ tcw,
                                                        const float* __restrict__ twc,
                                                        float* __restrict__ us, float* __restrict__ depths,
                                                        float* __restrict__ cinv2ds,
                                                        float* __restrict__ colors,
                                                        int32_t* __restrict__ areas,
                                                        float4* __restrict__ rec, BinParams bp, BinCountOut bo,
                                                        uint8_t* __restrict__ visible,
                                                        float* __restrict__ dcolor_dpws) {
  // dcolor_dpws (nullable, [N][9]): dcolor/dpw of every Gaussian, for the backward pass -- the ONLY thing that pass
  // needs the SH coefficients for (eq (7): dL/dpw += dL/dcolor . dcolor/dpw; dL/dsh needs the basis alone).  36 B
  // written here save the 4K-byte SH row re-read there (192 B at SH degree 3).
  constexpr int K = 3 * NC;
  constexpr int KH = K - 3;   // width of high_shs
  constexpr int STAGE_FLOATS = (RAW && KH > 0 && RowStage<KH>::LDS_FLOATS > RowStage<12>::LDS_FLOATS)
                                   ? RowStage<KH>::LDS_FLOATS : RowStage<12>::LDS_FLOATS;
  __shared__ float stage[STAGE_FLOATS];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (bo.cr)     // the superblock sums of the depth sort that follows must start from zero
    for (uint32_t z = (uint32_t)i; z < bo.sort_sup_words; z += gridDim.x * 256u) bo.sort_sup[z] = 0u;
  uint32_t dkey = 0u;
  float sh[K];
  if constexpr (RAW) {   // 180-B high_shs rows cannot be dwordx4-loaded per lane: the workgroup's span through LDS
    if constexpr (KH > 0) {
      if constexpr (KH % 2 == 1) stage_span_in<KH>(shs_high, n, blockIdx.x * 256, stage, sh + 3);
      else stage_rows_in<KH>(shs_high, n, blockIdx.x * 256, stage, sh + 3);
    }
  }
  float4 r[3] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  float jw[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint4 crec = make_uint4(0u, 0u, 0u, 0u);
  if (i < n) {
    const f3 pw = ld3(pws + 3 * (size_t)i);
#if EGS_PRE_EARLY_LOADS
    // (requested with the position, not after the colour: one dependent round trip less per row)
    float4 q_in = *reinterpret_cast<const float4*>(rots + 4 * (size_t)i);
    f3 sc_in = ld3(scales + 3 * (size_t)i);
    const float alpha_in = (rec || bo.br) ? alphas[i] : 0.f;
#endif
    float col[3];
    {  // colour has no depth test in the reference (kernel.cu:619-725)
      if constexpr (RAW) {
        sh[0] = shs[3 * (size_t)i]; sh[1] = shs[3 * (size_t)i + 1]; sh[2] = shs[3 * (size_t)i + 2];
      } else {  // direct dwordx4 row loads: staging them through LDS measured 10 % slower here
        load_sh_row<K>(shs + (size_t)K * i, sh);
      }
      const ShDir<NC> d = sh_basis_f<NC>(pw, twc);
#if EGS_SH_FUSED_JAC_PRE
      if constexpr (JW) sh_color_and_jac_dpw<NC>(d, sh, col, jw);
      else sh_color_f<NC>(d, sh, col);
      if (colors) st3(colors + 3 * (size_t)i, {col[0], col[1], col[2]});
#else
      sh_color_f<NC>(d, sh, col);
      if (colors) st3(colors + 3 * (size_t)i, {col[0], col[1], col[2]});
      if constexpr (JW) sh_jac_dpw<NC>(d, sh, jw);
#endif
    }
    const Proj P = project_f(pw, Rcw, tcw, pp.fx, pp.fy, pp.cx, pp.cy);
    float u0 = 0.f, u1 = 0.f, depth = EGS_BAD_MARKER, ci[3] = {0.f, 0.f, 0.f};
    int rx = 0, ry = 0;
    if (!(pp.near_cull && P.pc.z < EGS_MIN_DEPTH)) {
      u0 = P.u0; u1 = P.u1; depth = P.pc.z;
#if EGS_PRE_EARLY_LOADS
      float4 q = q_in;
      f3 sc = sc_in;
#else
      float4 q = *reinterpret_cast<const float4*>(rots + 4 * (size_t)i);
      f3 sc = ld3(scales + 3 * (size_t)i);
#endif
      if constexpr (RAW) { float nrm; q = act_rot(q, nrm); sc = act_scale(sc); }
      const Cov3 c3 = cov3d_f(q, sc);
      const Cov2 c2 = cov2d_f(c3.c, P.pc, Rcw, pp.fx, pp.fy, pp.limx, pp.limy, pp.clamp_fov);
      const float det_inv = inv_cov2d_f(c2.c, pp.det_eps, ci);
      if (pp.nan_cull && isnan(det_inv)) {
        depth = EGS_BAD_MARKER; ci[0] = 0.f; ci[1] = 0.f; ci[2] = 0.f;
      } else {
        radius_f(c2.c, pp.radius_mode, rx, ry);
      }
    }
#if EGS_PRE_EARLY_LOADS
    const float alpha_act = (rec || bo.br) ? (RAW ? act_alpha(alpha_in) : alpha_in) : 0.f;
#else
    const float alpha_act = (rec || bo.br) ? (RAW ? act_alpha(alphas[i]) : alphas[i]) : 0.f;
#endif
    if (bo.br) {  // getRects + depth key of the binning stage, straight from registers (no k_bin_count pass)
      uint4 rect;
      bool cull;
      const uint32_t cnt = bin_count_one(bp, u0, u1, (float)rx, (float)ry, depth, rect, dkey, cull);
      if (cull) { depth = EGS_BAD_MARKER; rx = 0; ry = 0; }  // in-place contract of splat (kernel.cu:114-119)
      bo.ids[i] = (uint32_t)i;
      // the footprint record of the binning stage and the number of tiles the Gaussian is emitted for: its rect
      // (the reference's lists) or, bp.cull_lists, the tiles its footprint alpha' >= alpha_skip can reach
      const BinRec brec = make_binrec(u0, u1, ci[0], ci[1], ci[2], alpha_act, pp.alpha_skip, bp.cull_lists != 0,
                                      rect, cnt);
      if (cnt) {
        const uint32_t w = brec.wh & 0xFFFFu, h = brec.wh >> 16;
        if (w <= 4u && h <= 4u) {          // the blocks the footprint reaches, as a bitmap: emission is bit arithmetic
          const unsigned long long bits = foot_bitmap(brec);
          crec = make_uint4(brec.xy, brec.wh, (uint32_t)bits, (uint32_t)(bits >> 32));
        } else {                           // a bigger rect
          const bool walk = brec.m < __int_as_float(0x7f800000);
          if (walk && w <= 8u && h <= 8u) {   // its TILES as a bitmap; k_bin_emit evaluates the slabs of one tile
            const unsigned long long bits = foot_tilemap(brec);
            crec = make_uint4(brec.xy, brec.wh | EGS_CR_TILEMAP, (uint32_t)bits, (uint32_t)(bits >> 32));
          } else {                            // counted here, walked row by row by k_bin_emit
            crec = make_uint4(brec.xy, brec.wh | EGS_CR_BIG, walk ? foot_count(brec) : cnt, walk ? 1u : 0u);
          }
          if (walk) {
            float4* o = reinterpret_cast<float4*>(bo.br + i);
            o[0] = make_float4(brec.ux, brec.uy, brec.A, brec.Bh);
            o[1] = make_float4(brec.C, brec.m, __uint_as_float(brec.xy), __uint_as_float(brec.wh));
          }
        }
      }
      bo.dkeys[i] = dkey;
    }
    // us / cinv2ds / colors / areas are only needed by callers that go on with the seven-op surface; the
    // fused path draws from the packed records alone and passes NULL (40 B/Gaussian less to write)
    if (us) { us[2 * (size_t)i] = u0; us[2 * (size_t)i + 1] = u1; }
    depths[i] = depth;
    if (visible) visible[i] = depth > 0.2f;  // the mask GSFunction returns (gsmodel.py:50)
    if (cinv2ds) st3(cinv2ds + 3 * (size_t)i, {ci[0], ci[1], ci[2]});
    if (areas) { areas[2 * (size_t)i] = rx; areas[2 * (size_t)i + 1] = ry; }
    // the packed 2D record of the draw kernels, straight from registers (no k_pack_records pass)
    if (rec)
      make_record(u0, u1, ci[0], ci[1], ci[2], alpha_act, col[0], col[1], col[2], rx, ry,
                  pp.W, pp.H, pp.footprint, pp.alpha_skip, r);
  }
  if (bo.br) {
    __syncthreads();   // (RAW: every wave is done with the rows staged in)
    block_max_key(dkey, bo.maxkey, reinterpret_cast<uint32_t*>(stage));
    if (i < n) bo.cr[i] = crec;     // (16 B per lane, consecutive lanes: full lines)
  }
  // 48-B records leave as full lines (lane-strided 16-B pieces cost 3x the write requests)
  if (rec) stage_rows_out<12>(reinterpret_cast<const float*>(r), reinterpret_cast<float*>(rec), n, blockIdx.x * 256, stage);
  if constexpr (JW) rows_out<9>(jw, dcolor_dpws, n, blockIdx.x * 256, stage);
}

// backward.md eq (3)(4)(5)(7) == gsmodel.py:71-85 with every Jacobian re-derived in
// registers from the parameters.  gpack = the packed per-Gaussian gradient records
// written by k_draw_bwd: {dalpha, dcolor[3], du[2], dcinv[3], pad[3]}.
// RAW: parameters as in k_preprocess_fwd<.., true>; gradients come out with respect to the raw tensors
// (dL_dsh = low_shs [N,3], dL_dsh_high = high_shs [N,K-3]); alphas = alphas_raw (only read when RAW)
template <int NC, bool RAW, bool JW>
__global__ __launch_bounds__(256) void k_preprocess_bwd(
    int n, PreParams pp, const float* __restrict__ pws, const float* __restrict__ rots,
    const float* __restrict__ scales, const float* __restrict__ shs, const float* __restrict__ shs_high,
    const float* __restrict__ alphas, const float* __restrict__ Rcw,
    const float* __restrict__ tcw, const float* __restrict__ twc, const float* __restrict__ depths,
    const float4* __restrict__ gpack, float* __restrict__ dL_dpw, float* __restrict__ dL_dsh,
    float* __restrict__ dL_dsh_high, float* __restrict__ dL_dalpha, float* __restrict__ dL_dscale, float* __restrict__ dL_drot,
    float* __restrict__ dL_du, const float* __restrict__ dcolor_dpws, int mode) {
  // dcolor_dpws (nullable): [N][9] left by k_preprocess_fwd; with it this kernel never reads the SH coefficients
  // mode bit 1 (EGS_BWD_FACTORED_SH): the SH gradient stays in its factored form -- eq (5) is an outer product
  // dL/dcolour (x) basis, so dL_dsh receives the THREE floats dL/dcolour per Gaussian ([N][3], always written, never
  // accumulated) and the rows are formed once per step, for all views, by k_sh_grad_views; dL_dsh_high is not touched
  const int accum = mode & 1;
  const bool factored = (mode & 2) != 0;
  // accum: the five (six) parameter-gradient outputs already hold the gradients of EARLIER views of the step and this
  // view's are ADDED to them (dL_du is per view and always written): a rank that renders V views per step then needs
  // no separate accumulation kernels (torch's `.grad += new`: 976 B per Gaussian and view against 488 here)
  constexpr int K = 3 * NC;
  constexpr int KH = K - 3;
  constexpr int KS = RAW ? (KH > 0 ? KH : 1) : K;   // width of the rows that go through LDS
  __shared__ float stage[RowStage<KS>::LDS_FLOATS];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float sh[JW ? 1 : K], gsh[K];
  if constexpr (!JW) {
    if constexpr (RAW) {
      if constexpr (KH > 0) {
        if constexpr (KH % 2 == 1) stage_span_in<KH>(shs_high, n, blockIdx.x * 256, stage, sh + 3);
        else stage_rows_in<KH>(shs_high, n, blockIdx.x * 256, stage, sh + 3);
      }
      if (i < n) { sh[0] = shs[3 * (size_t)i]; sh[1] = shs[3 * (size_t)i + 1]; sh[2] = shs[3 * (size_t)i + 2]; }
    } else {
      if (pp.stage_in) stage_rows_in<K>(shs, n, blockIdx.x * 256, stage, sh);
      else if (i < n) load_sh_row<K>(shs + (size_t)K * i, sh);
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) gsh[k] = 0.f;
  f3 gcol_out = {0.f, 0.f, 0.f};
  if (i < n) {
    // Every input of the row is requested here, before anything is used: with the parameter loads behind the depth
    // test, the Jacobian row at its use and the old gradients (accum) at theirs, a row went through four dependent
    // round trips to memory (the ISA had a full wait after each group).
    const float4 ga = gpack[3 * (size_t)i], gb = gpack[3 * (size_t)i + 1], gc = gpack[3 * (size_t)i + 2];
    const float depth_i = depths[i];
    const f3 pw = ld3(pws + 3 * (size_t)i);
    float4 q = *reinterpret_cast<const float4*>(rots + 4 * (size_t)i);
    f3 s = ld3(scales + 3 * (size_t)i);
    float W[9];
    if constexpr (JW) load_row<9>(dcolor_dpws + 9 * (size_t)i, W);
    float al_raw = 0.f;
    if constexpr (RAW) al_raw = alphas[i];
    float4 o_rot = make_float4(0.f, 0.f, 0.f, 0.f);
    f3 o_scale = {0.f, 0.f, 0.f}, o_pw = {0.f, 0.f, 0.f};
    float o_alpha = 0.f;
    if (accum) {
      o_rot = *reinterpret_cast<const float4*>(dL_drot + 4 * (size_t)i);
      o_scale = ld3(dL_dscale + 3 * (size_t)i);
      o_pw = ld3(dL_dpw + 3 * (size_t)i);
      o_alpha = dL_dalpha[i];
    }
    const f3 gcol = {ga.y, ga.z, ga.w};
    const float gu0 = gb.x, gu1 = gb.y;
    const f3 gci = {gb.z, gb.w, gc.x};
    if constexpr (RAW) {
      const float al = act_alpha(al_raw);
      dL_dalpha[i] = ga.x * al * (1.f - al) + o_alpha;   // sigmoid'
    } else {
      dL_dalpha[i] = ga.x + o_alpha;
    }
    dL_du[2 * (size_t)i] = gu0; dL_du[2 * (size_t)i + 1] = gu1;
    if (pp.near_cull && depth_i < EGS_MIN_DEPTH) {  // culled: never drawn, all gradients are zero
      if (!accum) {
        st3(dL_dpw + 3 * (size_t)i, {0.f, 0.f, 0.f});
        st3(dL_dscale + 3 * (size_t)i, {0.f, 0.f, 0.f});
        st4(dL_drot + 4 * (size_t)i, {0.f, 0.f, 0.f, 0.f});
      }
    } else {
      float qnorm = 1.f;
      if constexpr (RAW) { q = act_rot(q, qnorm); s = act_scale(s); }
      const Proj P = project_f(pw, Rcw, tcw, pp.fx, pp.fy, pp.cx, pp.cy);
      const Cov3 c3 = cov3d_f(q, s);
      const Cov2 c2 = cov2d_f(c3.c, P.pc, Rcw, pp.fx, pp.fy, pp.limx, pp.limy, pp.clamp_fov);
      float ci[3];
      const float det_inv = inv_cov2d_f(c2.c, pp.det_eps, ci);
      float Ji[9];
      inv_cov2d_jac(c2.c, det_inv, Ji);
      // dL/dcov2d = dL/dcinv2d @ J  (row vector times 3x3)
      const float g2[3] = {gci.x * Ji[0] + gci.y * Ji[3] + gci.z * Ji[6],
                           gci.x * Ji[1] + gci.y * Ji[4] + gci.z * Ji[7],
                           gci.x * Ji[2] + gci.y * Ji[5] + gci.z * Ji[8]};
      float J3[18], Jp[9];
      cov2d_jac(c2, P.pc.z, Rcw, pp.fx, pp.fy, J3, Jp);
      float g3[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) g3[k] = g2[0] * J3[k] + g2[1] * J3[6 + k] + g2[2] * J3[12 + k];
      q4 gq; f3 gs;
      cov3d_vjp(c3, q, s, g3, gq, gs);
      if constexpr (RAW) {   // through normalize: (g - q (q.g)) / |r|; through exp: g * scale
        const float qg = q.x * gq.w + q.y * gq.x + q.z * gq.y + q.w * gq.z;
        gq = {(gq.w - q.x * qg) / qnorm, (gq.x - q.y * qg) / qnorm, (gq.y - q.z * qg) / qnorm,
              (gq.z - q.w * qg) / qnorm};
        gs = {gs.x * s.x, gs.y * s.y, gs.z * s.z};
      }
      if (accum) {
        gq = {gq.w + o_rot.x, gq.x + o_rot.y, gq.y + o_rot.z, gq.z + o_rot.w};
        gs = {gs.x + o_scale.x, gs.y + o_scale.y, gs.z + o_scale.z};
      }
      st4(dL_drot + 4 * (size_t)i, gq);      // eq (3)
      st3(dL_dscale + 3 * (size_t)i, gs);    // eq (4)
      float j00, j02, j11, j12;
      project_jac(P, pp.fx, pp.fy, j00, j02, j11, j12);
      const f3 gpc = {gu0 * j00 + g2[0] * Jp[0] + g2[1] * Jp[3] + g2[2] * Jp[6],
                      gu1 * j11 + g2[0] * Jp[1] + g2[1] * Jp[4] + g2[2] * Jp[7],
                      gu0 * j02 + gu1 * j12 + g2[0] * Jp[2] + g2[1] * Jp[5] + g2[2] * Jp[8]};
      const ShDir<NC> d = sh_basis_f<NC>(pw, twc);
      // eq (5): dL/dsh[c, rgb] = dL/dcolor[rgb] * basis[c]
      gcol_out = gcol;
      if (!factored) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          gsh[3 * c] = gcol.x * d.B[c]; gsh[3 * c + 1] = gcol.y * d.B[c]; gsh[3 * c + 2] = gcol.z * d.B[c];
        }
      }
      if constexpr (!JW) sh_jac_dpw<NC>(d, sh, W);
      float* opw = dL_dpw + 3 * (size_t)i;  // eq (7)
      const float opw_old[3] = {o_pw.x, o_pw.y, o_pw.z};
#pragma unroll
      for (int k = 0; k < 3; ++k)
        opw[k] = gpc.x * Rcw[k] + gpc.y * Rcw[3 + k] + gpc.z * Rcw[6 + k] + gcol.x * W[k] + gcol.y * W[3 + k] +
                 gcol.z * W[6 + k] + opw_old[k];
    }
  }
  if (factored) {   // (a kernel argument: the whole workgroup leaves here)
    if (i < n) st3(dL_dsh + 3 * (size_t)i, gcol_out);
    // the view's camera centre behind the [N][3] block: the row format of egs_sh_grad_views (dL_dsh_high = its address)
    if (dL_dsh_high && blockIdx.x == 0 && threadIdx.x < 3) dL_dsh_high[threadIdx.x] = twc[threadIdx.x];
    return;
  }
  if constexpr (RAW) {
    if (i < n) {
#pragma unroll
      for (int k = 0; k < 3; ++k) dL_dsh[3 * (size_t)i + k] = gsh[k] + (accum ? dL_dsh[3 * (size_t)i + k] : 0.f);
    }
    if constexpr (KH > 0) {
      if constexpr (KH % 2 == 1) stage_span_out<KH>(gsh + 3, dL_dsh_high, n, blockIdx.x * 256, stage, accum != 0);
      else stage_rows_out<KH>(gsh + 3, dL_dsh_high, n, blockIdx.x * 256, stage, accum != 0);
    }
  } else {
    stage_rows_out<K>(gsh, dL_dsh, n, blockIdx.x * 256, stage, accum != 0);
  }
}

// The SH gradient of a step from its FACTORED form (dist_views.FactoredShGrad).  For one view dL/dsh[c][rgb] is the
// outer product dL/dcolour[rgb] * basis_c(pw - camera centre) -- eq (5), gsmodel.py:84-85 -- so the views of a step
// (this rank's and, all-gathered, every other rank's) travel as 3 floats per Gaussian and view instead of 48 per
// Gaussian and rank, and this kernel forms  scale * sum_v dcolour_v (x) basis(pw - twc_v)  once.
// rows: [views][stride] floats, row v = {dL/dcolour [N][3], twc[3], padding}.  Outputs as k_preprocess_bwd's.
template <int NC, bool RAW>
__global__ __launch_bounds__(256) void k_sh_grad_views(int n, int views, const float* __restrict__ pws,
                                                       const float* __restrict__ rows, int64_t stride, float scale,
                                                       float* __restrict__ dL_dsh, float* __restrict__ dL_dsh_high,
                                                       int accum) {
  constexpr int K = 3 * NC;
  constexpr int KH = K - 3;
  constexpr int KS = RAW ? (KH > 0 ? KH : 1) : K;
  __shared__ float stage[RowStage<KS>::LDS_FLOATS];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float gsh[K];
#pragma unroll
  for (int k = 0; k < K; ++k) gsh[k] = 0.f;
  if (i < n) {
    const f3 pw = ld3(pws + 3 * (size_t)i);
    for (int v = 0; v < views; ++v) {
      const float* row = rows + (size_t)v * stride;
      const f3 g = ld3(row + 3 * (size_t)i);
      if (g.x == 0.f && g.y == 0.f && g.z == 0.f) continue;   // not drawn in this view
      const ShDir<NC> d = sh_basis_f<NC>(pw, row + 3 * (size_t)n);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        gsh[3 * c] = __builtin_fmaf(g.x, d.B[c], gsh[3 * c]);
        gsh[3 * c + 1] = __builtin_fmaf(g.y, d.B[c], gsh[3 * c + 1]);
        gsh[3 * c + 2] = __builtin_fmaf(g.z, d.B[c], gsh[3 * c + 2]);
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) gsh[k] *= scale;
  }
  if constexpr (RAW) {
    if (i < n) {
#pragma unroll
      for (int k = 0; k < 3; ++k) dL_dsh[3 * (size_t)i + k] = gsh[k] + (accum ? dL_dsh[3 * (size_t)i + k] : 0.f);
    }
    if constexpr (KH > 0) {
      if constexpr (KH % 2 == 1) stage_span_out<KH>(gsh + 3, dL_dsh_high, n, blockIdx.x * 256, stage, accum != 0);
      else stage_rows_out<KH>(gsh + 3, dL_dsh_high, n, blockIdx.x * 256, stage, accum != 0);
    }
  } else {
    stage_rows_out<K>(gsh, dL_dsh, n, blockIdx.x * 256, stage, accum != 0);
  }
}

}  // namespace egs

// ============================================================================
// C ABI
// ============================================================================
using namespace egs;

static void fov_limits(const EgsPolicy* pol, float fx, float fy, float width, float height, float* limx,
                       float* limy) {
  *limx = 0.f; *limy = 0.f;
  if (pol->fov_mode == 0) {  // gausplat.cu:225-226
    *limx = 1.3f * (width / (2 * fx));
    *limy = 1.3f * (height / (2 * fy));
  } else if (pol->fov_mode == 1) {  // gausplat.py:136-140 (an angle, misnamed tan there)
    *limx = (float)(1.3 * (2 * atan((double)width / (2 * (double)fx))));
    *limy = (float)(1.3 * (2 * atan((double)height / (2 * (double)fy))));
  }
}

extern "C" int egs_project(int n, const float* pws, const float* Rcw, const float* tcw, float fx, float fy,
                           float cx, float cy, const EgsPolicy* pol, float* us, float* pcs, float* depths,
                           float* du_dpcs, void* stream) {
  EGS_CHECK_ARG(n >= 0 && pol);
  if (n == 0) return 0;
  EGS_CHECK_ARG(pws && Rcw && tcw && us && pcs && depths);
  EGS_CHECK_ARG((((uintptr_t)us | (uintptr_t)du_dpcs) & 7) == 0);
  EGS_LAUNCH("k_project", k_project, dim3(div_up(n, 256)), dim3(256), (hipStream_t)stream, n, pws, Rcw, tcw, fx,
                     fy, cx, cy, pol->near_cull, us, pcs, depths, du_dpcs);
  EGS_LAUNCH_OK();
  return 0;
}

extern "C" int egs_cov3d(int n, const float* rots, const float* scales, const float* depths,
                         const EgsPolicy* pol, float* cov3ds, float* dcov3d_drots, float* dcov3d_dscales,
                         void* stream) {
  EGS_CHECK_ARG(n >= 0 && pol);
  if (n == 0) return 0;
  EGS_CHECK_ARG(rots && scales && depths && cov3ds);
  EGS_CHECK_ARG((dcov3d_drots == nullptr) == (dcov3d_dscales == nullptr));
  EGS_CHECK_ARG(((uintptr_t)rots & 15) == 0);
  EGS_CHECK_ARG((((uintptr_t)cov3ds | (uintptr_t)dcov3d_drots | (uintptr_t)dcov3d_dscales) & 15) == 0);
  EGS_LAUNCH("k_cov3d", k_cov3d, dim3(div_up(n, 256)), dim3(256), (hipStream_t)stream, n, rots, scales, depths,
                     pol->near_cull, cov3ds, dcov3d_drots, dcov3d_dscales);
  EGS_LAUNCH_OK();
  return 0;
}

extern "C" int egs_cov2d(int n, const float* cov3ds, const float* pcs, const float* Rcw, const float* depths,
                         float fx, float fy, float width, float height, const EgsPolicy* pol, float* cov2ds,
                         float* dcov2d_dcov3ds, float* dcov2d_dpcs, void* stream) {
  EGS_CHECK_ARG(n >= 0 && pol);
  if (n == 0) return 0;
  EGS_CHECK_ARG(cov3ds && pcs && Rcw && depths && cov2ds);
  EGS_CHECK_ARG((dcov2d_dcov3ds == nullptr) == (dcov2d_dpcs == nullptr));
  EGS_CHECK_ARG((((uintptr_t)cov3ds & 7) | (((uintptr_t)cov2ds | (uintptr_t)dcov2d_dcov3ds | (uintptr_t)dcov2d_dpcs) & 15)) == 0);
  float limx, limy;
  fov_limits(pol, fx, fy, width, height, &limx, &limy);
  EGS_LAUNCH("k_cov2d", k_cov2d, dim3(div_up(n, 256)), dim3(256), (hipStream_t)stream, n, cov3ds, pcs, Rcw,
                     depths, fx, fy, limx, limy, pol->fov_mode != 2, pol->near_cull, cov2ds, dcov2d_dcov3ds,
                     dcov2d_dpcs);
  EGS_LAUNCH_OK();
  return 0;
}

extern "C" int egs_sh2color(int n, int sh_dim, const float* shs, const float* pws, const float* twc,
                            float* colors, float* dcolor_dshs, float* dcolor_dpws, void* stream) {
  EGS_CHECK_ARG(n >= 0);
  EGS_CHECK_ARG(sh_dim == 3 || sh_dim == 12 || sh_dim == 27 || sh_dim == 48);
  if (n == 0) return 0;
  EGS_CHECK_ARG(shs && pws && twc && colors);
  EGS_CHECK_ARG((dcolor_dshs == nullptr) == (dcolor_dpws == nullptr));
  EGS_CHECK_ARG(sh_dim % 4 != 0 || ((uintptr_t)shs & 15) == 0);
  EGS_CHECK_ARG((((uintptr_t)colors | (uintptr_t)dcolor_dshs | (uintptr_t)dcolor_dpws) & 15) == 0);
  dim3 g(div_up(n, 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
  switch (sh_dim) {
    case 3: EGS_LAUNCH("k_sh2color", (k_sh2color<1>), g, b, s, n, shs, pws, twc, colors, dcolor_dshs, dcolor_dpws); break;
    case 12: EGS_LAUNCH("k_sh2color", (k_sh2color<4>), g, b, s, n, shs, pws, twc, colors, dcolor_dshs, dcolor_dpws); break;
    case 27: EGS_LAUNCH("k_sh2color", (k_sh2color<9>), g, b, s, n, shs, pws, twc, colors, dcolor_dshs, dcolor_dpws); break;
    default: EGS_LAUNCH("k_sh2color", (k_sh2color<16>), g, b, s, n, shs, pws, twc, colors, dcolor_dshs, dcolor_dpws); break;
  }
  EGS_LAUNCH_OK();
  return 0;
}

extern "C" int egs_inv_cov2d(int n, const float* cov2ds, float* depths, const EgsPolicy* pol, float* cinv2ds,
                             int32_t* areas, float* dcinv2d_dcov2ds, void* stream) {
  EGS_CHECK_ARG(n >= 0 && pol);
  if (n == 0) return 0;
  EGS_CHECK_ARG(cov2ds && depths && cinv2ds && areas);
  EGS_CHECK_ARG((((uintptr_t)cinv2ds | (uintptr_t)areas | (uintptr_t)dcinv2d_dcov2ds) & 15) == 0);
  EGS_LAUNCH("k_inv_cov2d", k_inv_cov2d, dim3(div_up(n, 256)), dim3(256), (hipStream_t)stream, n, cov2ds, depths,
                     pol->det_eps, pol->near_cull, pol->nan_cull, pol->radius_mode, cinv2ds, areas,
                     dcinv2d_dcov2ds);
  EGS_LAUNCH_OK();
  return 0;
}

extern "C" int egs_chain_rule(int n, int sh_dim, const float* dloss_dus, const float* dloss_dcinv2ds,
                              const float* dloss_dcolors, const float* Rcw, const float* dcinv2d_dcov2ds,
                              const float* dcov2d_dcov3ds, const float* dcov3d_drots,
                              const float* dcov3d_dscales, const float* dcolor_dshs, const float* du_dpcs,
                              const float* dcov2d_dpcs, const float* dcolor_dpws, float* dloss_dpws,
                              float* dloss_dshs, float* dloss_dscales, float* dloss_drots, void* stream) {
  EGS_CHECK_ARG(n >= 0);
  EGS_CHECK_ARG(sh_dim == 3 || sh_dim == 12 || sh_dim == 27 || sh_dim == 48);
  if (n == 0) return 0;
  EGS_CHECK_ARG(dloss_dus && dloss_dcinv2ds && dloss_dcolors && Rcw && dcinv2d_dcov2ds && dcov2d_dcov3ds &&
                dcov3d_drots && dcov3d_dscales && dcolor_dshs && du_dpcs && dcov2d_dpcs && dcolor_dpws &&
                dloss_dpws && dloss_dshs && dloss_dscales && dloss_drots);
  EGS_CHECK_ARG((((uintptr_t)dloss_dpws | (uintptr_t)dloss_dshs | (uintptr_t)dloss_dscales | (uintptr_t)dloss_drots |
                  (uintptr_t)dcov3d_drots | (sh_dim % 12 == 0 ? (uintptr_t)dcolor_dshs : 0)) & 15) == 0);
  EGS_CHECK_ARG((((uintptr_t)dloss_dus | (uintptr_t)dcov2d_dcov3ds | (uintptr_t)dcov3d_dscales | (uintptr_t)du_dpcs) & 7) == 0);
  dim3 g(div_up(n, 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
#define EGS_CHAIN(NC)                                                                                        \
  EGS_LAUNCH("k_chain_rule", (k_chain_rule<NC>), g, b, s, n, dloss_dus, dloss_dcinv2ds, dloss_dcolors, Rcw,          \
                     dcinv2d_dcov2ds, dcov2d_dcov3ds, dcov3d_drots, dcov3d_dscales, dcolor_dshs, du_dpcs,     \
                     dcov2d_dpcs, dcolor_dpws, dloss_dpws, dloss_dshs, dloss_dscales, dloss_drots)
  switch (sh_dim) {
    case 3: EGS_CHAIN(1); break;
    case 12: EGS_CHAIN(4); break;
    case 27: EGS_CHAIN(9); break;
    default: EGS_CHAIN(16); break;
  }
#undef EGS_CHAIN
  EGS_LAUNCH_OK();
  return 0;
}

static PreParams make_pre_params(const EgsPolicy* pol, float fx, float fy, float cx, float cy, int width,
                                 int height) {
  PreParams pp;
  pp.fx = fx; pp.fy = fy; pp.cx = cx; pp.cy = cy;
  fov_limits(pol, fx, fy, (float)width, (float)height, &pp.limx, &pp.limy);
  pp.det_eps = pol->det_eps; pp.alpha_skip = pol->alpha_skip;
  pp.footprint = pol->footprint; pp.W = width; pp.H = height;
  pp.clamp_fov = pol->fov_mode != 2; pp.near_cull = pol->near_cull; pp.nan_cull = pol->nan_cull;
  pp.radius_mode = pol->radius_mode;
  static const int stage_in = [] { const char* e = getenv("EGS_PRE_STAGE_IN"); return e ? atoi(e) : 0; }();
  pp.stage_in = stage_in;
  return pp;
}

static int fused_forward_impl(bool raw, int n, int sh_dim, const float* pws, const float* rots, const float* scales,
                              const float* shs, const float* shs_high, const float* alphas, const float* Rcw,
                              const float* tcw, const float* twc, float fx, float fy, float cx, float cy, int width,
                              int height, const EgsPolicy* pol, float* us, float* depths, float* cinv2ds,
                              float* colors, int32_t* areas, void* rec, uint8_t* visible, float* dcolor_dpws,
                              int cull_lists, int key_bits_hint, void* ws_bin, size_t ws_bin_bytes,
                              uint32_t* total_patches, uint32_t* host_totals, void* stream) {
  EGS_CHECK_ARG(n >= 0 && pol && width > 0 && height > 0 && total_patches);
  EGS_CHECK_ARG(!cull_lists || (rec && n < (1 << EGS_GSID_BITS)));   // culled lists are drawn from the records only
  EGS_CHECK_ARG(((uintptr_t)dcolor_dpws & 15) == 0);
  EGS_CHECK_ARG(width < 32768 && height < 32768);
  EGS_CHECK_ARG(sh_dim == 3 || sh_dim == 12 || sh_dim == 27 || sh_dim == 48);
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    EGS_HIP(hipMemsetAsync(total_patches, 0, 8, s));
    if (host_totals) EGS_HIP(hipMemcpyAsync(host_totals, total_patches, 8, hipMemcpyDeviceToHost, s));
    return 0;
  }
  EGS_CHECK_ARG(pws && rots && scales && shs && Rcw && tcw && twc && depths);
  EGS_CHECK_ARG(rec || (us && cinv2ds && colors && areas));   // something must carry the 2D Gaussians on
  EGS_CHECK_ARG(!rec || alphas);
  EGS_CHECK_ARG(ws_bin);
  EGS_CHECK_ARG(((uintptr_t)rots & 15) == 0);
  if (raw) EGS_CHECK_ARG(sh_dim == 3 || (shs_high && ((uintptr_t)shs_high & 15) == 0));
  else EGS_CHECK_ARG(sh_dim % 4 != 0 || ((uintptr_t)shs & 15) == 0);
  BinCountOut bo;
  if (!bin_count_outputs(ws_bin, ws_bin_bytes, n, &bo)) {
    set_error(EGS_ERR_WORKSPACE, "bin workspace too small", __FILE__, __LINE__);
    return EGS_ERR_WORKSPACE;
  }
  const BinParams bp = make_bin_params(width, height, pol, cull_lists != 0);
  const PreParams pp = make_pre_params(pol, fx, fy, cx, cy, width, height);
  dim3 g(div_up(n, 256)), b(256);
  // EGS_PRE_LDS_PAD (bytes of dynamic LDS, experiment knob): caps the resident workgroups per CU of this kernel
  static const size_t lds_pad = [] { const char* e = getenv("EGS_PRE_LDS_PAD"); return e ? (size_t)atoi(e) : (size_t)0; }();
#define EGS_PRE(NC, RAW)                                                                                        \
  do {                                                                                                          \
    if (dcolor_dpws)                                                                                            \
      EGS_LAUNCH_LDS("k_preprocess_fwd", (k_preprocess_fwd<NC, RAW, true>), g, b, lds_pad, s, n, pp, pws, rots, scales, shs, \
                 shs_high, alphas, Rcw, tcw, twc, us, depths, cinv2ds, colors, areas, (float4*)rec, bp, bo,     \
                 visible, dcolor_dpws);                                                                         \
    else                                                                                                        \
      EGS_LAUNCH_LDS("k_preprocess_fwd", (k_preprocess_fwd<NC, RAW, false>), g, b, lds_pad, s, n, pp, pws, rots, scales, shs, \
                 shs_high, alphas, Rcw, tcw, twc, us, depths, cinv2ds, colors, areas, (float4*)rec, bp, bo,     \
                 visible, dcolor_dpws);                                                                         \
  } while (0)
  switch (sh_dim * 2 + (raw ? 1 : 0)) {
    case 6: EGS_PRE(1, false); break;
    case 7: EGS_PRE(1, true); break;
    case 24: EGS_PRE(4, false); break;
    case 25: EGS_PRE(4, true); break;
    case 54: EGS_PRE(9, false); break;
    case 55: EGS_PRE(9, true); break;
    case 96: EGS_PRE(16, false); break;
    default: EGS_PRE(16, true); break;
  }
#undef EGS_PRE
  EGS_LAUNCH_OK();
  // the kernel above already did getRects + depth keys (k_bin_count of egs_splat_bin)
  return splat_bin_after_count(n, key_bits_hint, ws_bin, ws_bin_bytes, total_patches, stream, host_totals);
}

extern "C" int egs_fused_forward(int n, int sh_dim, const float* pws, const float* rots, const float* scales,
                                 const float* shs, const float* alphas, const float* Rcw, const float* tcw,
                                 const float* twc, float fx, float fy, float cx, float cy, int width, int height,
                                 const EgsPolicy* pol, float* us, float* depths, float* cinv2ds, float* colors,
                                 int32_t* areas, void* rec, uint8_t* visible, float* dcolor_dpws, int cull_lists,
                                 int key_bits_hint, void* ws_bin, size_t ws_bin_bytes, uint32_t* total_patches,
                                 uint32_t* host_totals, void* stream) {
  return fused_forward_impl(false, n, sh_dim, pws, rots, scales, shs, nullptr, alphas, Rcw, tcw, twc, fx, fy, cx, cy,
                            width, height, pol, us, depths, cinv2ds, colors, areas, rec, visible, dcolor_dpws,
                            cull_lists, key_bits_hint, ws_bin, ws_bin_bytes, total_patches, host_totals, stream);
}

extern "C" int egs_fused_forward_raw(int n, int sh_dim, const float* pws, const float* rots_raw,
                                     const float* scales_raw, const float* low_shs, const float* high_shs,
                                     const float* alphas_raw, const float* Rcw, const float* tcw, const float* twc,
                                     float fx, float fy, float cx, float cy, int width, int height,
                                     const EgsPolicy* pol, float* us, float* depths, float* cinv2ds, float* colors,
                                     int32_t* areas, void* rec, uint8_t* visible, float* dcolor_dpws,
                                     int cull_lists, int key_bits_hint, void* ws_bin, size_t ws_bin_bytes,
                                     uint32_t* total_patches, uint32_t* host_totals, void* stream) {
  EGS_CHECK_ARG(n == 0 || (rec && alphas_raw));  // the activated alpha only exists inside the records
  return fused_forward_impl(true, n, sh_dim, pws, rots_raw, scales_raw, low_shs, high_shs, alphas_raw, Rcw, tcw, twc,
                            fx, fy, cx, cy, width, height, pol, us, depths, cinv2ds, colors, areas, rec, visible,
                            dcolor_dpws, cull_lists, key_bits_hint, ws_bin, ws_bin_bytes, total_patches, host_totals,
                            stream);
}

extern "C" size_t egs_fused_backward_ws_bytes(int n) { return egs_splat_bwd_ws_bytes(n); }

static int fused_backward_impl(bool raw, int n, int sh_dim, int64_t patches, int width, int height, const float* pws,
                               const float* rots, const float* scales, const float* shs, const float* shs_high,
                               const float* alphas, const float* Rcw, const float* tcw, const float* twc, float fx,
                               float fy, float cx, float cy, const EgsPolicy* pol, const float* us,
                               const float* cinv2ds, const float* colors, const int32_t* areas, const void* rec,
                               const float* depths, const int32_t* contrib, const float* final_tau,
                               const int32_t* patch_range_per_tile, const int32_t* gsid_per_patch,
                               const float* dloss_dgammas, void* ws, size_t ws_bytes, float* dloss_dpws,
                               float* dloss_dshs, float* dloss_dshs_high, float* dloss_dalphas, float* dloss_dscales,
                               float* dloss_drots, float* dloss_dus, const int32_t* tile_order,
                               float* grad_records, const float* dcolor_dpws, int phase, int row_begin, int row_count,
                               void* seg_ws, size_t seg_ws_bytes, void* stream) {
  // phase 0: everything; 1: only the draw pass (-> packed gradient records in ws); 2: only the per-Gaussian
  // chain rule, for rows [row_begin, row_begin + row_count) -- a data-parallel caller launches the rows in a
  // few chunks and starts exchanging a chunk's gradients while the next one is computed (dist_views)
  EGS_CHECK_ARG(n >= 0 && pol && width > 0 && height > 0 && patches >= 0);
  const bool keep_order = (phase & EGS_BWD_KEEP_FORWARD_ORDER) != 0;
  const bool masked = (phase & EGS_BWD_CULLED_LISTS) != 0;
  // kernel `mode`: bit 0 accumulate, bit 1 factored SH gradient (dloss_dshs = dL/dcolour [N][3])
  const int accum = ((phase & EGS_BWD_ACCUMULATE) ? 1 : 0) | ((phase & EGS_BWD_FACTORED_SH) ? 2 : 0);
  const bool factored = (phase & EGS_BWD_FACTORED_SH) != 0;
  phase &= ~(EGS_BWD_KEEP_FORWARD_ORDER | EGS_BWD_CULLED_LISTS | EGS_BWD_ACCUMULATE | EGS_BWD_FACTORED_SH);
  EGS_CHECK_ARG(phase >= 0 && phase <= 2);
  if (phase != 2) { row_begin = 0; row_count = n; }
  EGS_CHECK_ARG(row_begin >= 0 && row_count >= 0 && row_begin + (int64_t)row_count <= n && row_begin % 256 == 0);
  EGS_CHECK_ARG(sh_dim == 3 || sh_dim == 12 || sh_dim == 27 || sh_dim == 48);
  if (n == 0) return 0;
  EGS_CHECK_ARG(pws && rots && scales && shs && alphas && Rcw && tcw && twc && depths && ws && dloss_dpws &&
                dloss_dshs && dloss_dalphas && dloss_dscales && dloss_drots && dloss_dus);
  if (raw) EGS_CHECK_ARG(rec && (sh_dim == 3 || (shs_high && (dloss_dshs_high || factored))));
  if (ws_bytes < egs_splat_bwd_ws_bytes(n)) {
    set_error(EGS_ERR_WORKSPACE, "fused_backward workspace too small", __FILE__, __LINE__);
    return EGS_ERR_WORKSPACE;
  }
  // where splat_bwd_packed puts the records
  float* gpack = grad_records ? grad_records : (float*)((char*)ws + align_up((size_t)n * 48, 256));
  if (phase != 2) {
    int rc = splat_bwd_packed(n, patches, width, height, us, cinv2ds, alphas, colors, areas, pol, contrib, final_tau,
                              patch_range_per_tile, gsid_per_patch, dloss_dgammas, ws, ws_bytes, &gpack, stream, rec,
                              tile_order, grad_records, keep_order, masked, seg_ws, seg_ws_bytes);
    if (rc) return rc;
    if (phase == 1) return 0;
  }
  if (row_count == 0) return 0;
  const PreParams pp = make_pre_params(pol, fx, fy, cx, cy, width, height);
  const int kh = sh_dim - 3;
  const size_t r0 = (size_t)row_begin;
  dim3 g(div_up(row_count, 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
  // row_begin is a multiple of the workgroup's 256 rows: every offset pointer keeps its 16-B alignment
#define EGS_PREB_ARGS(NC, RAW)                                                                                    \
  row_count, pp, pws + 3 * r0, rots + 4 * r0, scales + 3 * r0, shs + (RAW ? 3 : sh_dim) * r0,                      \
      (RAW && shs_high) ? shs_high + kh * r0 : shs_high, alphas + r0, Rcw, tcw, twc, depths + r0,                  \
      (const float4*)gpack + 3 * r0, dloss_dpws + 3 * r0, dloss_dshs + (factored ? 3 : (RAW ? 3 : sh_dim)) * r0,   \
      factored ? (row_begin == 0 ? dloss_dshs + 3 * (size_t)n : nullptr)                                           \
               : ((RAW && dloss_dshs_high) ? dloss_dshs_high + kh * r0 : dloss_dshs_high), dloss_dalphas + r0,       \
      dloss_dscales + 3 * r0, dloss_drots + 4 * r0, dloss_dus + 2 * r0, dcolor_dpws ? dcolor_dpws + 9 * r0 : dcolor_dpws, \
      accum
#define EGS_PREB(NC, RAW)                                                                                         \
  do {                                                                                                            \
    if (dcolor_dpws) EGS_LAUNCH("k_preprocess_bwd", (k_preprocess_bwd<NC, RAW, true>), g, b, s, EGS_PREB_ARGS(NC, RAW)); \
    else EGS_LAUNCH("k_preprocess_bwd", (k_preprocess_bwd<NC, RAW, false>), g, b, s, EGS_PREB_ARGS(NC, RAW));     \
  } while (0)
  switch (sh_dim * 2 + (raw ? 1 : 0)) {
    case 6: EGS_PREB(1, false); break;
    case 7: EGS_PREB(1, true); break;
    case 24: EGS_PREB(4, false); break;
    case 25: EGS_PREB(4, true); break;
    case 54: EGS_PREB(9, false); break;
    case 55: EGS_PREB(9, true); break;
    case 96: EGS_PREB(16, false); break;
    default: EGS_PREB(16, true); break;
  }
#undef EGS_PREB
#undef EGS_PREB_ARGS
  EGS_LAUNCH_OK();
  return 0;
}

extern "C" int egs_fused_backward(int n, int sh_dim, int64_t patches, int width, int height, const float* pws,
                                  const float* rots, const float* scales, const float* shs, const float* alphas,
                                  const float* Rcw, const float* tcw, const float* twc, float fx, float fy, float cx,
                                  float cy, const EgsPolicy* pol, const float* us, const float* cinv2ds,
                                  const float* colors, const int32_t* areas, const void* rec, const float* depths,
                                  const int32_t* contrib, const float* final_tau,
                                  const int32_t* patch_range_per_tile, const int32_t* gsid_per_patch,
                                  const float* dloss_dgammas, void* ws, size_t ws_bytes, float* dloss_dpws,
                                  float* dloss_dshs, float* dloss_dalphas, float* dloss_dscales,
                                  float* dloss_drots, float* dloss_dus, const int32_t* tile_order,
                                  float* grad_records, const float* dcolor_dpws, int phase, int row_begin,
                                  int row_count, void* seg_ws, size_t seg_ws_bytes, void* stream) {
  return fused_backward_impl(false, n, sh_dim, patches, width, height, pws, rots, scales, shs, nullptr, alphas, Rcw,
                             tcw, twc, fx, fy, cx, cy, pol, us, cinv2ds, colors, areas, rec, depths, contrib,
                             final_tau, patch_range_per_tile, gsid_per_patch, dloss_dgammas, ws, ws_bytes, dloss_dpws,
                             dloss_dshs, nullptr, dloss_dalphas, dloss_dscales, dloss_drots, dloss_dus, tile_order,
                             grad_records, dcolor_dpws, phase, row_begin, row_count, seg_ws, seg_ws_bytes, stream);
}

extern "C" int egs_fused_backward_raw(int n, int sh_dim, int64_t patches, int width, int height, const float* pws,
                                      const float* rots_raw, const float* scales_raw, const float* low_shs,
                                      const float* high_shs, const float* alphas_raw, const float* Rcw,
                                      const float* tcw, const float* twc, float fx, float fy, float cx, float cy,
                                      const EgsPolicy* pol, const float* us, const float* cinv2ds,
                                      const float* colors, const int32_t* areas, const void* rec,
                                      const float* depths, const int32_t* contrib, const float* final_tau,
                                      const int32_t* patch_range_per_tile, const int32_t* gsid_per_patch,
                                      const float* dloss_dgammas, void* ws, size_t ws_bytes, float* dloss_dpws,
                                      float* dloss_dlow_shs, float* dloss_dhigh_shs, float* dloss_dalphas_raw,
                                      float* dloss_dscales_raw, float* dloss_drots_raw, float* dloss_dus,
                                      const int32_t* tile_order, float* grad_records, const float* dcolor_dpws,
                                      int phase, int row_begin, int row_count, void* seg_ws, size_t seg_ws_bytes,
                                      void* stream) {
  return fused_backward_impl(true, n, sh_dim, patches, width, height, pws, rots_raw, scales_raw, low_shs, high_shs,
                             alphas_raw, Rcw, tcw, twc, fx, fy, cx, cy, pol, us, cinv2ds, colors, areas, rec, depths,
                             contrib, final_tau, patch_range_per_tile, gsid_per_patch, dloss_dgammas, ws, ws_bytes,
                             dloss_dpws, dloss_dlow_shs, dloss_dhigh_shs, dloss_dalphas_raw, dloss_dscales_raw,
                             dloss_drots_raw, dloss_dus, tile_order, grad_records, dcolor_dpws, phase, row_begin,
                             row_count, seg_ws, seg_ws_bytes, stream);
}

extern "C" int egs_sh_grad_views(int n, int sh_dim, int views, const float* pws, const float* rows,
                                 int64_t row_stride, float scale, float* dloss_dshs, float* dloss_dhigh_shs,
                                 int accumulate, void* stream) {
  EGS_CHECK_ARG(n >= 0 && views >= 0);
  EGS_CHECK_ARG(sh_dim == 3 || sh_dim == 12 || sh_dim == 27 || sh_dim == 48);
  if (n == 0) return 0;
  EGS_CHECK_ARG(pws && dloss_dshs && (views == 0 || rows) && row_stride >= 3 * (int64_t)n + 3);
  const bool raw = dloss_dhigh_shs != nullptr;
  EGS_CHECK_ARG((((uintptr_t)dloss_dshs | (uintptr_t)dloss_dhigh_shs) & 15) == 0);
  dim3 g(div_up(n, 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
#define EGS_SHV(NC, RAW)                                                                                     \
  EGS_LAUNCH("k_sh_grad_views", (k_sh_grad_views<NC, RAW>), g, b, s, n, views, pws, rows, row_stride, scale, \
             dloss_dshs, dloss_dhigh_shs, accumulate ? 1 : 0)
  switch (sh_dim * 2 + (raw ? 1 : 0)) {
    case 6: case 7: EGS_SHV(1, false); break;   // degree 0: the row IS the low part ([N][3] either way)
    case 24: EGS_SHV(4, false); break;
    case 25: EGS_SHV(4, true); break;
    case 54: EGS_SHV(9, false); break;
    case 55: EGS_SHV(9, true); break;
    case 96: EGS_SHV(16, false); break;
    default: EGS_SHV(16, true); break;
  }
#undef EGS_SHV
  EGS_LAUNCH_OK();
  return 0;
}
