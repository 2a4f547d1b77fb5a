// Per-Gaussian stages of the splatting pipeline for gfx950 (wave64):
//   project / cov3d / cov2d / sh2color / inv_cov2d  (+ optional Jacobians)
//   and the fused chain rule that consumes those Jacobians.
//
// Reference behaviour restated (not ported): gsplatcu/kernel.cu:274-807.  The
// reference multiplies dense Matrix<6,9>x<9,4> objects that are mostly zeros
// (kernel.cu:382-409, 512-537); here the block structure of those Jacobians is
// applied directly, so a thread needs ~40 live VGPRs instead of ~150.
//
// All kernels: one Gaussian per lane, 256-thread workgroups (4 waves), >= 3900
// workgroups at N = 1 M so all 256 CUs are covered many times over.  They are
// HBM-bound streaming kernels; algorithmic bytes per Gaussian are listed in
// DESIGN.md.
#include "egs_common.h"

namespace egs {

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// ----------------------------------------------------------------------------
// project: F.1.1 / F.1.2, near cull, B.1.2           (reference kernel.cu:553-617)
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_project(int n, const float* __restrict__ pws,
                                                 const float* __restrict__ Rcw,
                                                 const float* __restrict__ tcw, float fx, float fy,
                                                 float cx, float cy, int near_cull,
                                                 float* __restrict__ us, float* __restrict__ pcs,
                                                 float* __restrict__ depths,
                                                 float* __restrict__ du_dpcs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const f3 pw = ld3(pws + 3 * (size_t)i);
  const float x = Rcw[0] * pw.x + Rcw[1] * pw.y + Rcw[2] * pw.z + tcw[0];
  const float y = Rcw[3] * pw.x + Rcw[4] * pw.y + Rcw[5] * pw.z + tcw[1];
  const float z = Rcw[6] * pw.x + Rcw[7] * pw.y + Rcw[8] * pw.z + tcw[2];
  if (near_cull && z < EGS_MIN_DEPTH) {
    depths[i] = EGS_BAD_MARKER;  // everything else stays 0 (caller zero-fills)
    return;
  }
  const float z_inv = 1.f / z;
  const float z2_inv = z_inv * z_inv;
  const float xf = x * fx, yf = y * fy;
  us[2 * (size_t)i + 0] = xf * z_inv + cx;
  us[2 * (size_t)i + 1] = yf * z_inv + cy;
  pcs[3 * (size_t)i + 0] = x;
  pcs[3 * (size_t)i + 1] = y;
  pcs[3 * (size_t)i + 2] = z;
  depths[i] = z;
  if (du_dpcs) {
    float* J = du_dpcs + 6 * (size_t)i;  // entries 1,3 stay 0
    J[0] = fx * z_inv;
    J[2] = -xf * z2_inv;
    J[4] = fy * z_inv;
    J[5] = -yf * z2_inv;
  }
}

// ----------------------------------------------------------------------------
// cov3d: F.2, B.2a, B.2b                              (reference kernel.cu:326-423)
// ----------------------------------------------------------------------------
struct q4 { float w, x, y, z; };
// row-vector (3) times a 3x4 block given as three q4 rows
__device__ __forceinline__ q4 vm(f3 v, q4 r0, q4 r1, q4 r2) {
  return {v.x * r0.w + v.y * r1.w + v.z * r2.w, v.x * r0.x + v.y * r1.x + v.z * r2.x,
          v.x * r0.y + v.y * r1.y + v.z * r2.y, v.x * r0.z + v.y * r1.z + v.z * r2.z};
}
__device__ __forceinline__ q4 operator+(q4 a, q4 b) { return {a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ q4 operator*(float s, q4 a) { return {s * a.w, s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ void st4(float* p, q4 v) { p[0] = v.w; p[1] = v.x; p[2] = v.y; p[3] = v.z; }
__device__ __forceinline__ void st3(float* p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ f3 had(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }

__global__ __launch_bounds__(256) void k_cov3d(int n, const float* __restrict__ rots,
                                               const float* __restrict__ scales,
                                               const float* __restrict__ depths, int near_cull,
                                               float* __restrict__ cov3ds,
                                               float* __restrict__ dcov3d_drots,
                                               float* __restrict__ dcov3d_dscales) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (near_cull && depths[i] < EGS_MIN_DEPTH) return;
  const float4 q = *reinterpret_cast<const float4*>(rots + 4 * (size_t)i);  // 16-B aligned rows
  const float w = q.x, x = q.y, y = q.z, z = q.w;                          // (w,x,y,z); NOT normalised
  const f3 s = ld3(scales + 3 * (size_t)i);
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
  const float xw = x * w, yw = y * w, zw = z * w;
  const f3 R0 = {1.f - 2.f * (yy + zz), 2.f * (xy - zw), 2.f * (xz + yw)};
  const f3 R1 = {2.f * (xy + zw), 1.f - 2.f * (xx + zz), 2.f * (yz - xw)};
  const f3 R2 = {2.f * (xz - yw), 2.f * (yz + xw), 1.f - 2.f * (xx + yy)};
  const f3 M0 = had(R0, s), M1 = had(R1, s), M2 = had(R2, s);  // M = R diag(s)
  float* c = cov3ds + 6 * (size_t)i;                          // upper triangle of M M^T
  c[0] = dot(M0, M0); c[1] = dot(M0, M1); c[2] = dot(M0, M2);
  c[3] = dot(M1, M1); c[4] = dot(M1, M2); c[5] = dot(M2, M2);
  if (dcov3d_drots && dcov3d_dscales) {
    const float s0 = s.x, s1 = s.y, s2 = s.z;
    // dM/dq as three 3x4 blocks (rows of M); columns d/dw d/dx d/dy d/dz   (kernel.cu:388-396)
    const q4 A0 = {0.f, 0.f, -4 * s0 * y, -4 * s0 * z};
    const q4 A1 = {-2 * s1 * z, 2 * s1 * y, 2 * s1 * x, -2 * s1 * w};
    const q4 A2 = {2 * s2 * y, 2 * s2 * z, 2 * s2 * w, 2 * s2 * x};
    const q4 B0 = {2 * s0 * z, 2 * s0 * y, 2 * s0 * x, 2 * s0 * w};
    const q4 B1 = {0.f, -4 * s1 * x, 0.f, -4 * s1 * z};
    const q4 B2 = {-2 * s2 * x, -2 * s2 * w, 2 * s2 * z, 2 * s2 * y};
    const q4 C0 = {-2 * s0 * y, 2 * s0 * z, -2 * s0 * w, 2 * s0 * x};
    const q4 C1 = {2 * s1 * x, 2 * s1 * w, 2 * s1 * z, 2 * s1 * y};
    const q4 C2 = {0.f, -4 * s2 * x, -4 * s2 * y, 0.f};
    // d(MM^T)/dM has the block rows [2M0,0,0] [M1,M0,0] [M2,0,M0] [0,2M1,0] [0,M2,M1] [0,0,2M2]
    float* dq = dcov3d_drots + 24 * (size_t)i;
    st4(dq + 0, 2.f * vm(M0, A0, A1, A2));
    st4(dq + 4, vm(M1, A0, A1, A2) + vm(M0, B0, B1, B2));
    st4(dq + 8, vm(M2, A0, A1, A2) + vm(M0, C0, C1, C2));
    st4(dq + 12, 2.f * vm(M1, B0, B1, B2));
    st4(dq + 16, vm(M2, B0, B1, B2) + vm(M1, C0, C1, C2));
    st4(dq + 20, 2.f * vm(M2, C0, C1, C2));
    // dM/ds = diag(R0) | diag(R1) | diag(R2)                               (kernel.cu:397-405)
    float* ds = dcov3d_dscales + 18 * (size_t)i;
    st3(ds + 0, 2.f * had(M0, R0));
    st3(ds + 3, had(M1, R0) + had(M0, R1));
    st3(ds + 6, had(M2, R0) + had(M0, R2));
    st3(ds + 9, 2.f * had(M1, R1));
    st3(ds + 12, had(M2, R1) + had(M1, R2));
    st3(ds + 15, 2.f * had(M2, R2));
  }
}

// ----------------------------------------------------------------------------
// cov2d: F.3 (+0.3), B.3a, B.3b                        (reference kernel.cu:425-551)
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cov2d(int n, const float* __restrict__ cov3ds,
                                               const float* __restrict__ pcs,
                                               const float* __restrict__ Rcw,
                                               const float* __restrict__ depths, float fx, float fy,
                                               float limx, float limy, int clamp_fov, int near_cull,
                                               float* __restrict__ cov2ds,
                                               float* __restrict__ dcov2d_dcov3ds,
                                               float* __restrict__ dcov2d_dpcs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (near_cull && depths[i] < EGS_MIN_DEPTH) return;
  const f3 pc = ld3(pcs + 3 * (size_t)i);
  float x = pc.x, y = pc.y;
  const float z = pc.z;
  const float* cv = cov3ds + 6 * (size_t)i;
  const float a = cv[0], b = cv[1], c = cv[2], d = cv[3], e = cv[4], f = cv[5];
  if (clamp_fov) {  // the Jacobians below use the clamped x, y as the reference does
    x = fminf(limx, fmaxf(-limx, x / z)) * z;
    y = fminf(limy, fmaxf(-limy, y / z)) * z;
  }
  const float z2 = z * z;
  const f3 R0 = ld3(Rcw), R1 = ld3(Rcw + 3), R2 = ld3(Rcw + 6);
  const float j00 = fx / z, j02 = -(fx * x) / z2, j11 = fy / z, j12 = -(fy * y) / z2;
  const f3 M0 = j00 * R0 + j02 * R2;  // M = J Rcw (2x3)
  const f3 M1 = j11 * R1 + j12 * R2;
  // v = Sigma M^T
  const f3 v0 = {a * M0.x + b * M0.y + c * M0.z, b * M0.x + d * M0.y + e * M0.z, c * M0.x + e * M0.y + f * M0.z};
  const f3 v1 = {a * M1.x + b * M1.y + c * M1.z, b * M1.x + d * M1.y + e * M1.z, c * M1.x + e * M1.y + f * M1.z};
  float* o = cov2ds + 3 * (size_t)i;
  o[0] = dot(M0, v0) + 0.3f;
  o[1] = dot(M0, v1);
  o[2] = dot(M1, v1) + 0.3f;
  if (dcov2d_dcov3ds && dcov2d_dpcs) {
    float* J3 = dcov2d_dcov3ds + 18 * (size_t)i;  // B.3a  (kernel.cu:493-510)
    J3[0] = M0.x * M0.x; J3[1] = 2 * M0.x * M0.y; J3[2] = 2 * M0.x * M0.z;
    J3[3] = M0.y * M0.y; J3[4] = 2 * M0.y * M0.z; J3[5] = M0.z * M0.z;
    J3[6] = M0.x * M1.x; J3[7] = M0.x * M1.y + M0.y * M1.x; J3[8] = M0.x * M1.z + M0.z * M1.x;
    J3[9] = M0.y * M1.y; J3[10] = M0.y * M1.z + M0.z * M1.y; J3[11] = M0.z * M1.z;
    J3[12] = M1.x * M1.x; J3[13] = 2 * M1.x * M1.y; J3[14] = 2 * M1.x * M1.z;
    J3[15] = M1.y * M1.y; J3[16] = 2 * M1.y * M1.z; J3[17] = M1.z * M1.z;
    // B.3b: dcov2d/dM = [2v0,0 ; v1,v0 ; 0,2v1],  dM0/dpc = D0, dM1/dpc = D1 (kernel.cu:512-537)
    const float z2i = 1.f / z2, z3i = z2i / z;
    // D0 row k: [-fx R2k z2i, 0, -fx R0k z2i + 2 fx R2k x z3i];  D1 row k: [0, -fy R2k z2i, -fy R1k z2i + 2 fy R2k y z3i]
    const f3 d0c0 = (-fx * z2i) * R2;                                // column 0 of D0
    const f3 d0c2 = (-fx * z2i) * R0 + (2 * fx * x * z3i) * R2;      // column 2 of D0
    const f3 d1c1 = (-fy * z2i) * R2;                                // column 1 of D1
    const f3 d1c2 = (-fy * z2i) * R1 + (2 * fy * y * z3i) * R2;      // column 2 of D1
    float* Jp = dcov2d_dpcs + 9 * (size_t)i;
    Jp[0] = 2 * dot(v0, d0c0); Jp[1] = 0.f;               Jp[2] = 2 * dot(v0, d0c2);
    Jp[3] = dot(v1, d0c0);     Jp[4] = dot(v0, d1c1);     Jp[5] = dot(v1, d0c2) + dot(v0, d1c2);
    Jp[6] = 0.f;               Jp[7] = 2 * dot(v1, d1c1); Jp[8] = 2 * dot(v1, d1c2);
  }
}

// ----------------------------------------------------------------------------
// sh2color: F.4 and its Jacobians                      (reference kernel.cu:619-807)
// constants: reference common.cuh:28-43 == gsplat/sh_coef.py:5-23
// ----------------------------------------------------------------------------
#define SH_C0_0 0.28209479177387814f
#define SH_C1_0 (-0.4886025119029199f)
#define SH_C1_1 0.4886025119029199f
#define SH_C1_2 (-0.4886025119029199f)
#define SH_C2_0 1.0925484305920792f
#define SH_C2_1 (-1.0925484305920792f)
#define SH_C2_2 0.31539156525252005f
#define SH_C2_3 (-1.0925484305920792f)
#define SH_C2_4 0.5462742152960396f
#define SH_C3_0 (-0.5900435899266435f)
#define SH_C3_1 2.890611442640554f
#define SH_C3_2 (-0.4570457994644658f)
#define SH_C3_3 0.3731763325901154f
#define SH_C3_4 (-0.4570457994644658f)
#define SH_C3_5 1.445305721320277f
#define SH_C3_6 (-0.5900435899266435f)

// NC = number of SH coefficients per colour channel (1, 4, 9, 16)
template <int NC>
__global__ __launch_bounds__(256) void k_sh2color(int n, const float* __restrict__ shs,
                                                  const float* __restrict__ pws,
                                                  const float* __restrict__ twc,
                                                  float* __restrict__ colors,
                                                  float* __restrict__ dcolor_dshs,
                                                  float* __restrict__ dcolor_dpws) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  constexpr int K = 3 * NC;
  float sh[K];
  const float* row = shs + (size_t)K * i;
  if constexpr (K % 4 == 0) {  // 48- or 192-B rows: dwordx4 loads
#pragma unroll
    for (int j = 0; j < K / 4; ++j) {
      const float4 v = reinterpret_cast<const float4*>(row)[j];
      sh[4 * j] = v.x; sh[4 * j + 1] = v.y; sh[4 * j + 2] = v.z; sh[4 * j + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) sh[j] = row[j];
  }
  float B[NC];  // basis values == dcolor/dsh (shared by r,g,b)
  B[0] = SH_C0_0;
  float d0 = 0, d1 = 0, d2 = 0, ninv = 0, x = 0, y = 0, z = 0;
  float xx = 0, yy = 0, zz = 0, xy = 0, yz = 0, xz = 0;
  if constexpr (NC > 1) {
    const f3 pw = ld3(pws + 3 * (size_t)i);
    d0 = pw.x - twc[0]; d1 = pw.y - twc[1]; d2 = pw.z - twc[2];
    ninv = 1.f / sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    x = d0 * ninv; y = d1 * ninv; z = d2 * ninv;
    B[1] = SH_C1_0 * y; B[2] = SH_C1_1 * z; B[3] = SH_C1_2 * x;
  }
  if constexpr (NC > 4) {
    xx = x * x; yy = y * y; zz = z * z; xy = x * y; yz = y * z; xz = x * z;
    B[4] = SH_C2_0 * xy; B[5] = SH_C2_1 * yz; B[6] = SH_C2_2 * (2.0f * zz - xx - yy);
    B[7] = SH_C2_3 * xz; B[8] = SH_C2_4 * (xx - yy);
  }
  if constexpr (NC > 9) {
    B[9] = SH_C3_0 * y * (3.0f * xx - yy);
    B[10] = SH_C3_1 * xy * z;
    B[11] = SH_C3_2 * y * (4.0f * zz - xx - yy);
    B[12] = SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    B[13] = SH_C3_4 * x * (4.0f * zz - xx - yy);
    B[14] = SH_C3_5 * z * (xx - yy);
    B[15] = SH_C3_6 * x * (xx - 3.0f * yy);
  }
  float cr = 0.5f, cg = 0.5f, cb = 0.5f;  // no clamp to >= 0 (kernel.cu:652,725)
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    cr += B[c] * sh[3 * c]; cg += B[c] * sh[3 * c + 1]; cb += B[c] * sh[3 * c + 2];
  }
  float* co = colors + 3 * (size_t)i;
  co[0] = cr; co[1] = cg; co[2] = cb;

  if (dcolor_dshs && dcolor_dpws) {
    float* js = dcolor_dshs + (size_t)NC * i;
#pragma unroll
    for (int c = 0; c < NC; ++c) js[c] = B[c];
    float* jp = dcolor_dpws + 9 * (size_t)i;
    if constexpr (NC == 1) {
#pragma unroll
      for (int j = 0; j < 9; ++j) jp[j] = 0.f;
    } else {
      // dcolor[rgb]/ddir[xyz]: coefficient (per SH index) of each direction component
      float gx[NC], gy[NC], gz[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) { gx[c] = 0.f; gy[c] = 0.f; gz[c] = 0.f; }
      gx[3] = SH_C1_2; gy[1] = SH_C1_0; gz[2] = SH_C1_1;                       // kernel.cu:751-753
      if constexpr (NC > 4) {                                                  // kernel.cu:762-764
        gx[4] = SH_C2_0 * y; gx[6] = -SH_C2_2 * 2 * x; gx[7] = SH_C2_3 * z; gx[8] = SH_C2_4 * 2 * x;
        gy[4] = SH_C2_0 * x; gy[5] = SH_C2_1 * z; gy[6] = -SH_C2_2 * 2 * y; gy[8] = -SH_C2_4 * 2 * y;
        gz[5] = SH_C2_1 * y; gz[6] = SH_C2_2 * 4 * z; gz[7] = SH_C2_3 * x;
      }
      if constexpr (NC > 9) {                                                  // kernel.cu:776-793
        gx[9] = 6.0f * SH_C3_0 * xy; gx[10] = SH_C3_1 * yz; gx[11] = -2 * SH_C3_2 * xy;
        gx[12] = -6.0f * SH_C3_3 * xz; gx[13] = SH_C3_4 * (4.0f * zz - 3.0f * xx - yy);
        gx[14] = 2 * SH_C3_5 * xz; gx[15] = SH_C3_6 * (3 * xx - 3 * yy);
        gy[9] = SH_C3_0 * (3.0f * xx - 3.0f * yy); gy[10] = SH_C3_1 * xz;
        gy[11] = SH_C3_2 * (-xx - 3.0f * yy + 4.0f * zz); gy[12] = -6.0f * SH_C3_3 * yz;
        gy[13] = SH_C3_4 * (-2 * xy); gy[14] = -2 * SH_C3_5 * yz; gy[15] = -6.0f * SH_C3_6 * xy;
        gz[10] = SH_C3_1 * xy; gz[11] = 8.0f * SH_C3_2 * yz;
        gz[12] = SH_C3_3 * (-3.0f * xx - 3.0f * yy + 6.0f * zz); gz[13] = 8.0f * SH_C3_4 * xz;
        gz[14] = SH_C3_5 * (xx - yy);
      }
      float dr[3][3];  // dr[rgb][xyz]
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float sx = 0, sy = 0, sz = 0;
#pragma unroll
        for (int c = 1; c < NC; ++c) { sx += gx[c] * sh[3 * c + ch]; sy += gy[c] * sh[3 * c + ch]; sz += gz[c] * sh[3 * c + ch]; }
        dr[ch][0] = sx; dr[ch][1] = sy; dr[ch][2] = sz;
      }
      const float n3 = ninv * ninv * ninv;  // d dir / d pw (symmetric)     kernel.cu:738-745
      const float p00 = -d0 * d0 * n3 + ninv, p11 = -d1 * d1 * n3 + ninv, p22 = -d2 * d2 * n3 + ninv;
      const float p01 = -d0 * d1 * n3, p02 = -d0 * d2 * n3, p12 = -d1 * d2 * n3;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        jp[3 * ch + 0] = dr[ch][0] * p00 + dr[ch][1] * p01 + dr[ch][2] * p02;
        jp[3 * ch + 1] = dr[ch][0] * p01 + dr[ch][1] * p11 + dr[ch][2] * p12;
        jp[3 * ch + 2] = dr[ch][0] * p02 + dr[ch][1] * p12 + dr[ch][2] * p22;
      }
    }
  }
}

// ----------------------------------------------------------------------------
// inverse_cov2d: F.5.3, radius, B.5.3                  (reference kernel.cu:274-324)
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_inv_cov2d(int n, const float* __restrict__ cov2ds,
                                                   float* __restrict__ depths, float det_eps,
                                                   int near_cull, int nan_cull, int radius_mode,
                                                   float* __restrict__ cinv2ds,
                                                   int32_t* __restrict__ areas,
                                                   float* __restrict__ dcinv2d_dcov2ds) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (near_cull && depths[i] < EGS_MIN_DEPTH) return;
  const float a = cov2ds[3 * (size_t)i], b = cov2ds[3 * (size_t)i + 1], c = cov2ds[3 * (size_t)i + 2];
  const float det_inv = 1.f / (a * c - b * b + det_eps);
  if (nan_cull && isnan(det_inv)) {
    depths[i] = EGS_BAD_MARKER;  // in-place contract GSFunction relies on (gsmodel.py:50)
    return;
  }
  float* ci = cinv2ds + 3 * (size_t)i;
  ci[0] = det_inv * c; ci[1] = -det_inv * b; ci[2] = det_inv * a;
  int rx, ry;
  if (radius_mode == 0) {  // ceil(3 sqrt|a|)
    rx = (int)ceilf(3.f * sqrtf(fabsf(a)));
    ry = (int)ceilf(3.f * sqrtf(fabsf(c)));
  } else {                 // numpy astype(int32): truncation toward zero
    rx = (int)(3.f * sqrtf(a));
    ry = (int)(3.f * sqrtf(c));
  }
  areas[2 * (size_t)i] = rx;
  areas[2 * (size_t)i + 1] = ry;
  if (dcinv2d_dcov2ds) {
    const float d2 = det_inv * det_inv;
    float* J = dcinv2d_dcov2ds + 9 * (size_t)i;
    J[0] = -c * c * d2; J[1] = 2 * b * c * d2; J[2] = -a * c * d2 + det_inv;
    J[3] = b * c * d2; J[4] = -2 * b * b * d2 - det_inv; J[5] = a * b * d2;
    J[6] = -a * c * d2 + det_inv; J[7] = 2 * a * b * d2; J[8] = -a * a * d2;
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
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const f3 gci = ld3(dL_dcinv + 3 * (size_t)i);
  const float* A = J_cinv_cov2 + 9 * (size_t)i;
  // dL/dcov2d = dL/dcinv2d @ J (row vector times 3x3)
  const f3 gc2 = {gci.x * A[0] + gci.y * A[3] + gci.z * A[6], gci.x * A[1] + gci.y * A[4] + gci.z * A[7],
                  gci.x * A[2] + gci.y * A[5] + gci.z * A[8]};
  const float* Bm = J_cov2_cov3 + 18 * (size_t)i;
  float gc3[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) gc3[k] = gc2.x * Bm[k] + gc2.y * Bm[6 + k] + gc2.z * Bm[12 + k];
  const float* Cq = J_cov3_rot + 24 * (size_t)i;
  float* orot = dL_drot + 4 * (size_t)i;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 6; ++r) s += gc3[r] * Cq[4 * r + k];
    orot[k] = s;
  }
  const float* Cs = J_cov3_scale + 18 * (size_t)i;
  float* osc = dL_dscale + 3 * (size_t)i;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 6; ++r) s += gc3[r] * Cs[3 * r + k];
    osc[k] = s;
  }
  const f3 gcol = ld3(dL_dcolor + 3 * (size_t)i);
  const float* Bs = J_color_sh + (size_t)NC * i;
  float* osh = dL_dsh + (size_t)(3 * NC) * i;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float bc = Bs[c];
    osh[3 * c] = gcol.x * bc; osh[3 * c + 1] = gcol.y * bc; osh[3 * c + 2] = gcol.z * bc;
  }
  // dL/dpc = dL/du @ du_dpc + dL/dcov2d @ dcov2d_dpc ; dL/dpw = dL/dpc @ Rcw + dL/dcolor @ dcolor_dpw
  const float gu0 = dL_du[2 * (size_t)i], gu1 = dL_du[2 * (size_t)i + 1];
  const float* U = J_u_pc + 6 * (size_t)i;
  const float* Pc = J_cov2_pc + 9 * (size_t)i;
  const f3 gpc = {gu0 * U[0] + gu1 * U[3] + gc2.x * Pc[0] + gc2.y * Pc[3] + gc2.z * Pc[6],
                  gu0 * U[1] + gu1 * U[4] + gc2.x * Pc[1] + gc2.y * Pc[4] + gc2.z * Pc[7],
                  gu0 * U[2] + gu1 * U[5] + gc2.x * Pc[2] + gc2.y * Pc[5] + gc2.z * Pc[8]};
  const float* W = J_color_pw + 9 * (size_t)i;
  float* opw = dL_dpw + 3 * (size_t)i;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    opw[k] = gpc.x * Rcw[k] + gpc.y * Rcw[3 + k] + gpc.z * Rcw[6 + k] + gcol.x * W[k] + gcol.y * W[3 + k] +
             gcol.z * W[6 + k];
}

}  // namespace egs

// ============================================================================
// C ABI
// ============================================================================
using namespace egs;

extern "C" int egs_project(int n, const float* pws, const float* Rcw, const float* tcw, float fx, float fy,
                           float cx, float cy, const EgsPolicy* pol, float* us, float* pcs, float* depths,
                           float* du_dpcs, void* stream) {
  EGS_CHECK_ARG(n >= 0 && pol);
  if (n == 0) return 0;
  EGS_CHECK_ARG(pws && Rcw && tcw && us && pcs && depths);
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
  float limx = 0.f, limy = 0.f;
  if (pol->fov_mode == 0) {  // gausplat.cu:225-226
    limx = 1.3f * (width / (2 * fx));
    limy = 1.3f * (height / (2 * fy));
  } else if (pol->fov_mode == 1) {  // gausplat.py:136-140 (an angle, misnamed tan there)
    limx = (float)(1.3 * (2 * atan((double)width / (2 * (double)fx))));
    limy = (float)(1.3 * (2 * atan((double)height / (2 * (double)fy))));
  }
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
