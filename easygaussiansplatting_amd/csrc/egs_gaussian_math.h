// Per-Gaussian math of the splatting pipeline as register-resident device
// functions: values AND Jacobians (F.1-F.5.3, B.1.2-B.5.3 of the reference's
// docs/forward.md, docs/backward.md; behaviour of gsplatcu/kernel.cu:274-807).
// One source of truth for the seven-op surface (egs_preprocess.hip, Jacobians
// stored to HBM as the reference's ops do) and for the fused training path
// (Jacobians never leave registers).
//
// The reference multiplies dense Matrix<6,9>x<9,4> objects that are mostly zeros
// (kernel.cu:382-409, 512-537); here the block structure of those Jacobians is
// applied directly.
#pragma once
#include "egs_common.h"

namespace egs {

struct f3 { float x, y, z; };
struct q4 { float w, x, y, z; };
__device__ __forceinline__ f3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 had(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
// row-vector (3) times a 3x4 block given as three q4 rows
__device__ __forceinline__ q4 vm(f3 v, q4 r0, q4 r1, q4 r2) {
  return {v.x * r0.w + v.y * r1.w + v.z * r2.w, v.x * r0.x + v.y * r1.x + v.z * r2.x,
          v.x * r0.y + v.y * r1.y + v.z * r2.y, v.x * r0.z + v.y * r1.z + v.z * r2.z};
}
__device__ __forceinline__ q4 operator+(q4 a, q4 b) { return {a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ q4 operator*(float s, q4 a) { return {s * a.w, s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ void st4(float* p, q4 v) { p[0] = v.w; p[1] = v.x; p[2] = v.y; p[3] = v.z; }

// ---- project: F.1.1 / F.1.2, B.1.2                     (reference kernel.cu:553-617)
struct Proj {
  f3 pc;
  float u0, u1, z_inv;
};
__device__ __forceinline__ Proj project_f(f3 pw, const float* __restrict__ Rcw,
                                          const float* __restrict__ tcw, float fx, float fy, float cx,
                                          float cy) {
  Proj o;
  o.pc.x = Rcw[0] * pw.x + Rcw[1] * pw.y + Rcw[2] * pw.z + tcw[0];
  o.pc.y = Rcw[3] * pw.x + Rcw[4] * pw.y + Rcw[5] * pw.z + tcw[1];
  o.pc.z = Rcw[6] * pw.x + Rcw[7] * pw.y + Rcw[8] * pw.z + tcw[2];
  o.z_inv = 1.f / o.pc.z;
  o.u0 = (o.pc.x * fx) * o.z_inv + cx;
  o.u1 = (o.pc.y * fy) * o.z_inv + cy;
  return o;
}
// du/dpc (2x3): J = [j00 0 j02; 0 j11 j12]
__device__ __forceinline__ void project_jac(const Proj& p, float fx, float fy, float& j00, float& j02,
                                            float& j11, float& j12) {
  const float z2_inv = p.z_inv * p.z_inv;
  j00 = fx * p.z_inv;
  j02 = -(p.pc.x * fx) * z2_inv;
  j11 = fy * p.z_inv;
  j12 = -(p.pc.y * fy) * z2_inv;
}

// ---- cov3d: F.2, B.2a, B.2b                             (reference kernel.cu:326-423)
struct Cov3 {
  f3 R0, R1, R2;  // rotation from the (un-normalised) quaternion
  f3 M0, M1, M2;  // M = R diag(s)
  float c[6];     // upper triangle of M M^T
};
__device__ __forceinline__ Cov3 cov3d_f(float4 q, f3 s) {
  const float w = q.x, x = q.y, y = q.z, z = q.w;  // (w,x,y,z); NOT normalised (kernel.cu:342-347)
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
  const float xw = x * w, yw = y * w, zw = z * w;
  Cov3 o;
  o.R0 = {1.f - 2.f * (yy + zz), 2.f * (xy - zw), 2.f * (xz + yw)};
  o.R1 = {2.f * (xy + zw), 1.f - 2.f * (xx + zz), 2.f * (yz - xw)};
  o.R2 = {2.f * (xz - yw), 2.f * (yz + xw), 1.f - 2.f * (xx + yy)};
  o.M0 = had(o.R0, s); o.M1 = had(o.R1, s); o.M2 = had(o.R2, s);
  o.c[0] = dot(o.M0, o.M0); o.c[1] = dot(o.M0, o.M1); o.c[2] = dot(o.M0, o.M2);
  o.c[3] = dot(o.M1, o.M1); o.c[4] = dot(o.M1, o.M2); o.c[5] = dot(o.M2, o.M2);
  return o;
}
// dM/dq as three 3x4 blocks (rows of M); columns d/dw d/dx d/dy d/dz   (kernel.cu:388-396)
struct DMdq { q4 A0, A1, A2, B0, B1, B2, C0, C1, C2; };
__device__ __forceinline__ DMdq dm_dq(float4 q, f3 s) {
  const float w = q.x, x = q.y, y = q.z, z = q.w;
  const float s0 = s.x, s1 = s.y, s2 = s.z;
  DMdq d;
  d.A0 = {0.f, 0.f, -4 * s0 * y, -4 * s0 * z};
  d.A1 = {-2 * s1 * z, 2 * s1 * y, 2 * s1 * x, -2 * s1 * w};
  d.A2 = {2 * s2 * y, 2 * s2 * z, 2 * s2 * w, 2 * s2 * x};
  d.B0 = {2 * s0 * z, 2 * s0 * y, 2 * s0 * x, 2 * s0 * w};
  d.B1 = {0.f, -4 * s1 * x, 0.f, -4 * s1 * z};
  d.B2 = {-2 * s2 * x, -2 * s2 * w, 2 * s2 * z, 2 * s2 * y};
  d.C0 = {-2 * s0 * y, 2 * s0 * z, -2 * s0 * w, 2 * s0 * x};
  d.C1 = {2 * s1 * x, 2 * s1 * w, 2 * s1 * z, 2 * s1 * y};
  d.C2 = {0.f, -4 * s2 * x, -4 * s2 * y, 0.f};
  return d;
}
// full Jacobians dcov3d/dq [6x4] and dcov3d/ds [6x3], row-major
// d(MM^T)/dM has the block rows [2M0,0,0] [M1,M0,0] [M2,0,M0] [0,2M1,0] [0,M2,M1] [0,0,2M2]
__device__ __forceinline__ void cov3d_jac(const Cov3& c, float4 q, f3 s, float* dq /*24*/, float* ds /*18*/) {
  const DMdq d = dm_dq(q, s);
  st4(dq + 0, 2.f * vm(c.M0, d.A0, d.A1, d.A2));
  st4(dq + 4, vm(c.M1, d.A0, d.A1, d.A2) + vm(c.M0, d.B0, d.B1, d.B2));
  st4(dq + 8, vm(c.M2, d.A0, d.A1, d.A2) + vm(c.M0, d.C0, d.C1, d.C2));
  st4(dq + 12, 2.f * vm(c.M1, d.B0, d.B1, d.B2));
  st4(dq + 16, vm(c.M2, d.B0, d.B1, d.B2) + vm(c.M1, d.C0, d.C1, d.C2));
  st4(dq + 20, 2.f * vm(c.M2, d.C0, d.C1, d.C2));
  // dM/ds = diag(R0) | diag(R1) | diag(R2)                               (kernel.cu:397-405)
  st3(ds + 0, 2.f * had(c.M0, c.R0));
  st3(ds + 3, had(c.M1, c.R0) + had(c.M0, c.R1));
  st3(ds + 6, had(c.M2, c.R0) + had(c.M0, c.R2));
  st3(ds + 9, 2.f * had(c.M1, c.R1));
  st3(ds + 12, had(c.M2, c.R1) + had(c.M1, c.R2));
  st3(ds + 15, 2.f * had(c.M2, c.R2));
}
// vector-Jacobian product: g[6] = dL/dcov3d  ->  dL/dq, dL/ds, without forming the Jacobians:
// dL/dM rows = gA, gB, gC;  dL/dq = sum_rows vm(.,block), dL/ds = sum_rows had(., R_row)
__device__ __forceinline__ void cov3d_vjp(const Cov3& c, float4 q, f3 s, const float* g, q4& gq, f3& gs) {
  const f3 gA = (2.f * g[0]) * c.M0 + g[1] * c.M1 + g[2] * c.M2;
  const f3 gB = g[1] * c.M0 + (2.f * g[3]) * c.M1 + g[4] * c.M2;
  const f3 gC = g[2] * c.M0 + g[4] * c.M1 + (2.f * g[5]) * c.M2;
  const DMdq d = dm_dq(q, s);
  gq = vm(gA, d.A0, d.A1, d.A2) + vm(gB, d.B0, d.B1, d.B2) + vm(gC, d.C0, d.C1, d.C2);
  gs = had(gA, c.R0) + had(gB, c.R1) + had(gC, c.R2);
}

// ---- cov2d: F.3 (+0.3), B.3a, B.3b                      (reference kernel.cu:425-551)
struct Cov2 {
  f3 M0, M1;   // M = J Rcw (2x3)
  f3 v0, v1;   // Sigma M^T columns
  float x, y;  // the (fov-clamped) camera-space x, y the Jacobians use
  float c[3];  // cov2d incl. the +0.3
};
__device__ __forceinline__ Cov2 cov2d_f(const float* cv, f3 pc, const float* __restrict__ Rcw, float fx,
                                        float fy, float limx, float limy, int clamp_fov) {
  Cov2 o;
  o.x = pc.x; o.y = pc.y;
  const float z = pc.z;
  if (clamp_fov) {
    o.x = fminf(limx, fmaxf(-limx, o.x / z)) * z;
    o.y = fminf(limy, fmaxf(-limy, o.y / z)) * z;
  }
  const float z2 = z * z;
  const f3 R0 = ld3(Rcw), R1 = ld3(Rcw + 3), R2 = ld3(Rcw + 6);
  const float j00 = fx / z, j02 = -(fx * o.x) / z2, j11 = fy / z, j12 = -(fy * o.y) / z2;
  o.M0 = j00 * R0 + j02 * R2;
  o.M1 = j11 * R1 + j12 * R2;
  const float a = cv[0], b = cv[1], c = cv[2], d = cv[3], e = cv[4], f = cv[5];
  o.v0 = {a * o.M0.x + b * o.M0.y + c * o.M0.z, b * o.M0.x + d * o.M0.y + e * o.M0.z,
          c * o.M0.x + e * o.M0.y + f * o.M0.z};
  o.v1 = {a * o.M1.x + b * o.M1.y + c * o.M1.z, b * o.M1.x + d * o.M1.y + e * o.M1.z,
          c * o.M1.x + e * o.M1.y + f * o.M1.z};
  o.c[0] = dot(o.M0, o.v0) + 0.3f;
  o.c[1] = dot(o.M0, o.v1);
  o.c[2] = dot(o.M1, o.v1) + 0.3f;
  return o;
}
// dcov2d/dcov3d [3x6] (B.3a, kernel.cu:493-510) and dcov2d/dpc [3x3] (B.3b, kernel.cu:512-537)
__device__ __forceinline__ void cov2d_jac(const Cov2& o, float z, const float* __restrict__ Rcw, float fx,
                                          float fy, float* J3 /*18*/, float* Jp /*9*/) {
  const f3 M0 = o.M0, M1 = o.M1;
  J3[0] = M0.x * M0.x; J3[1] = 2 * M0.x * M0.y; J3[2] = 2 * M0.x * M0.z;
  J3[3] = M0.y * M0.y; J3[4] = 2 * M0.y * M0.z; J3[5] = M0.z * M0.z;
  J3[6] = M0.x * M1.x; J3[7] = M0.x * M1.y + M0.y * M1.x; J3[8] = M0.x * M1.z + M0.z * M1.x;
  J3[9] = M0.y * M1.y; J3[10] = M0.y * M1.z + M0.z * M1.y; J3[11] = M0.z * M1.z;
  J3[12] = M1.x * M1.x; J3[13] = 2 * M1.x * M1.y; J3[14] = 2 * M1.x * M1.z;
  J3[15] = M1.y * M1.y; J3[16] = 2 * M1.y * M1.z; J3[17] = M1.z * M1.z;
  // dcov2d/dM = [2v0,0 ; v1,v0 ; 0,2v1],  dM0/dpc = D0, dM1/dpc = D1
  const f3 R0 = ld3(Rcw), R1 = ld3(Rcw + 3), R2 = ld3(Rcw + 6);
  const float z2i = 1.f / (z * z), z3i = z2i / z;
  const f3 d0c0 = (-fx * z2i) * R2;                            // column 0 of D0
  const f3 d0c2 = (-fx * z2i) * R0 + (2 * fx * o.x * z3i) * R2;  // column 2 of D0
  const f3 d1c1 = (-fy * z2i) * R2;                            // column 1 of D1
  const f3 d1c2 = (-fy * z2i) * R1 + (2 * fy * o.y * z3i) * R2;  // column 2 of D1
  Jp[0] = 2 * dot(o.v0, d0c0); Jp[1] = 0.f;                 Jp[2] = 2 * dot(o.v0, d0c2);
  Jp[3] = dot(o.v1, d0c0);     Jp[4] = dot(o.v0, d1c1);     Jp[5] = dot(o.v1, d0c2) + dot(o.v0, d1c2);
  Jp[6] = 0.f;                 Jp[7] = 2 * dot(o.v1, d1c1); Jp[8] = 2 * dot(o.v1, d1c2);
}

// ---- inverse_cov2d: F.5.3, radius, B.5.3                (reference kernel.cu:274-324)
__device__ __forceinline__ float inv_cov2d_f(const float* c2, float det_eps, float* cinv) {
  const float det_inv = 1.f / (c2[0] * c2[2] - c2[1] * c2[1] + det_eps);
  cinv[0] = det_inv * c2[2]; cinv[1] = -det_inv * c2[1]; cinv[2] = det_inv * c2[0];
  return det_inv;
}
__device__ __forceinline__ void radius_f(const float* c2, int radius_mode, int& rx, int& ry) {
  if (radius_mode == 0) {  // ceil(3 sqrt|a|)  (kernel.cu:308)
    rx = (int)ceilf(3.f * sqrtf(fabsf(c2[0])));
    ry = (int)ceilf(3.f * sqrtf(fabsf(c2[2])));
  } else {                 // numpy astype(int32): truncation toward zero (gausplat.py:181-182)
    rx = (int)(3.f * sqrtf(c2[0]));
    ry = (int)(3.f * sqrtf(c2[2]));
  }
}
__device__ __forceinline__ void inv_cov2d_jac(const float* c2, float det_inv, float* J /*9*/) {
  const float a = c2[0], b = c2[1], c = c2[2], d2 = det_inv * det_inv;
  J[0] = -c * c * d2; J[1] = 2 * b * c * d2; J[2] = -a * c * d2 + det_inv;
  J[3] = b * c * d2; J[4] = -2 * b * b * d2 - det_inv; J[5] = a * b * d2;
  J[6] = -a * c * d2 + det_inv; J[7] = 2 * a * b * d2; J[8] = -a * a * d2;
}

// ---- sh2color: F.4 and its Jacobians                    (reference kernel.cu:619-807)
// constants: reference common.cuh:28-43 == gsplat/sh_coef.py:5-23
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

template <int NC>
struct ShDir {
  float d0, d1, d2, ninv, x, y, z;
  float B[NC];  // basis values == dcolor/dsh (shared by r,g,b)
};
// NC = number of SH coefficients per colour channel (1, 4, 9, 16)
template <int NC>
__device__ __forceinline__ ShDir<NC> sh_basis_f(f3 pw, const float* __restrict__ twc) {
  ShDir<NC> o;
  o.d0 = 0; o.d1 = 0; o.d2 = 0; o.ninv = 0; o.x = 0; o.y = 0; o.z = 0;
  o.B[0] = SH_C0_0;
  if constexpr (NC > 1) {
    o.d0 = pw.x - twc[0]; o.d1 = pw.y - twc[1]; o.d2 = pw.z - twc[2];
    o.ninv = 1.f / sqrtf(o.d0 * o.d0 + o.d1 * o.d1 + o.d2 * o.d2);
    o.x = o.d0 * o.ninv; o.y = o.d1 * o.ninv; o.z = o.d2 * o.ninv;
    o.B[1] = SH_C1_0 * o.y; o.B[2] = SH_C1_1 * o.z; o.B[3] = SH_C1_2 * o.x;
  }
  if constexpr (NC > 4) {
    const float x = o.x, y = o.y, z = o.z;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    o.B[4] = SH_C2_0 * xy; o.B[5] = SH_C2_1 * yz; o.B[6] = SH_C2_2 * (2.0f * zz - xx - yy);
    o.B[7] = SH_C2_3 * xz; o.B[8] = SH_C2_4 * (xx - yy);
    if constexpr (NC > 9) {
      o.B[9] = SH_C3_0 * y * (3.0f * xx - yy);
      o.B[10] = SH_C3_1 * xy * z;
      o.B[11] = SH_C3_2 * y * (4.0f * zz - xx - yy);
      o.B[12] = SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
      o.B[13] = SH_C3_4 * x * (4.0f * zz - xx - yy);
      o.B[14] = SH_C3_5 * z * (xx - yy);
      o.B[15] = SH_C3_6 * x * (xx - 3.0f * yy);
    }
  }
  return o;
}
template <int NC>
__device__ __forceinline__ void sh_color_f(const ShDir<NC>& o, const float* sh, float* col) {
  float cr = 0.5f, cg = 0.5f, cb = 0.5f;  // no clamp to >= 0 (kernel.cu:652,725)
#pragma unroll
  for (int c = 0; c < NC; ++c) { cr += o.B[c] * sh[3 * c]; cg += o.B[c] * sh[3 * c + 1]; cb += o.B[c] * sh[3 * c + 2]; }
  col[0] = cr; col[1] = cg; col[2] = cb;
}
// dcolor[rgb]/ddir[xyz] (dr) -- kernel.cu:751-793
template <int NC>
__device__ __forceinline__ void sh_dcolor_ddir(const ShDir<NC>& o, const float* sh, float dr[3][3]) {
  const float x = o.x, y = o.y, z = o.z;
  float gx[NC], gy[NC], gz[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { gx[c] = 0.f; gy[c] = 0.f; gz[c] = 0.f; }
  if constexpr (NC > 1) { gx[3] = SH_C1_2; gy[1] = SH_C1_0; gz[2] = SH_C1_1; }
  if constexpr (NC > 4) {
    gx[4] = SH_C2_0 * y; gx[6] = -SH_C2_2 * 2 * x; gx[7] = SH_C2_3 * z; gx[8] = SH_C2_4 * 2 * x;
    gy[4] = SH_C2_0 * x; gy[5] = SH_C2_1 * z; gy[6] = -SH_C2_2 * 2 * y; gy[8] = -SH_C2_4 * 2 * y;
    gz[5] = SH_C2_1 * y; gz[6] = SH_C2_2 * 4 * z; gz[7] = SH_C2_3 * x;
  }
  if constexpr (NC > 9) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
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
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float sx = 0, sy = 0, sz = 0;
#pragma unroll
    for (int c = 1; c < NC; ++c) { sx += gx[c] * sh[3 * c + ch]; sy += gy[c] * sh[3 * c + ch]; sz += gz[c] * sh[3 * c + ch]; }
    dr[ch][0] = sx; dr[ch][1] = sy; dr[ch][2] = sz;
  }
}
// ddir/dpw (symmetric 3x3; kernel.cu:738-745) as 6 unique entries
template <int NC>
__device__ __forceinline__ void sh_ddir_dpw(const ShDir<NC>& o, float& p00, float& p11, float& p22, float& p01,
                                            float& p02, float& p12) {
  const float n3 = o.ninv * o.ninv * o.ninv;
  p00 = -o.d0 * o.d0 * n3 + o.ninv; p11 = -o.d1 * o.d1 * n3 + o.ninv; p22 = -o.d2 * o.d2 * n3 + o.ninv;
  p01 = -o.d0 * o.d1 * n3; p02 = -o.d0 * o.d2 * n3; p12 = -o.d1 * o.d2 * n3;
}
// dcolor/dpw [3x3] row-major
template <int NC>
__device__ __forceinline__ void sh_jac_dpw(const ShDir<NC>& o, const float* sh, float* jp /*9*/) {
  if constexpr (NC == 1) {
#pragma unroll
    for (int j = 0; j < 9; ++j) jp[j] = 0.f;
  } else {
    float dr[3][3];
    sh_dcolor_ddir<NC>(o, sh, dr);
    float p00, p11, p22, p01, p02, p12;
    sh_ddir_dpw<NC>(o, p00, p11, p22, p01, p02, p12);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      jp[3 * ch + 0] = dr[ch][0] * p00 + dr[ch][1] * p01 + dr[ch][2] * p02;
      jp[3 * ch + 1] = dr[ch][0] * p01 + dr[ch][1] * p11 + dr[ch][2] * p12;
      jp[3 * ch + 2] = dr[ch][0] * p02 + dr[ch][1] * p12 + dr[ch][2] * p22;
    }
  }
}

// basis value of coefficient C from the direction alone, evaluated with NO contraction: the same bits in every
// instantiation that inlines it (sh_color_and_jac_dpw<.., INLINE_B> with and without the Jacobian sums)
#pragma clang fp contract(off)
template <int C>
__device__ __forceinline__ float sh_basis_term(float x, float y, float z) {
  if constexpr (C == 0) return SH_C0_0;
  else if constexpr (C == 1) return SH_C1_0 * y;
  else if constexpr (C == 2) return SH_C1_1 * z;
  else if constexpr (C == 3) return SH_C1_2 * x;
  else if constexpr (C == 4) return SH_C2_0 * (x * y);
  else if constexpr (C == 5) return SH_C2_1 * (y * z);
  else if constexpr (C == 6) return SH_C2_2 * (2.0f * (z * z) - x * x - y * y);
  else if constexpr (C == 7) return SH_C2_3 * (x * z);
  else if constexpr (C == 8) return SH_C2_4 * (x * x - y * y);
  else if constexpr (C == 9) return SH_C3_0 * y * (3.0f * (x * x) - y * y);
  else if constexpr (C == 10) return SH_C3_1 * (x * y) * z;
  else if constexpr (C == 11) return SH_C3_2 * y * (4.0f * (z * z) - x * x - y * y);
  else if constexpr (C == 12) return SH_C3_3 * z * (2.0f * (z * z) - 3.0f * (x * x) - 3.0f * (y * y));
  else if constexpr (C == 13) return SH_C3_4 * x * (4.0f * (z * z) - x * x - y * y);
  else if constexpr (C == 14) return SH_C3_5 * z * (x * x - y * y);
  else return SH_C3_6 * x * (x * x - 3.0f * (y * y));
}
#pragma clang fp contract(fast)
// colour AND dcolor/dpw in ONE pass over the coefficients (round 4): every SH value is consumed by both sums the
// moment it is first touched and its register is free afterwards -- with sh_color_f followed by sh_jac_dpw the whole
// 4K-byte row stayed live through the first sum (k_preprocess_fwd<.., JW>: 100 VGPRs / 5 waves per SIMD, k_sh2color:
// 106 / 4).  Same terms in the same order as the two functions it replaces (terms with a zero gradient are left out).
// INLINE_B: the basis value of a term is re-evaluated from the direction where it is used instead of read from o.B --
// for a caller that has already stored o.B (k_sh2color's dcolor_dshs output) and wants its 16 registers back.
// JAC = false: the colour sum alone, through the SAME expressions (k_sh2color's calc_J = false path then rounds like its
// calc_J = true path).
template <int NC, bool INLINE_B = false, bool JAC = true>
__device__ __forceinline__ void sh_color_and_jac_dpw(const ShDir<NC>& o, const float* sh, float* col, float* jp /*9*/) {
  if constexpr (NC == 1) {
    sh_color_f<NC>(o, sh, col);
#pragma unroll
    for (int j = 0; j < 9; ++j) jp[j] = 0.f;
  } else {
    const float x = o.x, y = o.y, z = o.z;
    float cc[3] = {0.5f, 0.5f, 0.5f};
    float dr[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    // one coefficient: colour += B s, d colour / d dir += (gx, gy, gz) s  (compile-time flags drop the zero gradients)
#define EGS_SH_TERM(c, BEXPR, HX, GX, HY, GY, HZ, GZ)                                   \
    do {                                                                                \
      const float b_ = INLINE_B ? sh_basis_term<c>(x, y, z) : o.B[c];                   \
      _Pragma("unroll") for (int ch = 0; ch < 3; ++ch) {                                \
        const float s_ = sh[3 * (c) + ch];                                              \
        cc[ch] = __builtin_fmaf(b_, s_, cc[ch]);     /* (explicit: the same colour bits with and without JAC) */ \
        if (JAC && (HX)) dr[ch][0] += (GX) * s_;                                        \
        if (JAC && (HY)) dr[ch][1] += (GY) * s_;                                        \
        if (JAC && (HZ)) dr[ch][2] += (GZ) * s_;                                        \
      }                                                                                 \
    } while (0)
    EGS_SH_TERM(0, SH_C0_0, false, 0.f, false, 0.f, false, 0.f);
    EGS_SH_TERM(1, SH_C1_0 * y, false, 0.f, true, SH_C1_0, false, 0.f);
    EGS_SH_TERM(2, SH_C1_1 * z, false, 0.f, false, 0.f, true, SH_C1_1);
    EGS_SH_TERM(3, SH_C1_2 * x, true, SH_C1_2, false, 0.f, false, 0.f);
    if constexpr (NC > 4) {
      EGS_SH_TERM(4, SH_C2_0 * (x * y), true, SH_C2_0 * y, true, SH_C2_0 * x, false, 0.f);
      EGS_SH_TERM(5, SH_C2_1 * (y * z), false, 0.f, true, SH_C2_1 * z, true, SH_C2_1 * y);
      EGS_SH_TERM(6, SH_C2_2 * (2.0f * (z * z) - x * x - y * y), true, -SH_C2_2 * 2 * x, true, -SH_C2_2 * 2 * y, true, SH_C2_2 * 4 * z);
      EGS_SH_TERM(7, SH_C2_3 * (x * z), true, SH_C2_3 * z, false, 0.f, true, SH_C2_3 * x);
      EGS_SH_TERM(8, SH_C2_4 * (x * x - y * y), true, SH_C2_4 * 2 * x, true, -SH_C2_4 * 2 * y, false, 0.f);
    }
    if constexpr (NC > 9) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      EGS_SH_TERM(9, SH_C3_0 * y * (3.0f * xx - yy), true, 6.0f * SH_C3_0 * xy, true, SH_C3_0 * (3.0f * xx - 3.0f * yy), false, 0.f);
      EGS_SH_TERM(10, SH_C3_1 * xy * z, true, SH_C3_1 * yz, true, SH_C3_1 * xz, true, SH_C3_1 * xy);
      EGS_SH_TERM(11, SH_C3_2 * y * (4.0f * zz - xx - yy), true, -2 * SH_C3_2 * xy, true, SH_C3_2 * (-xx - 3.0f * yy + 4.0f * zz), true, 8.0f * SH_C3_2 * yz);
      EGS_SH_TERM(12, SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), true, -6.0f * SH_C3_3 * xz, true, -6.0f * SH_C3_3 * yz, true,
                  SH_C3_3 * (-3.0f * xx - 3.0f * yy + 6.0f * zz));
      EGS_SH_TERM(13, SH_C3_4 * x * (4.0f * zz - xx - yy), true, SH_C3_4 * (4.0f * zz - 3.0f * xx - yy), true, SH_C3_4 * (-2 * xy), true, 8.0f * SH_C3_4 * xz);
      EGS_SH_TERM(14, SH_C3_5 * z * (xx - yy), true, 2 * SH_C3_5 * xz, true, -2 * SH_C3_5 * yz, true, SH_C3_5 * (xx - yy));
      EGS_SH_TERM(15, SH_C3_6 * x * (xx - 3.0f * yy), true, SH_C3_6 * (3 * xx - 3 * yy), true, -6.0f * SH_C3_6 * xy, false, 0.f);
    }
#undef EGS_SH_TERM
    col[0] = cc[0]; col[1] = cc[1]; col[2] = cc[2];
    if constexpr (JAC) {
      float p00, p11, p22, p01, p02, p12;
      sh_ddir_dpw<NC>(o, p00, p11, p22, p01, p02, p12);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        jp[3 * ch + 0] = dr[ch][0] * p00 + dr[ch][1] * p01 + dr[ch][2] * p02;
        jp[3 * ch + 1] = dr[ch][0] * p01 + dr[ch][1] * p11 + dr[ch][2] * p12;
        jp[3 * ch + 2] = dr[ch][0] * p02 + dr[ch][1] * p12 + dr[ch][2] * p22;
      }
    }
  }
}

#ifndef EGS_SH_NT_LOAD
#define EGS_SH_NT_LOAD 0
#endif
template <int K>
__device__ __forceinline__ void load_sh_row(const float* __restrict__ row, float* sh) {
  if constexpr (K % 4 == 0) {  // 48- or 192-B rows: dwordx4 loads
#pragma unroll
    for (int j = 0; j < K / 4; ++j) {
#if EGS_SH_NT_LOAD        // A/B knob: streaming (non-temporal) loads of the SH rows -- measured k_preprocess_fwd 99 -> 172 us,
                          // k_sh2color 86 -> 132 us: the 12 loads of a lane re-touch its lines and live on the cache hits
      typedef float f4v_ __attribute__((ext_vector_type(4)));
      const f4v_ v = __builtin_nontemporal_load(reinterpret_cast<const f4v_*>(row) + j);
#else
      const float4 v = reinterpret_cast<const float4*>(row)[j];
#endif
      sh[4 * j] = v.x; sh[4 * j + 1] = v.y; sh[4 * j + 2] = v.z; sh[4 * j + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) sh[j] = row[j];
  }
}

// ---- packed 2D record of the draw kernels (layout: see egs_bin.hip k_pack_records) ----
#define EGS_NHL2E (-0.72134752044f)  // -0.5 * log2(e)
// saturating float -> int (v_cvt_i32_f32 semantics; NaN -> 0)
__device__ __forceinline__ int f2i(float v) { return (int)v; }
// pixel box of gausplat.py:212-215
__device__ __forceinline__ void pixel_box(float ux, float uy, float rx, float ry, int W, int H, int& x0,
                                          int& x1, int& y0, int& y1) {
  x0 = f2i(fmaxf(fminf(ux - rx, (float)W), 0.f));
  x1 = f2i(fmaxf(fminf(ux + rx, (float)W), 0.f));
  y0 = f2i(fmaxf(fminf(uy - ry, (float)H), 0.f));
  y1 = f2i(fmaxf(fminf(uy + ry, (float)H), 0.f));
}
// getRects (reference kernel.cu:82-122) + the depth key of createKeys (kernel.cu:73) for one Gaussian.
// Returns the patch count; `cull` = the reference's in-place marking of a Gaussian whose tile rect is
// empty (depth = -1, areas = 0; kernel.cu:114-119) applies.
__device__ __forceinline__ uint32_t bin_count_one(const BinParams& p, float ux, float uy, float xs, float ys,
                                                  float depth, uint4& rect, uint32_t& key, bool& cull) {
  uint32_t cnt = 0;
  rect = {0u, 0u, 0u, 0u};
  key = 0u;  // culled Gaussians emit nothing: any key will do, 0 keeps the maximum small
  cull = false;
  if (p.footprint == 0) {
    if (!(depth < EGS_MIN_DEPTH)) {
      const float B = (float)EGS_TILE;
      const int x0 = min(p.gx, max(0, f2i((ux - xs) / B)));
      const int y0 = min(p.gy, max(0, f2i((uy - ys) / B)));
      const int x1 = min(p.gx, max(0, f2i((ux + xs + B - 1.f) / B)));  // DIV_ROUND_UP in float (common.cuh:14)
      const int y1 = min(p.gy, max(0, f2i((uy + ys + B - 1.f) / B)));
      // (a reversed rect -- only possible with negative radii fed by the caller --
      //  would wrap in the reference's unsigned product; it is treated as empty)
      cnt = (x1 > x0 && y1 > y0) ? (uint32_t)(y1 - y0) * (uint32_t)(x1 - x0) : 0u;
      if (cnt == 0) cull = p.mutate != 0;
      else rect = {(uint32_t)x0, (uint32_t)y0, (uint32_t)x1, (uint32_t)y1};
    }
  } else {
    bool vis = !(depth < 0.2f || depth > 100.f);                                        // gausplat.py:204
    vis = vis && !(fabsf(ux / (float)p.W) > 1.3f) && !(fabsf(uy / (float)p.H) > 1.3f);  // gausplat.py:208
    if (vis) {
      int x0, x1, y0, y1;
      pixel_box(ux, uy, xs, ys, p.W, p.H, x0, x1, y0, y1);
      if ((x1 - x0) * (y1 - y0) != 0 && x1 > x0 && y1 > y0) {
        rect = {(uint32_t)(x0 / EGS_TILE), (uint32_t)(y0 / EGS_TILE), (uint32_t)((x1 + EGS_TILE - 1) / EGS_TILE),
                (uint32_t)((y1 + EGS_TILE - 1) / EGS_TILE)};
        cnt = (rect.w - rect.y) * (rect.z - rect.x);
      }
    }
  }
  if (cnt != 0) key = (p.depth_key == 0) ? (uint32_t)(depth * 1000.f) : __float_as_uint(depth);
  return cnt;
}

// ---- exact footprint of a Gaussian on the tile grid (culled lists of the fused path) ---------------------------
// One source for the COUNT (k_preprocess_fwd: how many tiles the Gaussian is emitted for) and the EMISSION
// (k_bin_emit: which ones): both evaluate these functions on the same stored BinRec, so they must be bit-reproducible
// across kernels -- no FMA contraction (pragma below), hardware rcp / sqrt (one instruction each, the same bits
// wherever they are issued).  Conservative by construction: the x-extent of {footprint ellipse} n {8-pixel slab} is
// exact in real arithmetic, the slab is taken continuous in y, and every bound is widened by `eps` (1 % of the
// extent + 0.05 px, the slack of the draw kernels' own certain-miss box) before it is rounded to pixel centres.
struct Foot {
  float ux, uy, A, Bh, C, m;
  float det, rA, ymax, k, eps;
  int px_lo, px_hi;   // pixel range of the tile rect in x
  bool cull;          // false: emit the whole rect, every block reachable
};
struct SlabPx { int pl, pr; };   // pixel centres [pl, pr] of a slab inside the footprint; empty: pl > pr
#pragma clang fp contract(off)
__device__ __forceinline__ Foot foot_setup(const BinRec& b) {
  Foot f;
  f.ux = b.ux; f.uy = b.uy; f.A = b.A; f.Bh = b.Bh; f.C = b.C; f.m = b.m;
  f.cull = b.m < __int_as_float(0x7f800000);
  const int x0 = (int)(b.xy & 0xFFFFu), w = (int)(b.wh & 0xFFFFu);
  f.px_lo = x0 * EGS_TILE;
  f.px_hi = (x0 + w) * EGS_TILE - 1;
  f.det = f.A * f.C - f.Bh * f.Bh;
  const float rdet = __builtin_amdgcn_rcpf(f.det);
  f.rA = __builtin_amdgcn_rcpf(f.A);
  const float xext = __builtin_amdgcn_sqrtf(f.m * f.C * rdet);   // largest |dx| of the footprint
  f.ymax = __builtin_amdgcn_sqrtf(f.m * f.A * rdet);             // largest |dy|
  f.k = f.Bh * __builtin_amdgcn_rcpf(f.C) * xext;                // the rightmost point sits at dy = -k, the leftmost at +k
  f.eps = 0.05f + 0.01f * fmaxf(xext, f.ymax);
  return f;
}
// pixel centres of rows [Y0, Y0 + 7] that can lie inside the footprint
__device__ __forceinline__ SlabPx foot_slab(const Foot& f, int Y0) {
  SlabPx o;
  o.pl = 0x7fffffff; o.pr = (int)0x80000000;
  const float ya = fmaxf((float)Y0 - f.uy - f.eps, -f.ymax - f.eps);
  const float yb = fminf((float)(Y0 + 7) - f.uy + f.eps, f.ymax + f.eps);
  if (!(ya <= yb)) return o;
  const float yr = fminf(fmaxf(-f.k, ya), yb), yl = fminf(fmaxf(f.k, ya), yb);
  const float Am = f.A * f.m;
  const float sr = __builtin_amdgcn_sqrtf(fmaxf(Am - f.det * yr * yr, 0.f));
  const float sl = __builtin_amdgcn_sqrtf(fmaxf(Am - f.det * yl * yl, 0.f));
  const float xr = f.ux + ((sr - f.Bh * yr) * f.rA + f.eps);
  const float xl = f.ux + ((-sl - f.Bh * yl) * f.rA - f.eps);
  if (!(xl <= xr)) return o;     // (NaN: cannot happen for a footprint that passed foot_cullable; stay empty)
  o.pl = max(f.px_lo, (int)ceilf(fmaxf(xl, -1.0e9f)));
  o.pr = min(f.px_hi, (int)floorf(fminf(xr, 1.0e9f)));
  if (o.pl > o.pr) { o.pl = 0x7fffffff; o.pr = (int)0x80000000; }
  return o;
}
// tiles [lo, hi] of tile row `ty` the Gaussian is emitted for (lo > hi: none) and the two slabs of that row
__device__ __forceinline__ void foot_row(const Foot& f, int ty, SlabPx& s0, SlabPx& s1, int& lo, int& hi) {
  s0 = foot_slab(f, ty * EGS_TILE);
  s1 = foot_slab(f, ty * EGS_TILE + 8);
  lo = min(s0.pl, s1.pl) >> 4;          // (empty slabs hold +-INT extremes: min / max ignore them)
  hi = max(s0.pr, s1.pr) >> 4;
  if (s0.pl > s0.pr && s1.pl > s1.pr) { lo = 1; hi = 0; }
}
// 4-bit reach mask of tile column tx (bit k = block (k&1, k>>1))
__device__ __forceinline__ uint32_t foot_mask(const SlabPx& s0, const SlabPx& s1, int tx) {
  const int X = tx * EGS_TILE;
  uint32_t m = 0u;
  if (s0.pl <= X + 7 && s0.pr >= X) m |= 1u;
  if (s0.pl <= X + 15 && s0.pr >= X + 8) m |= 2u;
  if (s1.pl <= X + 7 && s1.pr >= X) m |= 4u;
  if (s1.pl <= X + 15 && s1.pr >= X + 8) m |= 8u;
  return m;
}
// number of tiles the Gaussian is emitted for (the whole rect when it is not cullable)
__device__ __forceinline__ uint32_t foot_count(const BinRec& b) {
  const int w = (int)(b.wh & 0xFFFFu), h = (int)(b.wh >> 16), y0 = (int)(b.xy >> 16);
  if (b.m < 0.f) return 0u;                    // never blends (alpha < alpha_skip)
  const Foot f = foot_setup(b);
  if (!f.cull) return (uint32_t)(w * h);
  uint32_t c = 0u;
  for (int ry = 0; ry < h; ++ry) {
    SlabPx s0, s1;
    int lo, hi;
    foot_row(f, y0 + ry, s0, s1, lo, hi);
    if (hi >= lo) c += (uint32_t)(hi - lo + 1);
  }
  return c;
}
// the 64-bit block bitmap of a rect of at most 4 x 4 tiles (bit 8 by + bx, blocks relative to the rect's first one)
__device__ __forceinline__ unsigned long long foot_bitmap(const BinRec& b) {
  const int x0 = (int)(b.xy & 0xFFFFu), y0 = (int)(b.xy >> 16), w = (int)(b.wh & 0xFFFFu), h = (int)(b.wh >> 16);
  const uint32_t rowfull = (1u << (2 * w)) - 1u;
  unsigned long long bits = 0ull;
  if (b.m < 0.f) return 0ull;                  // never blends (alpha < alpha_skip)
  const Foot f = foot_setup(b);
  for (int s = 0; s < 2 * h; ++s) {
    uint32_t row = rowfull;
    if (f.cull) {
      const SlabPx sp = foot_slab(f, y0 * EGS_TILE + 8 * s);
      row = 0u;
      if (sp.pl <= sp.pr) {
        const int bl = (sp.pl >> 3) - 2 * x0, br = (sp.pr >> 3) - 2 * x0;     // 0 <= bl <= br < 2 w  (pixel range clamped)
        row = ((2u << br) - 1u) & ~((1u << bl) - 1u);
      }
    }
    bits |= (unsigned long long)row << (8 * s);
  }
  return bits;
}
// the 64-bit TILE bitmap of a cullable rect of at most 8 x 8 tiles (bit 8 ty + tx)
__device__ __forceinline__ unsigned long long foot_tilemap(const BinRec& b) {
  const int x0 = (int)(b.xy & 0xFFFFu), y0 = (int)(b.xy >> 16), h = (int)(b.wh >> 16);
  if (b.m < 0.f) return 0ull;
  const Foot f = foot_setup(b);
  unsigned long long bits = 0ull;
  for (int ry = 0; ry < h; ++ry) {
    SlabPx s0, s1;
    int lo, hi;
    foot_row(f, y0 + ry, s0, s1, lo, hi);
    if (hi >= lo) bits |= (unsigned long long)(((2u << (hi - x0)) - 1u) & ~((1u << (lo - x0)) - 1u)) << (8 * ry);
  }
  return bits;
}
#pragma clang fp contract(fast)
// tiles of a <= 4 x 4 rect that have a block set: bit 16 ty + 2 tx
__device__ __forceinline__ unsigned long long cr_tile_bits(unsigned long long blocks) {
  unsigned long long t = blocks | (blocks >> 1);
  t |= t >> 8;
  return t & 0x0055005500550055ull;
}
__device__ __forceinline__ uint32_t cr_count(const uint4& c) {
  if (c.y & EGS_CR_BIG) return c.z;
  if (c.y & EGS_CR_ALLTILES) return (c.y & 0xFFFFu) * ((c.y & EGS_CR_WH_MASK) >> 16);   // the whole rect
  const unsigned long long b = ((unsigned long long)c.w << 32) | c.z;
  return (uint32_t)__popcll((c.y & EGS_CR_TILEMAP) ? b : cr_tile_bits(b));
}
// The footprint record of one Gaussian.  `cull`: the lists may drop tiles the footprint cannot reach (fused path);
// otherwise, and whenever the conic does not describe an ellipse the bounds above hold for (same rule as the
// certain-miss box of make_record), m = +inf and the Gaussian keeps every tile of its rect.
__device__ __forceinline__ BinRec make_binrec(float ux, float uy, float c0, float c1, float c2, float alpha,
                                              float alpha_skip, bool cull, const uint4& rect, uint32_t cnt_rect) {
  BinRec b;
  const float inf = __int_as_float(0x7f800000);
  b.ux = ux; b.uy = uy;
  b.A = -EGS_NHL2E * c0; b.Bh = -EGS_NHL2E * c1; b.C = -EGS_NHL2E * c2;   // -(qxx, qxy / 2, qyy) of the draw record
  b.m = inf;
  b.xy = cnt_rect ? (rect.x | (rect.y << 16)) : 0u;
  b.wh = cnt_rect ? ((rect.z - rect.x) | ((rect.w - rect.y) << 16)) : 0u;
  if (cull && alpha_skip > 0.f) {
    const float det = c0 * c2 - c1 * c1;
    if (det > 1e-4f * c0 * c2 && c0 > 0.f && c2 > 0.f && alpha == alpha) {
      if (alpha >= alpha_skip) {
        const float m = log2f(alpha / alpha_skip);      // = -thr of the draw record
        if (m == m && m < inf) b.m = m;
      } else {
        b.m = -1.f;                                      // alpha' <= alpha < skip everywhere: never blends (kernel.cu:246)
      }
    }
  }
  return b;
}

// per-workgroup (256 threads) maximum of the depth keys -> maxkey[1 + workgroup]; no atomics (a
// same-address atomicMax per wave measured +170 us); k_max_reduce folds the <= 4 K partial maxima
// `wm`: four words of LDS (the caller's: k_preprocess_fwd lends its staging buffer -- 16 bytes of its own would
// be the 128 bytes too many that keep an eighth workgroup off the CU)
__device__ __forceinline__ void block_max_key(uint32_t key, uint32_t* __restrict__ maxkey, uint32_t* wm) {
  uint32_t mk = key;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mk = max(mk, (uint32_t)__shfl_xor((int)mk, d, 64));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = mk;
  __syncthreads();
  if (threadIdx.x == 0) maxkey[1 + blockIdx.x] = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
}

__device__ __forceinline__ void make_record(float ux, float uy, float c0, float c1, float c2, float alpha,
                                            float r, float g, float b, int area_x, int area_y, int W, int H,
                                            int footprint, float alpha_skip, float4* __restrict__ out) {
  const float inf = __int_as_float(0x7f800000);
  float e1, e2;
  if (footprint == 1) {
    int x0, x1, y0, y1;
    pixel_box(ux, uy, (float)area_x, (float)area_y, W, H, x0, x1, y0, y1);
    e1 = __uint_as_float((uint32_t)x0 | ((uint32_t)x1 << 16));
    e2 = __uint_as_float((uint32_t)y0 | ((uint32_t)y1 << 16));
  } else {
    e1 = inf; e2 = inf;
    const float det = c0 * c2 - c1 * c1;
    // (det must not be the result of catastrophic cancellation: eigenvalue ratio < 1e4)
    if (alpha_skip > 0.f && det > 1e-4f * c0 * c2 && c0 > 0.f && c2 > 0.f) {
      if (alpha > alpha_skip) {
        const float mstar = 2.f * logf(alpha / alpha_skip);
        const float sxx = c2 / det, syy = c0 / det;  // Sigma = cinv^-1
        e1 = sqrtf(mstar * sxx) * 1.01f + 0.05f;
        e2 = sqrtf(mstar * syy) * 1.01f + 0.05f;
      } else if (alpha <= alpha_skip * 0.999f) {
        e1 = -inf; e2 = -inf;  // alpha' <= alpha < skip everywhere: never contributes
      }
    }
    if (!(e1 == e1) || !(e2 == e2)) { e1 = inf; e2 = inf; }  // NaN guard
  }
  // skip threshold in the exponent domain
  float thr;
  if (alpha_skip > 0.f) thr = (alpha >= alpha_skip) ? log2f(alpha_skip / alpha) : inf;  // alpha < skip never blends
  else thr = (alpha < 0.f) ? inf : -inf;  // !(alpha' < 0)
  out[0] = make_float4(ux, uy, EGS_NHL2E * c0, (2.f * EGS_NHL2E) * c1);
  out[1] = make_float4(EGS_NHL2E * c2, alpha, r, g);
  out[2] = make_float4(b, e1, e2, thr);
}

}  // namespace egs
