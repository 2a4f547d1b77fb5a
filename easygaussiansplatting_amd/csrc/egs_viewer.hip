// The viewer's per-frame Gaussian preprocess (reference viewer/shaders/gau_prep.glsl, an OpenGL compute
// shader dispatched by viewer/custom_items/gaussian_item.py:264-272) as a consumer of the same device
// functions as the rasterizer (last item).  One Gaussian per lane:
//
//   gs_data [N, 11 + K]  = {pos 3, rot 4 (w,x,y,z), scale 3, alpha 1, sh K}      (gau_prep.glsl:33-37)
//   gs_prep [N, 12]      = {u 3 (NDC), covinv 3, color 3, area 2, alpha 1}       (gau_prep.glsl:39-44)
//   depth   [N]          = view-space z (the key of the viewer's sort)            (gau_prep.glsl:188)
//
// Semantics are the shader's, not gsplatcu's: cull when |u.xy| > 1.3 or |u.z| > 1 in NDC or det == 0
// (only u = -100 is written then, the rest of the row is left as it was, like the shader does); no fov
// clamp in the covariance projection; area = 3 sqrt(diag); colour = SH + 0.5 without clamping.
// Matrices are 4x4 row-major in the mathematical convention (pc = V pw, u = P pc) -- what
// gaussian_item.py holds before set_uniform_mat4 transposes them for OpenGL.
#include "egs_gaussian_math.h"

namespace egs {

struct ViewerParams {
  float V[16], P[16];      // row-major
  float cam[3];            // inverse(V)[:3, 3]
  float fx, fy;
};

template <int NC>
__global__ __launch_bounds__(256) void k_viewer_prep(int n, ViewerParams vp, const float* __restrict__ gs_data,
                                                     float* __restrict__ gs_prep, float* __restrict__ depth) {
  constexpr int K = 3 * NC, DIM = 11 + K;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;                                    // (the shader tests `>`: one row past the end)
  const float* __restrict__ g = gs_data + (size_t)DIM * i;
  float* __restrict__ o = gs_prep + 12 * (size_t)i;
  const f3 pw = {g[0], g[1], g[2]};
  const float* V = vp.V;
  const float* P = vp.P;
  const float pcx = V[0] * pw.x + V[1] * pw.y + V[2] * pw.z + V[3];
  const float pcy = V[4] * pw.x + V[5] * pw.y + V[6] * pw.z + V[7];
  const float pcz = V[8] * pw.x + V[9] * pw.y + V[10] * pw.z + V[11];
  const float pcw = V[12] * pw.x + V[13] * pw.y + V[14] * pw.z + V[15];
  depth[i] = pcz;                                        // gau_prep.glsl:188
  float u[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) u[r] = P[4 * r] * pcx + P[4 * r + 1] * pcy + P[4 * r + 2] * pcz + P[4 * r + 3] * pcw;
  const float ux = u[0] / u[3], uy = u[1] / u[3], uz = u[2] / u[3];
  if (fabsf(ux) > 1.3f || fabsf(uy) > 1.3f || fabsf(uz) > 1.f) {   // gau_prep.glsl:192-203
    o[0] = -100.f; o[1] = -100.f; o[2] = -100.f;
    return;
  }
  const float4 q = make_float4(g[3], g[4], g[5], g[6]);  // (w, x, y, z), used as is
  const Cov3 c3 = cov3d_f(q, f3{g[7], g[8], g[9]});
  // computeCov2D (gau_prep.glsl:93-112): T = J W, cov = T Sigma T^T, + 0.3 on the diagonal; no fov clamp
  const float z2 = pcz * pcz;
  const float j00 = vp.fx / pcz, j02 = -(vp.fx * pcx) / z2, j11 = vp.fy / pcz, j12 = -(vp.fy * pcy) / z2;
  float T0[3], T1[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    T0[c] = j00 * V[c] + j02 * V[8 + c];
    T1[c] = j11 * V[4 + c] + j12 * V[8 + c];
  }
  // Sigma from its 6 unique entries (xx, xy, xz, yy, yz, zz)
  const float* s = c3.c;
  const float S0[3] = {s[0] * T0[0] + s[1] * T0[1] + s[2] * T0[2], s[1] * T0[0] + s[3] * T0[1] + s[4] * T0[2],
                       s[2] * T0[0] + s[4] * T0[1] + s[5] * T0[2]};
  const float S1[3] = {s[0] * T1[0] + s[1] * T1[1] + s[2] * T1[2], s[1] * T1[0] + s[3] * T1[1] + s[4] * T1[2],
                       s[2] * T1[0] + s[4] * T1[1] + s[5] * T1[2]};
  const float c00 = T0[0] * S0[0] + T0[1] * S0[1] + T0[2] * S0[2] + 0.3f;
  const float c01 = T0[0] * S1[0] + T0[1] * S1[1] + T0[2] * S1[2];
  const float c11 = T1[0] * S1[0] + T1[1] * S1[1] + T1[2] * S1[2] + 0.3f;
  const float det = c00 * c11 - c01 * c01;
  if (det == 0.f) {                                      // gau_prep.glsl:219-223
    o[0] = -100.f; o[1] = -100.f; o[2] = -100.f;
    return;
  }
  const float det_inv = 1.f / det;
  // colour: SH of the normalised ray from the camera centre, + 0.5, not clamped (gau_prep.glsl:128-176, 231-237)
  float sh[K];
#pragma unroll
  for (int k = 0; k < K; ++k) sh[k] = g[11 + k];
  const float twc[3] = {vp.cam[0], vp.cam[1], vp.cam[2]};
  const ShDir<NC> d = sh_basis_f<NC>(pw, twc);
  float col[3];
  sh_color_f<NC>(d, sh, col);   // SH + 0.5, no clamp: the shader's computeColor
  o[0] = ux; o[1] = uy; o[2] = uz;
  o[3] = c11 * det_inv; o[4] = -c01 * det_inv; o[5] = c00 * det_inv;
  o[6] = col[0]; o[7] = col[1]; o[8] = col[2];
  o[9] = 3.f * sqrtf(c00); o[10] = 3.f * sqrtf(c11);     // drawing area: 3 sigma of x and y
  o[11] = g[10];
}

// 4x4 inverse (row-major) by Gauss-Jordan with partial pivoting; false if singular
static bool invert4(const float* m, double* inv) {
  double a[4][8];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) { a[r][c] = m[4 * r + c]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    for (int r = col + 1; r < 4; ++r)
      if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
    if (a[piv][col] == 0.0) return false;
    for (int c = 0; c < 8; ++c) { const double t = a[col][c]; a[col][c] = a[piv][c]; a[piv][c] = t; }
    const double d = a[col][col];
    for (int c = 0; c < 8; ++c) a[col][c] /= d;
    for (int r = 0; r < 4; ++r) {
      if (r == col) continue;
      const double f = a[r][col];
      for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
    }
  }
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) inv[4 * r + c] = a[r][4 + c];
  return true;
}

}  // namespace egs

using namespace egs;

extern "C" int egs_viewer_prep(int n, int sh_dim, const float* gs_data, const float* view_matrix,
                               const float* projection_matrix, float focal_x, float focal_y, float* gs_prep,
                               float* depth, void* stream) {
  EGS_CHECK_ARG(n >= 0 && view_matrix && projection_matrix);
  EGS_CHECK_ARG(sh_dim == 3 || sh_dim == 12 || sh_dim == 27 || sh_dim == 48);
  if (n == 0) return 0;
  EGS_CHECK_ARG(gs_data && gs_prep && depth);
  ViewerParams vp;
  for (int k = 0; k < 16; ++k) { vp.V[k] = view_matrix[k]; vp.P[k] = projection_matrix[k]; }
  double inv[16];
  EGS_CHECK_ARG(invert4(view_matrix, inv));               // cam_pos = inverse(view_matrix)[3].xyz
  vp.cam[0] = (float)inv[3]; vp.cam[1] = (float)inv[7]; vp.cam[2] = (float)inv[11];
  vp.fx = focal_x; vp.fy = focal_y;
  hipStream_t s = (hipStream_t)stream;
  dim3 g(div_up(n, 256)), b(256);
  switch (sh_dim) {
    case 3: EGS_LAUNCH("k_viewer_prep", k_viewer_prep<1>, g, b, s, n, vp, gs_data, gs_prep, depth); break;
    case 12: EGS_LAUNCH("k_viewer_prep", k_viewer_prep<4>, g, b, s, n, vp, gs_data, gs_prep, depth); break;
    case 27: EGS_LAUNCH("k_viewer_prep", k_viewer_prep<9>, g, b, s, n, vp, gs_data, gs_prep, depth); break;
    default: EGS_LAUNCH("k_viewer_prep", k_viewer_prep<16>, g, b, s, n, vp, gs_data, gs_prep, depth); break;
  }
  EGS_LAUNCH_OK();
  return 0;
}
