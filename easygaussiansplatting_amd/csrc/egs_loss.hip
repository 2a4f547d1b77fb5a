// Fused training loss for gfx950: gau_loss = (1-lambda) * L1 + lambda * (1 - SSIM)
// (reference gsplat/pytorch_ssim.py:10-66: five depthwise 11x11 Gaussian-window conv2d
// calls forward plus their autograd backward in PyTorch -- measured 10.9 ms per 1920x1080
// image on MI355X).  Here: two tiled kernels, separable 11-tap window staged in LDS,
// producing the loss AND dL/dimage (the input of splatB) directly.
//
//   k_ssim_fwd : per 64x16 tile and channel: mu1, mu2, E[x^2], E[y^2], E[xy] -> SSIM map value,
//                per-workgroup partial sums of |x-y| and SSIM, and the three partial-derivative
//                maps Pm = dS/dmu1, P11 = dS/dE11, P12 = dS/dE12.
//   k_ssim_bwd : dS_total/dx = W*Pm + 2x (W*P11) + y (W*P12)   (W symmetric, zero padding),
//                dL/dx = (1-lambda)/M sign(x-y) - lambda/M dS_total/dx.
//   k_loss_finalize : deterministic reduction of the per-workgroup partials.
//
// HBM-bound: 275 MB per 1080p image (x,y in; 3 maps out; 3 maps + x,y in; grad out).
#include "egs_common.h"

namespace egs {

constexpr int LW = 11, LR = 5;            // window size / radius (pytorch_ssim.py:52: window_size=11)
constexpr int TW = 64, TH = 16;           // output tile
constexpr int IW = TW + 2 * LR, IH = TH + 2 * LR;  // input tile with halo: 74 x 26

struct LossWin { float g[LW]; };          // normalised 1-D Gaussian, sigma = 1.5 (pytorch_ssim.py:11-13,17)

__global__ __launch_bounds__(256) void k_ssim_fwd(int H, int W, LossWin win, const float* __restrict__ img,
                                                  const float* __restrict__ gt, float* __restrict__ Pm,
                                                  float* __restrict__ P11, float* __restrict__ P12,
                                                  float* __restrict__ partials /* [nblocks][2] */) {
  __shared__ float sx[IH][IW + 1], sy[IH][IW + 1];
  __shared__ float h[5][IH][TW + 1];      // horizontally filtered x, y, xx, yy, xy
  __shared__ float red[2][4];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, ch = blockIdx.z;
  const size_t plane = (size_t)ch * H * W;
  for (int i = tid; i < IH * IW; i += 256) {  // zero padding outside the image (F.conv2d padding=5)
    const int r = i / IW, c = i - r * IW;
    const int yy = y0 + r - LR, xx = x0 + c - LR;
    float a = 0.f, b = 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      a = img[plane + (size_t)yy * W + xx];
      b = gt[plane + (size_t)yy * W + xx];
    }
    sx[r][c] = a; sy[r][c] = b;
  }
  __syncthreads();
  for (int i = tid; i < IH * TW; i += 256) {  // horizontal 11-tap pass
    const int r = i / TW, c = i - r * TW;
    float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
    for (int k = 0; k < LW; ++k) {
      const float w = win.g[k], u = sx[r][c + k], v = sy[r][c + k];
      a += w * u; b += w * v; aa += w * (u * u); bb += w * (v * v); ab += w * (u * v);
    }
    h[0][r][c] = a; h[1][r][c] = b; h[2][r][c] = aa; h[3][r][c] = bb; h[4][r][c] = ab;
  }
  __syncthreads();
  float sum_l1 = 0.f, sum_s = 0.f;
  constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;  // pytorch_ssim.py:39-40
  for (int i = tid; i < TH * TW; i += 256) {  // vertical pass + SSIM + partials
    const int r = i / TW, c = i - r * TW;
    const int yy = y0 + r, xx = x0 + c;
    if (yy >= H || xx >= W) continue;
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < LW; ++k) {
      const float w = win.g[k];
      mu1 += w * h[0][r + k][c]; mu2 += w * h[1][r + k][c];
      e11 += w * h[2][r + k][c]; e22 += w * h[3][r + k][c]; e12 += w * h[4][r + k][c];
    }
    const float s11 = e11 - mu1 * mu1, s22 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
    const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2;
    const float B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
    const float inv = 1.f / (B1 * B2);
    const float S = A1 * A2 * inv;  // pytorch_ssim.py:42-43
    // dS/d(mu1, E11, E12) with sigma1_sq = E11 - mu1^2, sigma12 = E12 - mu1 mu2
    const float dA1 = A2 * inv, dA2 = A1 * inv, dB1 = -S / B1, dB2 = -S / B2;
    const size_t o = plane + (size_t)yy * W + xx;
    Pm[o] = 2.f * mu2 * (dA1 - dA2) + 2.f * mu1 * (dB1 - dB2);
    P11[o] = dB2;
    P12[o] = 2.f * dA2;
    sum_s += S;
    sum_l1 += fabsf(sx[r + LR][c + LR] - sy[r + LR][c + LR]);
  }
  sum_l1 = wave_sum(sum_l1); sum_s = wave_sum(sum_s);
  if ((tid & 63) == 0) { red[0][tid >> 6] = sum_l1; red[1][tid >> 6] = sum_s; }
  __syncthreads();
  if (tid == 0) {
    const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partials[2 * b] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partials[2 * b + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// loss_out = {loss, l1, ssim}; fixed summation order -> bit-reproducible
__global__ __launch_bounds__(256) void k_loss_finalize(int nparts, const float* __restrict__ partials, float lambda,
                                                       float inv_count, float* __restrict__ loss_out) {
  __shared__ double red[2][4];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) { a += partials[2 * (size_t)i]; b += partials[2 * (size_t)i + 1]; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double l1 = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) * inv_count;
    const double ss = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) * inv_count;
    loss_out[0] = (float)((1.0 - lambda) * l1 + lambda * (1.0 - ss));  // pytorch_ssim.py:63-66
    loss_out[1] = (float)l1;
    loss_out[2] = (float)ss;
  }
}

__global__ __launch_bounds__(256) void k_ssim_bwd(int H, int W, LossWin win, const float* __restrict__ img,
                                                  const float* __restrict__ gt, const float* __restrict__ Pm,
                                                  const float* __restrict__ P11, const float* __restrict__ P12,
                                                  float c_l1, float c_ssim, float* __restrict__ dimg) {
  __shared__ float s[3][IH][IW + 1];
  __shared__ float h[3][IH][TW + 1];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, ch = blockIdx.z;
  const size_t plane = (size_t)ch * H * W;
  for (int i = tid; i < IH * IW; i += 256) {
    const int r = i / IW, c = i - r * IW;
    const int yy = y0 + r - LR, xx = x0 + c - LR;
    float a = 0.f, b = 0.f, d = 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      const size_t o = plane + (size_t)yy * W + xx;
      a = Pm[o]; b = P11[o]; d = P12[o];
    }
    s[0][r][c] = a; s[1][r][c] = b; s[2][r][c] = d;
  }
  __syncthreads();
  for (int i = tid; i < IH * TW; i += 256) {
    const int r = i / TW, c = i - r * TW;
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < LW; ++k) {
      const float w = win.g[k];
      a += w * s[0][r][c + k]; b += w * s[1][r][c + k]; d += w * s[2][r][c + k];
    }
    h[0][r][c] = a; h[1][r][c] = b; h[2][r][c] = d;
  }
  __syncthreads();
  for (int i = tid; i < TH * TW; i += 256) {
    const int r = i / TW, c = i - r * TW;
    const int yy = y0 + r, xx = x0 + c;
    if (yy >= H || xx >= W) continue;
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < LW; ++k) {
      const float w = win.g[k];
      a += w * h[0][r + k][c]; b += w * h[1][r + k][c]; d += w * h[2][r + k][c];
    }
    const size_t o = plane + (size_t)yy * W + xx;
    const float x = img[o], y = gt[o];
    const float dS = a + 2.f * x * b + y * d;   // d(sum of the SSIM map)/dx
    const float df = x - y;
    const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);  // torch.abs' subgradient: sign(0) = 0
    dimg[o] = c_l1 * sg + c_ssim * dS;
  }
}

}  // namespace egs

using namespace egs;

static int loss_blocks(int H, int W) { return div_up(W, TW) * div_up(H, TH) * 3; }

extern "C" size_t egs_gau_loss_ws_bytes(int height, int width) {
  if (height <= 0 || width <= 0) return 256;
  return 3 * align_up((size_t)3 * height * width * 4, 256) + align_up((size_t)loss_blocks(height, width) * 8, 256) + 256;
}

extern "C" int egs_gau_loss(int height, int width, const float* image, const float* gt_image, float loss_lambda,
                            float grad_scale, void* ws, size_t ws_bytes, float* loss_out, float* dloss_dimage,
                            void* stream) {
  EGS_CHECK_ARG(height > 0 && width > 0 && image && gt_image && ws && loss_out);
  if (ws_bytes < egs_gau_loss_ws_bytes(height, width)) {
    set_error(EGS_ERR_WORKSPACE, "gau_loss workspace too small", __FILE__, __LINE__);
    return EGS_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  Carver cv(ws, ws_bytes);
  const size_t npix = (size_t)3 * height * width;
  float* Pm = cv.take<float>(npix);
  float* P11 = cv.take<float>(npix);
  float* P12 = cv.take<float>(npix);
  const int nb = loss_blocks(height, width);
  float* partials = cv.take<float>((size_t)nb * 2);
  LossWin win;
  double sum = 0.0, g[LW];
  for (int i = 0; i < LW; ++i) { g[i] = exp(-(double)((i - LR) * (i - LR)) / (2.0 * 1.5 * 1.5)); sum += g[i]; }
  for (int i = 0; i < LW; ++i) win.g[i] = (float)(g[i] / sum);
  dim3 grid(div_up(width, TW), div_up(height, TH), 3);
  EGS_LAUNCH("k_ssim_fwd", k_ssim_fwd, grid, dim3(256), s, height, width, win, image, gt_image, Pm, P11, P12,
             partials);
  const float inv_count = (float)(1.0 / (double)npix);
  EGS_LAUNCH("k_loss_finalize", k_loss_finalize, dim3(1), dim3(256), s, nb, partials, loss_lambda, inv_count,
             loss_out);
  if (dloss_dimage) {
    const float c_l1 = grad_scale * (1.f - loss_lambda) * inv_count;
    const float c_ssim = -grad_scale * loss_lambda * inv_count;
    EGS_LAUNCH("k_ssim_bwd", k_ssim_bwd, grid, dim3(256), s, height, width, win, image, gt_image, Pm, P11, P12, c_l1,
               c_ssim, dloss_dimage);
  }
  EGS_LAUNCH_OK();
  return 0;
}
