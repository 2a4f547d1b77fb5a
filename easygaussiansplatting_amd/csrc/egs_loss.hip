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
// 275 MB of HBM traffic per 1080p image (x,y in; 3 maps out; 3 maps + x,y in; grad out); the window
// passes are what costs: both are register-blocked (a thread slides the 11 taps over a run of 8 outputs of a
// row / 4 outputs of a column, reading every LDS value once per run instead of once per tap).
#include "egs_common.h"

namespace egs {

constexpr int LW = 11, LR = 5;            // window size / radius (pytorch_ssim.py:52: window_size=11)
constexpr int TW = 64, TH = 16;           // output tile
constexpr int IW = TW + 2 * LR, IH = TH + 2 * LR;  // input tile with halo: 74 x 26
constexpr int HRUN = 8, VRUN = 4;         // outputs per thread: horizontal pass (row run), vertical pass (column run)
constexpr int HSEG = TW / HRUN;           // 8 runs per tile row -> IH * HSEG = 208 of the 256 threads work
constexpr int NLOAD = (IH * IW + 255) / 256;  // halo-tile elements per thread
static_assert(IH * HSEG <= 256 && TW * (TH / VRUN) == 256, "pass shapes are tied to the 256-thread workgroup");

struct LossWin { float g[LW]; };          // normalised 1-D Gaussian, sigma = 1.5 (pytorch_ssim.py:11-13,17)

// image pixel (or 0: F.conv2d padding=5) of halo-tile element i
__device__ __forceinline__ bool tile_src(int i, int x0, int y0, int H, int W, int& r, int& c, size_t& off) {
  r = i / IW; c = i - r * IW;
  const int yy = y0 + r - LR, xx = x0 + c - LR;
  off = (size_t)yy * W + xx;
  return yy >= 0 && yy < H && xx >= 0 && xx < W;
}

// Persistent workgroups: a workgroup takes every step-th tile of its share (tile_walk); the halo tile of the NEXT tile is
// fetched into registers while the two window passes of the current one run, so the HBM latency is paid
// once per workgroup instead of once per tile (3 workgroups per CU fit: 49 KB of LDS each).
struct TileAt { int x0, y0; size_t plane; };
__device__ __forceinline__ TileAt tile_at(int t, int gx, int gy, int H, int W) {
  const int bx = t % gx, by = (t / gx) % gy, ch = t / (gx * gy);
  return {bx * TW, by * TH, (size_t)ch * H * W};
}

// Which tiles a persistent workgroup takes.  Workgroup b runs on XCD b % 8 (observed dispatch order; speed only), and
// every XCD has its own 4-MB L2: with tiles b, b + grid, ... the two tiles that share a halo column sit on different
// XCDs and the ones that share halo rows too (30 tiles per row, 30 % 8 != 0), so every halo came from HBM -- FETCH_SIZE
// read 155 MB for the forward kernel's 50 MB of pixels and 281 MB for the gradient kernel's 125 (rocprofv3 --pmc).
// Here every XCD walks a contiguous eighth of the tile list (row-major over channel, tile row, tile column), its
// resident workgroups side by side: the ~96 tiles in flight on an XCD span three tile rows (1.5 MB of halo tiles).
#ifndef EGS_LOSS_XCD_BANDS
#define EGS_LOSS_XCD_BANDS 1
#endif
// (Eight XCDs and the b % 8 dispatch are MI355X's -- this library is built for gfx950 only, csrc/Makefile.  The walk is
// CORRECT for any mapping of workgroups to dies: on a part with another die count the bands would merely stop matching
// the L2s, i.e. fall back to what the strided walk did.)
struct TileWalk { int t, end, step; };
__device__ __forceinline__ TileWalk tile_walk(int ntiles) {
  const int b = blockIdx.x, G = gridDim.x;
  if (!EGS_LOSS_XCD_BANDS || G < 16) return {b, ntiles, G};
  const int xcd = b & 7, slot = b >> 3;
  const int per = (G - xcd + 7) >> 3;                       // workgroups that run on this XCD
  const int lo = (int)((long long)ntiles * xcd / 8), hi = (int)((long long)ntiles * (xcd + 1) / 8);
  return {lo + slot, hi, per};
}

__global__ __launch_bounds__(256) void k_ssim_fwd(int H, int W, int gx, int gy, LossWin win,
                                                  const float* __restrict__ img, const float* __restrict__ gt,
                                                  float* __restrict__ Pm, float* __restrict__ P11,
                                                  float* __restrict__ P12,
                                                  float* __restrict__ partials /* [ntiles][2] */) {
  __shared__ float sx[IH][IW + 1], sy[IH][IW + 1];
  __shared__ float h[5][IH][TW + 1];      // horizontally filtered x, y, xx, yy, xy
  __shared__ float red[2][4];
  const int tid = threadIdx.x;
  const int ntiles = gx * gy * 3;
  float fa[NLOAD], fb[NLOAD];             // halo tile in flight: element tid + 256 j
  auto fetch = [&](int t) {               // every load is issued before anything waits on one
    const TileAt T = tile_at(t, gx, gy, H, W);
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
      int r, c; size_t off;
      fa[j] = 0.f; fb[j] = 0.f;
      const int i = tid + 256 * j;
      if (i < IH * IW && tile_src(i, T.x0, T.y0, H, W, r, c, off)) { fa[j] = img[T.plane + off]; fb[j] = gt[T.plane + off]; }
    }
  };
  const TileWalk tw = tile_walk(ntiles);
  int t = tw.t;
  if (t < tw.end) fetch(t);
  for (; t < tw.end; t += tw.step) {
  const TileAt T = tile_at(t, gx, gy, H, W);
  const int x0 = T.x0, y0 = T.y0;
  const size_t plane = T.plane;
#pragma unroll
  for (int j = 0; j < NLOAD; ++j) {
    const int i = tid + 256 * j;
    if (i < IH * IW) { const int r = i / IW, c = i - r * IW; sx[r][c] = fa[j]; sy[r][c] = fb[j]; }
  }
  __syncthreads();
  if (t + tw.step < tw.end) fetch(t + tw.step);
  if (tid < IH * HSEG) {  // horizontal 11-tap pass: HRUN outputs of row r from HRUN + 10 inputs
    const int r = tid % IH, c0 = (tid / IH) * HRUN;  // lanes of a wave walk down the rows: odd row stride, few bank conflicts
    float u[HRUN + LW - 1], v[HRUN + LW - 1];
#pragma unroll
    for (int k = 0; k < HRUN + LW - 1; ++k) { u[k] = sx[r][c0 + k]; v[k] = sy[r][c0 + k]; }
    float a[HRUN], b[HRUN], aa[HRUN], bb[HRUN], ab[HRUN];
#pragma unroll
    for (int o = 0; o < HRUN; ++o) { a[o] = 0.f; b[o] = 0.f; aa[o] = 0.f; bb[o] = 0.f; ab[o] = 0.f; }
#pragma unroll
    for (int k = 0; k < HRUN + LW - 1; ++k) {
      const float uu = u[k] * u[k], vv = v[k] * v[k], uv = u[k] * v[k];
#pragma unroll
      for (int o = 0; o < HRUN; ++o) {
        const int tp = k - o;  // tap index of input k for output o (ascending k == ascending tap: the reference's order)
        if (tp >= 0 && tp < LW) {
          const float w = win.g[tp];
          a[o] += w * u[k]; b[o] += w * v[k]; aa[o] += w * uu; bb[o] += w * vv; ab[o] += w * uv;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < HRUN; ++o) {
      h[0][r][c0 + o] = a[o]; h[1][r][c0 + o] = b[o]; h[2][r][c0 + o] = aa[o]; h[3][r][c0 + o] = bb[o];
      h[4][r][c0 + o] = ab[o];
    }
  }
  __syncthreads();
  float sum_l1 = 0.f, sum_s = 0.f;
  constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;  // pytorch_ssim.py:39-40
  {  // vertical pass + SSIM + partials: VRUN outputs of column c from VRUN + 10 rows
    const int c = tid % TW, r0 = (tid / TW) * VRUN;
    float acc[5][VRUN];
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
      for (int o = 0; o < VRUN; ++o) acc[m][o] = 0.f;
#pragma unroll
    for (int k = 0; k < VRUN + LW - 1; ++k) {
      float hv[5];
#pragma unroll
      for (int m = 0; m < 5; ++m) hv[m] = h[m][r0 + k][c];
#pragma unroll
      for (int o = 0; o < VRUN; ++o) {
        const int tp = k - o;
        if (tp >= 0 && tp < LW) {
#pragma unroll
          for (int m = 0; m < 5; ++m) acc[m][o] += win.g[tp] * hv[m];
        }
      }
    }
    const int xx = x0 + c;
#pragma unroll
    for (int o = 0; o < VRUN; ++o) {
      const int r = r0 + o, yy = y0 + r;
      if (yy >= H || xx >= W) continue;
      const float mu1 = acc[0][o], mu2 = acc[1][o], e11 = acc[2][o], e22 = acc[3][o], e12 = acc[4][o];
      const float s11 = e11 - mu1 * mu1, s22 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
      const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2;
      const float B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
      // two v_rcp_f32 (1 ulp) instead of three IEEE divisions (~10 instructions each): B1, B2 >= C1, C2 > 0
      const float rB1 = __builtin_amdgcn_rcpf(B1), rB2 = __builtin_amdgcn_rcpf(B2);
      const float inv = rB1 * rB2;
      const float S = A1 * A2 * inv;  // pytorch_ssim.py:42-43
      // dS/d(mu1, E11, E12) with sigma1_sq = E11 - mu1^2, sigma12 = E12 - mu1 mu2
      const float dA1 = A2 * inv, dA2 = A1 * inv, dB1 = -S * rB1, dB2 = -S * rB2;
      const size_t off = plane + (size_t)yy * W + xx;
      Pm[off] = 2.f * mu2 * (dA1 - dA2) + 2.f * mu1 * (dB1 - dB2);
      P11[off] = dB2;
      P12[off] = 2.f * dA2;
      sum_s += S;
      sum_l1 += fabsf(sx[r + LR][c + LR] - sy[r + LR][c + LR]);
    }
  }
  sum_l1 = wave_sum(sum_l1); sum_s = wave_sum(sum_s);
  if ((tid & 63) == 0) { red[0][tid >> 6] = sum_l1; red[1][tid >> 6] = sum_s; }
  __syncthreads();
  if (tid == 0) {  // one pair per TILE (not per workgroup): the summation order does not depend on the grid size
    partials[2 * (size_t)t] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partials[2 * (size_t)t + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
  __syncthreads();  // the next tile overwrites sx / sy / red
  }
}

// loss_out = {loss, l1, ssim}; fixed summation order -> bit-reproducible
__device__ __forceinline__ void loss_finalize(int nparts, const float* __restrict__ partials, float lambda,
                                              float inv_count, float* __restrict__ loss_out, double (*red)[4]) {
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
__global__ __launch_bounds__(256) void k_loss_finalize(int nparts, const float* __restrict__ partials, float lambda,
                                                       float inv_count, float* __restrict__ loss_out) {
  __shared__ double red[2][4];
  loss_finalize(nparts, partials, lambda, inv_count, loss_out, red);
}

__global__ __launch_bounds__(256) void k_ssim_bwd(int H, int W, int gx, int gy, LossWin win,
                                                  const float* __restrict__ img, const float* __restrict__ gt,
                                                  const float* __restrict__ Pm, const float* __restrict__ P11,
                                                  const float* __restrict__ P12, float c_l1, float c_ssim,
                                                  float* __restrict__ dimg, int nparts,
                                                  const float* __restrict__ partials, float lambda, float inv_count,
                                                  float* __restrict__ loss_out) {
  // loss_out != NULL: workgroup 0 also reduces the forward kernel's per-tile partials to {loss, l1, ssim} (what
  // k_loss_finalize does as a launch of its own: 8 us of a training step for a few thousand additions)
  __shared__ float s[3][IH][IW + 1];
  __shared__ float h[3][IH][TW + 1];
  __shared__ double fred[2][4];
  const int tid = threadIdx.x;
  const int ntiles = gx * gy * 3;
  float fa[NLOAD], fb[NLOAD], fd[NLOAD];  // halo tile of the three derivative maps in flight (see k_ssim_fwd)
  auto fetch = [&](int t) {
    const TileAt T = tile_at(t, gx, gy, H, W);
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
      int r, c; size_t off;
      fa[j] = 0.f; fb[j] = 0.f; fd[j] = 0.f;
      const int i = tid + 256 * j;
      if (i < IH * IW && tile_src(i, T.x0, T.y0, H, W, r, c, off)) {
        fa[j] = Pm[T.plane + off]; fb[j] = P11[T.plane + off]; fd[j] = P12[T.plane + off];
      }
    }
  };
  const TileWalk tw = tile_walk(ntiles);
  int t = tw.t;
  if (t < tw.end) fetch(t);
  if (loss_out && blockIdx.x == 0) loss_finalize(nparts, partials, lambda, inv_count, loss_out, fred);
  for (; t < tw.end; t += tw.step) {
  const TileAt T = tile_at(t, gx, gy, H, W);
  const int x0 = T.x0, y0 = T.y0;
  const size_t plane = T.plane;
  // this thread's VRUN output pixels: x, y are only needed at the very end -- fetched first, consumed last
  const int oc = tid % TW, or0 = (tid / TW) * VRUN;
  float px[VRUN], py[VRUN];
#pragma unroll
  for (int o = 0; o < VRUN; ++o) {
    const int yy = y0 + or0 + o, xx = x0 + oc;
    px[o] = 0.f; py[o] = 0.f;
    if (yy < H && xx < W) { px[o] = img[plane + (size_t)yy * W + xx]; py[o] = gt[plane + (size_t)yy * W + xx]; }
  }
#pragma unroll
  for (int j = 0; j < NLOAD; ++j) {
    const int i = tid + 256 * j;
    if (i < IH * IW) { const int r = i / IW, c = i - r * IW; s[0][r][c] = fa[j]; s[1][r][c] = fb[j]; s[2][r][c] = fd[j]; }
  }
  __syncthreads();
  if (t + tw.step < tw.end) fetch(t + tw.step);
  if (tid < IH * HSEG) {  // horizontal pass, HRUN outputs per thread (see k_ssim_fwd)
    const int r = tid % IH, c0 = (tid / IH) * HRUN;
    float acc[3][HRUN];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int o = 0; o < HRUN; ++o) acc[m][o] = 0.f;
#pragma unroll
    for (int k = 0; k < HRUN + LW - 1; ++k) {
      float sv[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) sv[m] = s[m][r][c0 + k];
#pragma unroll
      for (int o = 0; o < HRUN; ++o) {
        const int tp = k - o;
        if (tp >= 0 && tp < LW) {
#pragma unroll
          for (int m = 0; m < 3; ++m) acc[m][o] += win.g[tp] * sv[m];
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int o = 0; o < HRUN; ++o) h[m][r][c0 + o] = acc[m][o];
  }
  __syncthreads();
  {  // vertical pass, VRUN outputs per thread
    const int c = oc, r0 = or0;
    float acc[3][VRUN];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int o = 0; o < VRUN; ++o) acc[m][o] = 0.f;
#pragma unroll
    for (int k = 0; k < VRUN + LW - 1; ++k) {
      float hv[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) hv[m] = h[m][r0 + k][c];
#pragma unroll
      for (int o = 0; o < VRUN; ++o) {
        const int tp = k - o;
        if (tp >= 0 && tp < LW) {
#pragma unroll
          for (int m = 0; m < 3; ++m) acc[m][o] += win.g[tp] * hv[m];
        }
      }
    }
    const int xx = x0 + c;
#pragma unroll
    for (int o = 0; o < VRUN; ++o) {
      const int yy = y0 + r0 + o;
      if (yy >= H || xx >= W) continue;
      const size_t off = plane + (size_t)yy * W + xx;
      const float x = px[o], y = py[o];
      const float dS = acc[0][o] + 2.f * x * acc[1][o] + y * acc[2][o];   // d(sum of the SSIM map)/dx
      const float df = x - y;
      const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);  // torch.abs' subgradient: sign(0) = 0
      dimg[off] = c_l1 * sg + c_ssim * dS;
    }
  }
  __syncthreads();  // the next tile overwrites s / h
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
  const int gx = div_up(width, TW), gy = div_up(height, TH);
  static const int resident = [] {  // persistent workgroups: 3 per CU (LDS), every CU of the current device
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    return 3 * cus;
  }();
  const dim3 grid(nb < resident ? nb : resident);
  EGS_LAUNCH("k_ssim_fwd", k_ssim_fwd, grid, dim3(256), s, height, width, gx, gy, win, image, gt_image, Pm, P11, P12,
             partials);
  const float inv_count = (float)(1.0 / (double)npix);
  if (dloss_dimage) {   // (the gradient kernel's first workgroup reduces the partials on its way)
    const float c_l1 = grad_scale * (1.f - loss_lambda) * inv_count;
    const float c_ssim = -grad_scale * loss_lambda * inv_count;
    EGS_LAUNCH("k_ssim_bwd", k_ssim_bwd, grid, dim3(256), s, height, width, gx, gy, win, image, gt_image, Pm, P11, P12,
               c_l1, c_ssim, dloss_dimage, nb, partials, loss_lambda, inv_count, loss_out);
  } else {
    EGS_LAUNCH("k_loss_finalize", k_loss_finalize, dim3(1), dim3(256), s, nb, partials, loss_lambda, inv_count,
               loss_out);
  }
  EGS_LAUNCH_OK();
  return 0;
}
