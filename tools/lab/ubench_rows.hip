// How should a lane-per-Gaussian kernel read its 192-byte SH row?  (k_preprocess_fwd sits at
// 0.46 of the HBM peak, its waves wait on row reads.)  Standalone: hipcc --offload-arch=gfx950 -O3 ubench_rows.hip
//   A  direct      each lane 12 x global_load_dwordx4 from its own row (192-B stride: 64 lines per instruction) -- today
//   B  lds_dma     each wave 12 x global_load_lds_dwordx4 of its contiguous 12-KB span (8 lines per instruction), no
//                  VGPRs in flight, then 12 x ds_read_b128 per lane (48-dword stride: 4-way bank conflict)
//   C  reg_stage   coalesced dwordx4 loads -> ds_write_b128 (odd 16-B stride) -> barrier -> ds_read_b128 (RowStage)
//   D  lds_dma2    as B with TWO 64-row batches per wave in flight (24 KB of LDS per wave)
// Every kernel also reads the 12-byte position of the row and writes 16 bytes per row (a stand-in for the outputs).
// The source rotates over four 192-MB buffers (768 MB: beyond the 256-MB Infinity Cache).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

__device__ __forceinline__ float sum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }

__global__ __launch_bounds__(256) void k_direct(const float4* __restrict__ src, const float* __restrict__ pw,
                                                float4* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4* r = src + (size_t)i * 12;
  float4 v[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) v[k] = r[k];
  const float p = pw[3 * (size_t)i] + pw[3 * (size_t)i + 1] + pw[3 * (size_t)i + 2];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < 12; k += 3) { s0 += sum4(v[k]); s1 += sum4(v[k + 1]); s2 += sum4(v[k + 2]); }
  out[i] = make_float4(s0, s1, s2, p);
}

template <int BATCH>
__global__ __launch_bounds__(256) void k_lds_dma(const float4* __restrict__ src, const float* __restrict__ pw,
                                                 float4* __restrict__ out, int n) {
  __shared__ float4 buf[4][BATCH][12 * 64];      // 12 KB per wave and batch
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + wave) * 64 * BATCH;
  if (row0 >= n) return;
#pragma unroll
  for (int b = 0; b < BATCH; ++b) {
    const float4* g = src + (size_t)(row0 + 64 * b) * 12;
#pragma unroll
    for (int j = 0; j < 12; ++j)
      __builtin_amdgcn_global_load_lds((glb_void*)(g + j * 64 + lane), (lds_void*)&buf[wave][b][j * 64], 16, 0, 0);
  }
  float p[BATCH];
#pragma unroll
  for (int b = 0; b < BATCH; ++b) {
    const size_t i = (size_t)row0 + 64 * b + lane;
    p[b] = pw[3 * i] + pw[3 * i + 1] + pw[3 * i + 2];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int b = 0; b < BATCH; ++b) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 12; k += 3) {
      s0 += sum4(buf[wave][b][lane * 12 + k]); s1 += sum4(buf[wave][b][lane * 12 + k + 1]);
      s2 += sum4(buf[wave][b][lane * 12 + k + 2]);
    }
    out[(size_t)row0 + 64 * b + lane] = make_float4(s0, s1, s2, p[b]);
  }
}

__global__ __launch_bounds__(256) void k_reg_stage(const float4* __restrict__ src, const float* __restrict__ pw,
                                                   float4* __restrict__ out, int n) {
  constexpr int STRIDE = 13;                      // float4 per row in LDS (odd: conflict-free ds_read_b128)
  __shared__ float4 buf[256 * STRIDE];
  const int tid = threadIdx.x, base = blockIdx.x * 256;
  const float4* s4 = src + (size_t)base * 12;
  float4 v[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) v[j] = s4[tid + 256 * j];
  const size_t i = (size_t)base + tid;
  const float p = pw[3 * i] + pw[3 * i + 1] + pw[3 * i + 2];
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const int f = tid + 256 * j, r = f / 12, c = f - 12 * r;
    buf[r * STRIDE + c] = v[j];
  }
  __syncthreads();
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < 12; k += 3) {
    s0 += sum4(buf[tid * STRIDE + k]); s1 += sum4(buf[tid * STRIDE + k + 1]); s2 += sum4(buf[tid * STRIDE + k + 2]);
  }
  out[i] = make_float4(s0, s1, s2, p);
}

int main(int argc, char** argv) {
  const int n = 1 << 20, NB = 4, iters = 40;
  std::vector<float4*> src(NB);
  float *pw;
  float4* out;
  for (int b = 0; b < NB; ++b) { CHECK(hipMalloc(&src[b], (size_t)n * 192)); CHECK(hipMemset(src[b], 0x3c, (size_t)n * 192)); }
  CHECK(hipMalloc(&pw, (size_t)n * 12)); CHECK(hipMemset(pw, 0, (size_t)n * 12));
  CHECK(hipMalloc(&out, (size_t)n * 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double bytes = (double)n * (192 + 12 + 16);
  for (int which = 0; which < 4; ++which) {
    const char* name[] = {"A direct", "B lds_dma (1 batch)", "C reg_stage", "D lds_dma (2 batches)"};
    float best = 1e9f, sum = 0.f;
    for (int it = 0; it < iters + 5; ++it) {
      const float4* s = src[it % NB];
      CHECK(hipEventRecord(e0));
      switch (which) {
        case 0: hipLaunchKernelGGL(k_direct, dim3(n / 256), dim3(256), 0, 0, s, pw, out, n); break;
        case 1: hipLaunchKernelGGL(k_lds_dma<1>, dim3(n / 256), dim3(256), 0, 0, s, pw, out, n); break;
        case 2: hipLaunchKernelGGL(k_reg_stage, dim3(n / 256), dim3(256), 0, 0, s, pw, out, n); break;
        default: hipLaunchKernelGGL(k_lds_dma<2>, dim3(n / 512), dim3(256), 0, 0, s, pw, out, n); break;
      }
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 5) { best = ms < best ? ms : best; sum += ms; }
    }
    CHECK(hipGetLastError());
    float4 h[2];
    CHECK(hipMemcpy(h, out + 12345, 32, hipMemcpyDeviceToHost));
    printf("%-24s avg %7.2f us  best %7.2f us  %6.2f TB/s (best)   check %.4f %.4f\n", name[which], 1e3 * sum / iters,
           1e3 * best, bytes / (best * 1e-3) / 1e12, h[0].x, h[1].z);
  }
  return 0;
}
