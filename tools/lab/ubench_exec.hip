// Micro-benchmark (round 4): what does a VALU instruction cost when only PART of the wave is enabled?
// One kernel per EXEC pattern, 8 waves/SIMD, independent chains of v_fma_f32 / v_exp_f32 / v_rcp_f32 under
// s_mov_b64 exec, <mask>.  If the SIMD16 skipped 16-lane passes whose lanes are all disabled, "one quarter" would cost
// a quarter of "all lanes"; if cost follows the NUMBER of enabled quarters, "one lane in every quarter" costs as much
// as "all lanes".
//   hipcc --offload-arch=gfx950 -O3 tools/lab/ubench_exec.hip -o /tmp/ubench_exec && /tmp/ubench_exec
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHAINS 8
#define ITERS 4096

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float seed, unsigned long long mask) {
  float a[CHAINS], b[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; b[i] = seed * 0.5f + i; }
  float c = seed * 1.0001f;
  unsigned long long saved;
  asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=s"(saved) : "s"(mask));
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
      if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
    }
  }
  asm volatile("s_mov_b64 exec, %0" :: "s"(saved));
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += a[i];
  if (s == 12345.678f) out[0] = s;
}

template <int OP>
void run(const char* name, float* d, unsigned long long mask, const char* mname) {
  const int blocks = 2048;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5f, mask);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5f, mask);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double winst = (double)blocks * 4 * ITERS * CHAINS / 1024.0;
  printf("%-12s %-34s %8.3f ms  %6.2f cycles per wave-instr per SIMD @2.4GHz\n", name, mname, ms, ms * 1e6 / winst * 2.4);
}

int main() {
  float* d; hipMalloc(&d, 1024);
  struct { unsigned long long m; const char* n; } masks[] = {
      {0xFFFFFFFFFFFFFFFFull, "all 64 lanes"},
      {0x00000000FFFFFFFFull, "lanes 0-31 (two quarters)"},
      {0x000000000000FFFFull, "lanes 0-15 (one quarter)"},
      {0x0000FFFF0000FFFFull, "quarters 0 and 2"},
      {0x0001000100010001ull, "one lane in every quarter"},
      {0x0000000000000001ull, "lane 0 only"},
      {0x00FF00FF00FF00FFull, "half of every quarter"},
      {0x000000000F0F0F0Full, "4x4-in-8x8: half rows of 2 quarters"},
  };
  for (int rep = 0; rep < 2; ++rep)
    for (auto& mk : masks) {
      run<0>("v_fma_f32", d, mk.m, mk.n);
      run<1>("v_exp_f32", d, mk.m, mk.n);
      run<2>("v_rcp_f32", d, mk.m, mk.n);
      run<3>("v_med3_f32", d, mk.m, mk.n);
    }
  return 0;
}
