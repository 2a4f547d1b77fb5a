// Micro-benchmark: issue cost of the VALU / LDS ops the draw kernels are made of,
// at 8 waves/SIMD (2048 blocks x 256 threads on 256 CUs).  Prints cycles per
// wave-instruction per SIMD assuming the reported clock.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/ubench_valu.hip -o gpurun_out/ubench && gpurun_out/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHAINS 8
#define ITERS 2048

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  __shared__ float4 sm[64];
  if (threadIdx.x < 64) sm[threadIdx.x] = make_float4(seed, seed, seed, seed);
  __syncthreads();
  float a[CHAINS], b[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; b[i] = seed * 0.5f + i; }
  float c = seed * 1.0001f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[i & ~1]) : "v"(*(double*)&b[i & ~1]));
      if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 4) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
      if (OP == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a[i & ~1]) : "v"(*(double*)&b[i & ~1]));
      if (OP == 8) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" :: "v"(a[i]), "v"(c) : "s20", "s21");
      if (OP == 9) { float4 v; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((it & 63) * 16)); a[i] += v.x; }
      if (OP == 10) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 11) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += a[i];
  if (s == 12345.678f) out[0] = s;
}

template <int OP>
void run(const char* name, float* d, int per_iter_insts, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD = blocks*4 waves * ITERS*CHAINS*per_iter / 1024 SIMDs
  double winst = (double)blocks * 4 * ITERS * CHAINS * per_iter_insts / 1024.0;
  printf("%-28s blocks=%5d  %8.3f ms   %6.2f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz)\n", name, blocks, ms,
         ms * 1e6 / winst, ms * 1e6 / winst * 2.4);
}

int main() {
  float* d; hipMalloc(&d, 1024);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  for (int blocks : {2048, 256}) {
    run<0>("v_fma_f32", d, 1, blocks);
    run<11>("v_fmac_f32", d, 1, blocks);
    run<10>("v_add_f32", d, 1, blocks);
    run<6>("v_mul_f32", d, 1, blocks);
    run<1>("v_pk_fma_f32", d, 1, blocks);
    run<7>("v_pk_mul_f32", d, 1, blocks);
    run<3>("v_max_f32", d, 1, blocks);
    run<2>("v_exp_f32", d, 1, blocks);
    run<5>("v_rcp_f32", d, 1, blocks);
    run<4>("v_cmp+v_cndmask (2 inst)", d, 2, blocks);
    run<8>("v_cmp_lt_f32 -> sgpr", d, 1, blocks);
    run<9>("ds_read_b128 bcast + wait", d, 1, blocks);
  }
  return 0;
}
