// Calibration micro-benchmarks for the SQ counters and for the instruction mix of the draw kernels (gfx950).
// Each kernel issues ONE kind of VALU instruction in 8 independent chains at 8 waves per SIMD, so the VALU pipe
// is the only limit: run plain it prints nominal cycles per wave-instruction per SIMD; run under
//   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
// the ratio SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) of the pure-FMA kernel is what the
// counter reads at 100 % issue utilisation -- the factor every "VALU busy" figure under profiles/ is divided by.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/ubench_calib.hip -o gpurun_out/ubench_calib && gpurun_out/ubench_calib
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHAINS 8
#define ITERS 4096

namespace egs {
enum { FMA, EXP, RCP, SWAP32, SWAP16, DPPADD, CNDMASK, READLANE, MOV64, MED3, CMPS, SWZ, BPERM, DPPMASK, VMIN, MOV32, CMPX,
       SWZADD, FMAC, PKFMA, PKMUL, PKADD, VADD, VMUL, DSR128B, DSR128, DSR64B, DSR32B, NOPS };

template <int OP>
__global__ __launch_bounds__(256) void k_ub(float* out, float seed) {
  float a[CHAINS], b[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; b[i] = seed * 0.5f + i; }
  float c = seed * 1.0001f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pa[CHAINS / 2], pb[CHAINS / 2], pc = {c, c * 0.999f};
#pragma unroll
  for (int i = 0; i < CHAINS / 2; ++i) { pa[i] = f2{a[2 * i], a[2 * i + 1]}; pb[i] = f2{b[2 * i], b[2 * i + 1]}; }
  const int addr = ((threadIdx.x & 63) ^ 32) * 4;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 q4[2] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
  __shared__ float lds_buf[4096];
  if (OP >= DSR128B && OP <= DSR32B) {
    for (int i = threadIdx.x; i < 4096; i += 256) lds_buf[i] = seed + i;
    __syncthreads();
  }
  const int zero = (int)(seed > 1e30f), lane16 = (threadIdx.x & 63) * 16;
  for (int it = 0; it < ITERS; ++it) {
    if (OP >= DSR128B && OP <= DSR32B) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
      if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (OP == SWAP32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
      if (OP == SWAP16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
      if (OP == DPPADD) asm volatile("v_add_f32_dpp %0, %1, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
      if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b[i]));
      if (OP == READLANE) asm volatile("v_readlane_b32 s22, %0, 3" :: "v"(a[i]) : "s22");
      if (OP == MOV64) asm volatile("v_mov_b64 %0, 0" : "=v"(*(double*)&a[i & ~1]));
      if (OP == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
      if (OP == CMPS) asm volatile("v_cmp_lt_f32 s[24:25], %0, %1" :: "v"(a[i]), "v"(c) : "s24", "s25");
      // LDS crossbar exchanges (no VALU slot): lane ^ 16 by swizzle (immediate pattern), lane ^ 32 by bpermute
      if (OP == SWZ) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM, \"0000p\")" : "+v"(a[i]));
      if (OP == BPERM) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[i]) : "v"(addr));
      if (OP == DPPMASK) asm volatile("v_add_f32_dpp %0, %1, %2 quad_perm:[0,1,2,3] row_mask:0xc bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]), "v"(c));
      if (OP == VMIN) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == MOV32) asm volatile("v_mov_b32 %0, 0" : "=v"(a[i]));
      if (OP == CMPX) asm volatile("v_cmpx_lt_f32 vcc, %0, %1\n s_mov_b64 exec, -1" :: "v"(a[i]), "v"(c) : "vcc");
      // the exchange + add pair of the reduction: swizzle, then a full-rate add of the returned value
      if (OP == SWZADD) { float t; asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(BITMASK_PERM, \"0000p\")" : "=v"(t) : "v"(b[i]));
                          asm volatile("s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(t)); }
      if (OP == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
      // packed f32: two lanes' worth of work per instruction on an aligned register pair (4 pairs = 8 values)
      if (OP == PKFMA && i < CHAINS / 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa[i]) : "v"(pc), "v"(pb[i]));
      if (OP == PKMUL && i < CHAINS / 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pc));
      if (OP == PKADD && i < CHAINS / 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pb[i]));
      // LDS reads: every lane the SAME address (broadcast: how the draw kernels fetch an entry's record) vs one
      // 16-B piece per lane; the LDS pipe is shared by the four SIMDs of a CU
      if (OP == DSR128B) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q4[i & 1]) : "v"(zero), "n"(16 * i));
      if (OP == DSR128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q4[i & 1]) : "v"(lane16), "n"(1024 * i));
      if (OP == DSR64B) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(pa[i & 3]) : "v"(zero), "n"(16 * i));
      if (OP == DSR32B) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[i]) : "v"(zero), "n"(16 * i));
      if (OP == VADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      if (OP == VMUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += a[i] + b[i];
#pragma unroll
  for (int i = 0; i < CHAINS / 2; ++i) s += pa[i].x + pa[i].y;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  s += q4[0].x + q4[1].y + lds_buf[threadIdx.x];
  if (s == 12345.678f) out[0] = s;
}
}  // namespace egs

template <int OP>
static void run(const char* name, float* d, int per_iter = CHAINS) {
  const int blocks = 2048;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(egs::k_ub<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(egs::k_ub<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winst = (double)blocks * 4 * ITERS * per_iter / 1024.0;   // wave-instructions per SIMD
  printf("%-10s %8.3f ms  %6.3f ns per wave-instr per SIMD (= %.2f cycles @2.4 GHz nominal)\n", name, ms,
         ms * 1e6 / winst, ms * 1e6 / winst * 2.4);
}

int main() {
  float* d; hipMalloc(&d, 1024);
  run<egs::FMA>("v_fma", d); run<egs::EXP>("v_exp", d); run<egs::RCP>("v_rcp", d);
  run<egs::SWAP32>("swap32", d); run<egs::SWAP16>("swap16", d); run<egs::DPPADD>("add_dpp", d);
  run<egs::CNDMASK>("cndmask", d); run<egs::READLANE>("readlane", d); run<egs::MOV64>("mov_b64", d);
  run<egs::MED3>("med3", d); run<egs::CMPS>("cmp->sgpr", d);
  run<egs::SWZ>("ds_swizzle", d); run<egs::BPERM>("ds_bpermute", d); run<egs::DPPMASK>("add_dpp_rowmask", d);
  run<egs::VMIN>("v_min", d); run<egs::MOV32>("mov_b32", d); run<egs::CMPX>("cmpx+exec", d);
  run<egs::SWZADD>("swizzle+wait+add", d); run<egs::FMAC>("v_fmac", d);
  // packed f32 (CHAINS/2 instructions per iteration, each doing two values per lane)
  run<egs::PKFMA>("v_pk_fma", d, CHAINS / 2); run<egs::PKMUL>("v_pk_mul", d, CHAINS / 2);
  run<egs::PKADD>("v_pk_add", d, CHAINS / 2);
  run<egs::VADD>("v_add", d); run<egs::VMUL>("v_mul", d);
  run<egs::DSR128B>("ds_read_b128 bcast", d); run<egs::DSR128>("ds_read_b128 lanes", d);
  run<egs::DSR64B>("ds_read_b64 bcast", d); run<egs::DSR32B>("ds_read_b32 bcast", d);
  return 0;
}
