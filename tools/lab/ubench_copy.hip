// Device-to-device copy variants: which shape reaches the achievable HBM bandwidth on this box?
//   hipcc --offload-arch=gfx950 -O3 tools/lab/ubench_copy.hip -o /tmp/ubench_copy && /tmp/ubench_copy
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f4v __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void k_copy(const f4v* __restrict__ src, f4v* __restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&src[i + u * stride]);
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

template <int U>
void run(int wg_per_cu, f4v* a, f4v* b, size_t n4) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k_copy<U>, dim3(256 * wg_per_cu), dim3(256), 0, 0, a, b, n4);
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_copy<U>, dim3(256 * wg_per_cu), dim3(256), 0, 0, a, b, n4);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("unroll %d, %2d WG/CU: %.1f GB/s (read + write)\n", U, wg_per_cu, 2.0 * n4 * 16 * 10 / (ms * 1e-3) / 1e9);
}

int main() {
  const size_t n4 = (size_t)1 << 26;   // 1 GiB
  f4v *a, *b; (void)hipMalloc(&a, n4 * 16); (void)hipMalloc(&b, n4 * 16); (void)hipMemset(a, 1, n4 * 16);
  for (int w : {4, 8, 16, 32}) { run<1>(w, a, b, n4); run<2>(w, a, b, n4); run<4>(w, a, b, n4); run<8>(w, a, b, n4); }
  float ms; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipMemcpyAsync(b, a, n4 * 16, hipMemcpyDeviceToDevice, 0);
  hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("hipMemcpyAsync D2D: %.1f GB/s\n", 2.0 * n4 * 16 * 10 / (ms * 1e-3) / 1e9);
  return 0;
}
