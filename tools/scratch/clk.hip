#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
__global__ void clk(unsigned long long* out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  float x = threadIdx.x;
  for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  unsigned long long h[3];
  for (int trial = 0; trial < 6; ++trial) {
    int iters = (trial % 2 == 0) ? 2000 : 2000000;
    if (trial >= 4) usleep(200000);
    hipLaunchKernelGGL(clk, dim3(trial >= 2 ? 1024 : 1), dim3(256), 0, 0, d, iters);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("trial %d iters %d blocks %d: cycles %llu real(100MHz) %llu -> %.1f MHz\n", trial, iters, trial >= 2 ? 1024 : 1, h[0], h[1], h[1] ? 100.0 * h[0] / h[1] : 0.0);
  }
  // many tiny kernels back to back
  for (int k = 0; k < 2000; ++k) hipLaunchKernelGGL(clk, dim3(64), dim3(256), 0, 0, d, 200);
  hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("after 2000 tiny kernels: cycles %llu real %llu -> %.1f MHz\n", h[0], h[1], h[1] ? 100.0 * h[0] / h[1] : 0.0);
  return 0;
}
