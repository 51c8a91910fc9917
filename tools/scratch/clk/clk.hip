// scratch: ratio of clock64() (s_memtime) to wall_clock64() (100 MHz) and cost of dependent VALU chains
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, float x, int n) {
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  float a = x;
  for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;   // dependent chain: 2n VALU ops (contract off -> mul, add)
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  double d = x;
  for (int i = 0; i < n; ++i) d = d * 1.0001 + 0.5;
  unsigned long long c2 = clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = c2 - c1; out[3] = (unsigned long long)(a + d); }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  for (int rep = 0; rep < 3; ++rep) k<<<1, 64>>>(d, 1.f, 100000);
  unsigned long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("clock64 ticks %llu, wall(100MHz) ticks %llu -> clock64 = %.1f MHz; f32 chain %.2f ticks/op; f64 chain %.2f ticks/op\n", h[0], h[1],
         100.0 * h[0] / h[1], h[0] / 200000.0, h[2] / 200000.0);
  return 0;
}
