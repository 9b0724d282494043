// Rounding of v_cvt_pk_u8_f32 (development probe): prints the byte for x.25, x.5, x.75, negatives and > 255.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n) {
  int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0u);
}
int main() {
  float h[] = {0.25f, 0.5f, 0.75f, 1.5f, 2.5f, 3.5f, 25.49f, 25.5f, 25.51f, 26.5f, -0.6f, -3.f, 254.5f, 255.4f, 255.6f, 300.f};
  const int n = sizeof(h) / 4;
  float* d; unsigned* o; unsigned r[32];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
  hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%g -> %u\n", h[i], r[i]);
  return 0;
}
