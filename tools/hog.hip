// Development tool: a kernel that occupies N CUs for a given time (one block per CU: 150 KiB of LDS), standing in
// for an RCCL collective that shares the GPU with the GEMMs of a data-parallel step.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void hog_kernel(long long clocks, int* sink) {
  extern __shared__ char smem[];
  const long long t0 = wall_clock64();
  int x = 0;
  while (wall_clock64() - t0 < clocks) { __builtin_amdgcn_s_sleep(64); x++; }   // wall_clock64: 100 MHz
  if (threadIdx.x == 0 && x < 0) { smem[0] = 1; sink[0] = x; }
}
extern "C" int hog_launch(int blocks, long long ticks_100mhz, int* sink, void* stream) {
  static bool done = false;
  if (!done) { hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); done = true; }
  hipLaunchKernelGGL(hog_kernel, dim3(blocks), dim3(256), 150 * 1024, (hipStream_t)stream, ticks_100mhz, sink);
  return (int)hipGetLastError();
}
