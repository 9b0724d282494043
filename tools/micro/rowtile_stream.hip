// Micro-benchmark behind DESIGN.md "full-row GEMM tiles" (VERDICT r03 item 2a): what the STAGING of a GEMM whose block owns
// whole output rows (ROWS x 768 tiles: residual add + LayerNorm in the epilogue) costs on its own.  256 persistent
// 8-wave workgroups, one per CU; per K step of BK every workgroup pulls its ROWS x BK slice of the activations and the
// whole 768 x BK slice of the weight matrix through global_load_lds into a double-buffered LDS image ((ROWS + 768) x BK x 2 B
// per stage: BK = 64 does not fit twice into 160 KiB at 128 rows), waits for it (counted vmcnt) and passes a barrier -
// no LDS reads, no MFMAs, no epilogue: a floor for the main loop of such a kernel, to set against the MFMA time of the
// same tiles and against today's GEMM + LayerNorm pair.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/rowtile_stream.hip -o /tmp/rowtile_stream && /tmp/rowtile_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int ROWS, int BK>
__global__ __launch_bounds__(512, 1) void stream_kernel(const char* A, const char* W, int M, int K, int N, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = (ROWS + 768) * BK * 2;          // bytes per K step
  constexpr int PIECES = STAGE / 1024;                   // 1-KiB wave-instructions per K step
  constexpr int PPW = (PIECES + 7) / 8;                  // per wave
  constexpr int RPP = 1024 / (BK * 2);                   // rows per piece
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles = M / ROWS, nk = K / BK;
  const int sub = lane / (BK * 2 / 16), col = (lane % (BK * 2 / 16)) * 16;   // row inside a piece, byte column
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    auto issue = [&](int ks, int buf) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int pc = wave * PPW + i;
        if (pc < PIECES) {
          const int row = pc * RPP + sub;                // row of the (ROWS + 768)-row stage image
          const char* src = row < ROWS ? A + ((size_t)(t * ROWS + row) * K + (size_t)ks * BK) * 2 + col
                                       : W + ((size_t)(row - ROWS) * K + (size_t)ks * BK) * 2 + col;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(smem + buf * STAGE + pc * 1024), 16, 0, 0);
        }
      }
    };
    issue(0, 0);
    for (int ks = 0; ks < nk; ++ks) {
      if (ks + 1 < nk) {
        issue(ks + 1, (ks + 1) & 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (sink != nullptr && ks == nk - 1 && threadIdx.x == 0) sink[blockIdx.x] = *(int*)(smem + (ks & 1) * STAGE);
      __syncthreads();
    }
  }
}

template <int ROWS, int BK>
int run(const char* A, const char* W, int M, int K, int* sink, const char* what) {
  constexpr int LDS = 2 * (ROWS + 768) * BK * 2;
  auto kern = stream_kernel<ROWS, BK>;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), LDS, 0, A, W, M, K, 768, sink);
  CHECK(hipEventRecord(e0));
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), LDS, 0, A, W, M, K, 768, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / it;
  const int tiles = M / ROWS, rounds = (tiles + 255) / 256;
  const double bytes_cu = (double)rounds * (ROWS + 768) * K * 2;              // a CU with `rounds` tiles
  const double mfma_us = (double)rounds * 2.0 * ROWS * 768 * K / (2.5e15 / 256) * 1e6;
  printf("%-34s rows %3d BK %2d LDS %3d KiB: %4d tiles = %d rounds, %7.1f us staging alone (%5.1f GB/s per CU, %5.2f TB/s L2->LDS chip-wide); "
         "MFMA time of those rounds at the 2.5 PFLOP/s peak %6.1f us\n", what, ROWS, BK, LDS / 1024, tiles, rounds, us, bytes_cu / us / 1e3,
         bytes_cu * 256 / us / 1e6, mfma_us);
  return 0;
}

int main() {
  const int M = 47360;      // B = 256 x 185 fused tokens
  char *A, *W;
  int* sink;
  CHECK(hipMalloc(&A, (size_t)M * 3072 * 2));
  CHECK(hipMalloc(&W, (size_t)768 * 3072 * 2));
  CHECK(hipMalloc(&sink, 256 * sizeof(int)));
  std::vector<unsigned short> h((size_t)M * 3072);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff));   // random data (DVFS: never zeros)
  CHECK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(W, h.data(), (size_t)768 * 3072 * 2, hipMemcpyHostToDevice));
  printf("M = %d tokens, N = 768 (whole rows per workgroup), 256 workgroups of 8 waves\n", M);
  if (run<128, 32>(A, W, M, 768, sink, "attention-out (K = 768)")) return 1;
  if (run<64, 32>(A, W, M, 768, sink, "attention-out (K = 768)")) return 1;
  if (run<192, 32>(A, W, M, 768, sink, "attention-out (K = 768)")) return 1;
  if (run<256, 32>(A, W, M, 768, sink, "attention-out (K = 768)")) return 1;
  if (run<128, 32>(A, W, M, 3072, sink, "FFN-out (K = 3072)")) return 1;
  if (run<256, 32>(A, W, M, 3072, sink, "FFN-out (K = 3072)")) return 1;
  return 0;
}
