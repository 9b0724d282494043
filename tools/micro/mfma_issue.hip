// Micro-benchmark behind DESIGN.md's "what limits the ring GEMM's main loop": one wave per SIMD (4 waves per CU,
// 512 registers each, as gemm256.hip runs), a loop of independent MFMAs with the main loop's other instructions
// added one kind at a time.  Prints chip-wide TFLOP/s per variant.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_issue.hip -o /tmp/mfma_issue && /tmp/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// VARIANT bits: 1 = 8 ds_read_b128 per 32 MFMAs, 2 = s_barrier per 32 MFMAs (lgkmcnt(0) in front), 4 = 4 global_load_lds
// per 32 MFMAs (counted vmcnt), 8 = 32x32x16 MFMAs (16 per "phase") instead of 16x16x32, 16 = setprio pair,
// 32 = memory instructions clustered after every 4th MFMA (as the old phase body) instead of spread
template <int V>
__global__ __launch_bounds__(256) void k(const char* src, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 a[4], b[4], nx[8];
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)(0.001f * (lane + i + j)); b[i][j] = (__bf16)(0.002f * (lane - i + j)); }
  }
  for (int i = 0; i < 8; ++i) nx[i] = a[i & 3];
  const uint32_t lds_rd = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 16384 + lane * 16;
  const char* gsrc = src + (size_t)blockIdx.x * 65536 + wave * 4096 + lane * 16;
  if constexpr ((V & 8) == 0) {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      if constexpr (V & 16) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a[m & 3]), "v"(b[(m >> 2) & 3]));
        constexpr bool cl = (V & 32) != 0;
        if constexpr (V & 1) {
          if (!cl && (m & 1) == 1 && m >= 2 && m < 18)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nx[(m - 2) >> 1]) : "v"(lds_rd), "i"(0));
          if (cl && (m & 3) == 3 && m >= 4 && m < 20) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(nx[(m - 4) >> 1]) : "v"(lds_rd));
            asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(nx[((m - 4) >> 1) + 1]) : "v"(lds_rd));
          }
        }
        if constexpr (V & 4) {
          if ((!cl && (m & 3) == 0 && m < 16) || (cl && (m & 3) == 3 && m < 16)) {
            const int pc = m >> 2;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                         :: "s"(__builtin_amdgcn_readfirstlane((int)(65536 + wave * 8192 + ((it & 1) * 4 + pc) * 1024))),
                            "v"(gsrc + ((it & 7) * 4 + pc) * 1024) : "memory");
          }
        }
      }
      if constexpr (V & 16) __builtin_amdgcn_s_setprio(0);
      if constexpr (V & 2) {
        if constexpr (V & 4) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else if constexpr (V & 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if constexpr (V & 1) {   // rotate the freshly read fragments in (keeps the reads alive; values are irrelevant)
#pragma unroll
        for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(nx[i]), "+v"(nx[i + 4])); }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += (float)nx[i][0];
    if (s == 12345.678f) sink[threadIdx.x] = s;
  } else {
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if constexpr (V & 16) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a[m & 3]), "v"(b[(m >> 2) & 3]));
        if constexpr (V & 1) {
          if (m >= 1 && m < 9) asm volatile("ds_read_b128 %0, %1" : "=v"(nx[m - 1]) : "v"(lds_rd));
        }
        if constexpr (V & 4) {
          if ((m & 1) == 0 && m < 8) {
            const int pc = m >> 1;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                         :: "s"(__builtin_amdgcn_readfirstlane((int)(65536 + wave * 8192 + ((it & 1) * 4 + pc) * 1024))),
                            "v"(gsrc + ((it & 7) * 4 + pc) * 1024) : "memory");
          }
        }
      }
      if constexpr (V & 16) __builtin_amdgcn_s_setprio(0);
      if constexpr (V & 2) {
        if constexpr (V & 4) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else if constexpr (V & 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if constexpr (V & 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(nx[i]), "+v"(nx[i + 4])); }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][15];
    for (int i = 0; i < 8; ++i) s += (float)nx[i][0];
    if (s == 12345.678f) sink[threadIdx.x] = s;
  }
}

template <int V>
int run(const char* name, const char* src, float* sink) {
  const int iters = 4000, lds = 160 * 1024 - 256;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), lds, 0, src, sink, iters);
  CHECK(hipEventRecord(e0));
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), lds, 0, src, sink, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double flops = 256.0 * 4 * iters * 32 * (2.0 * 16 * 16 * 32);   // same per "phase" for both shapes
  const double clk_per_phase = ms * 1e-3 / iters * 2.4e9;
  printf("%-58s %7.3f ms  %7.1f TFLOP/s  %6.0f clk/phase at 2.4 GHz (512 = peak)\n", name, ms, flops / ms * 1e-9, clk_per_phase);
  return 0;
}

int main() {
  char* src; float* sink;
  CHECK(hipMalloc(&src, 256 * 65536 + 65536));
  CHECK(hipMemset(src, 0, 256 * 65536 + 65536));
  CHECK(hipMalloc(&sink, 4096));
  run<0>("16x16x32: MFMA only", src, sink);
  run<8>("32x32x16: MFMA only", src, sink);
  run<1>("16x16x32 + 8 ds_read_b128 (spread)", src, sink);
  run<1 | 32>("16x16x32 + 8 ds_read_b128 (pairs after every 4th)", src, sink);
  run<8 | 1>("32x32x16 + 8 ds_read_b128", src, sink);
  run<2>("16x16x32 + barrier", src, sink);
  run<8 | 2>("32x32x16 + barrier", src, sink);
  run<4 | 2>("16x16x32 + 4 global_load_lds + vmcnt(20) + barrier", src, sink);
  run<8 | 4 | 2>("32x32x16 + 4 global_load_lds + vmcnt(20) + barrier", src, sink);
  run<1 | 2 | 4>("16x16x32 + reads + loads + barrier (spread)", src, sink);
  run<1 | 2 | 4 | 32>("16x16x32 + reads + loads + barrier (clustered)", src, sink);
  run<1 | 2 | 4 | 16>("16x16x32 + reads + loads + barrier + setprio", src, sink);
  run<8 | 1 | 2 | 4>("32x32x16 + reads + loads + barrier", src, sink);
  run<8 | 1 | 2 | 4 | 16>("32x32x16 + reads + loads + barrier + setprio", src, sink);
  return 0;
}
