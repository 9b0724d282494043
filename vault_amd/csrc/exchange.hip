// Device side of the data-parallel gradient exchange (the reference is single-device:
// ref vault/tmsc_utils/trainer.py:353-369 has nothing to mirror).  HBM-bound byte work, no MFMA:
//   * row-sparse exchange of an embedding table's gradient: the union of the token ids of all ranks (sorted, the
//     same list on every rank), gather of those rows into a compact [U][H] buffer that is all-reduced, scatter back;
//   * bf16 wire with f32 accumulation: a rank's chunk of every peer's bf16 gradient summed in f32 in rank order
//     (identical on every rank that would compute it), rounded to bf16 once, widened back to f32 after the all-gather.
#include "common.h"
#include "../../include/vault_hip.h"

namespace {

// keys equal to -1 are padding (a rank with fewer tokens than its declared capacity); any other key outside [0, V) is an
// error of the caller that would leave a touched row out of the union (each rank would keep its own local gradient for
// it: silently diverging replicas) - counted in *bad, the host raises
__global__ __launch_bounds__(256) void mark_rows_kernel(const long long* __restrict__ keys, long long n, int V,
                                                        int* __restrict__ flags, int* __restrict__ bad) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256ll) {
    const long long k = keys[i];
    if (k >= 0 && k < V) flags[k] = 1;    // (benign race: every writer stores the same value)
    else if (k != -1) atomicAdd(bad, 1);
  }
}

// one 1024-thread block: exclusive scan of the V flags -> sorted list of the marked rows + their count; clears the flags
__global__ __launch_bounds__(1024) void compact_rows_kernel(int* __restrict__ flags, int V, long long* __restrict__ uniq,
                                                            int* __restrict__ count) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (V + 1023) / 1024;
  const int lo = t * per, hi = min(V, lo + per);
  int c = 0;
  for (int i = lo; i < hi; ++i) c += flags[i];
  part[t] = c;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {       // Hillis-Steele inclusive scan
    const int v = (t >= o) ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int slot = part[t] - c;
  for (int i = lo; i < hi; ++i)
    if (flags[i]) {
      uniq[slot++] = i;
      flags[i] = 0;
    }
  if (t == 1023) *count = part[1023];
}

// dir 0: out[j][:] = table[idx[j]][:]   dir 1: table[idx[j]][:] = out[j][:]   (H % 4 == 0; one block per row)
__global__ __launch_bounds__(256) void rows_move_kernel(float* __restrict__ table, const long long* __restrict__ idx,
                                                        float* __restrict__ compact, int H, int dir) {
  const long long r = idx[blockIdx.x];
  f32x4* a = reinterpret_cast<f32x4*>(table + r * H);
  f32x4* b = reinterpret_cast<f32x4*>(compact + (long long)blockIdx.x * H);
  for (int i = threadIdx.x; i < H / 4; i += 256) {
    if (dir == 0) b[i] = a[i]; else a[i] = b[i];
  }
}

// out[i] = bf16( sum_{k < n_src} f32(src[k * chunk + i]) ), k ascending (the same order on every rank)
__global__ __launch_bounds__(256) void sum_chunks_bf16_kernel(const h16* __restrict__ src, int n_src, long long chunk8,
                                                              h16* __restrict__ out) {
  H16_SATURATE();
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < chunk8; i += (long long)gridDim.x * 256ll) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < n_src; ++k) {
      const h16x8 v = reinterpret_cast<const h16x8*>(src)[(long long)k * chunk8 + i];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
    }
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (h16)acc[e];
    reinterpret_cast<h16x8*>(out)[i] = o;
  }
}

__global__ __launch_bounds__(256) void widen_bf16_kernel(const h16* __restrict__ x, float* __restrict__ y, long long n4) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256ll) {
    const h16x4 v = reinterpret_cast<const h16x4*>(x)[i];
    reinterpret_cast<f32x4*>(y)[i] = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
}

inline int grid_for(long long n) { return (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256); }

}  // namespace

extern "C" int vault_rows_union(const long long* keys, long long n_keys, int V, int* flags_zeroed, long long* uniq,
                                int* count, void* stream) {
  if (!keys || !flags_zeroed || !uniq || !count || n_keys <= 0 || V <= 0) return VAULT_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(count + 1, 0, sizeof(int), s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(mark_rows_kernel, dim3(grid_for(n_keys)), dim3(256), 0, s, keys, n_keys, V, flags_zeroed, count + 1);
  hipLaunchKernelGGL(compact_rows_kernel, dim3(1), dim3(1024), 0, s, flags_zeroed, V, uniq, count);
  return (int)hipGetLastError();
}

extern "C" int vault_rows_gather_f32(const float* table, const long long* idx, int n_rows, int H, float* out, void* stream) {
  if (!table || !idx || !out || n_rows <= 0 || H <= 0 || (H & 3)) return VAULT_EINVAL;
  hipLaunchKernelGGL(rows_move_kernel, dim3(n_rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     const_cast<float*>(table), idx, out, H, 0);
  return (int)hipGetLastError();
}

extern "C" int vault_rows_scatter_f32(const float* src, const long long* idx, int n_rows, int H, float* table, void* stream) {
  if (!table || !idx || !src || n_rows <= 0 || H <= 0 || (H & 3)) return VAULT_EINVAL;
  hipLaunchKernelGGL(rows_move_kernel, dim3(n_rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), table, idx,
                     const_cast<float*>(src), H, 1);
  return (int)hipGetLastError();
}

extern "C" int vault_sum_chunks_bf16(const void* src_bf16, int n_src, long long chunk, void* out_bf16, void* stream) {
  if (!src_bf16 || !out_bf16 || n_src <= 0 || chunk <= 0 || (chunk & 7)) return VAULT_EINVAL;
  hipLaunchKernelGGL(sum_chunks_bf16_kernel, dim3(grid_for(chunk / 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const h16*>(src_bf16), n_src, chunk / 8, reinterpret_cast<h16*>(out_bf16));
  return (int)hipGetLastError();
}

extern "C" int vault_widen_bf16(const void* x_bf16, float* y, long long n, void* stream) {
  if (!x_bf16 || !y || n <= 0 || (n & 3)) return VAULT_EINVAL;
  hipLaunchKernelGGL(widen_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const h16*>(x_bf16), y, n / 4);
  return (int)hipGetLastError();
}
