// Classifier head + loss (fp32 VALU: B x 768 x n_classes is far too small for MFMA tiles).
//   pooled = tanh(pre)                 (pre = pooler dense output, from the GEMM)
//   logits = dropout(pooled) Wc^T + bc (ref: vault/models/vault/model.py:547-550,567-570)
//   loss   = mean CrossEntropy         (ref: vault/tmsc_utils/trainer.py:241-242)
//            or mean BCE-with-logits  (ref: vault/models/vault/trainer.py:55-56; n_classes = 1, float targets)
// and the backward of all three.  One wave per sample.
#include "common.h"
#include "../../include/vault_hip.h"

namespace {

constexpr int MAXC = 8;

__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ pre, const float* __restrict__ Wc,
                                                       const float* __restrict__ bc, const long long* __restrict__ labels,
                                                       float* __restrict__ pooled, float* __restrict__ logits,
                                                       float* __restrict__ loss_sum, int B, int H, int C, float loss_scale,
                                                       uint32_t drop_thresh, uint32_t drop_seed, uint32_t drop_stream,
                                                       float drop_scale, const float* __restrict__ targets) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float acc[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) acc[c] = 0.f;
  for (int n = lane; n < H; n += 64) {
    const float t = tanhf(pre[(size_t)b * H + n]);
    pooled[(size_t)b * H + n] = t;
    float z = t;
    if (drop_thresh != 0u) z = dropout_keep(drop_seed, drop_stream, (uint32_t)(b * H + n), drop_thresh) ? t * drop_scale : 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) acc[c] += z * Wc[(size_t)c * H + n];
  }
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) {
      acc[c] = wave_sum(acc[c]) + bc[c];
      mx = fmaxf(mx, acc[c]);
    }
  if (lane == 0) {
    float se = 0.f;
    for (int c = 0; c < C; ++c) {
      logits[(size_t)b * C + c] = acc[c];
      se += expf(acc[c] - mx);
    }
    if (targets != nullptr && loss_sum != nullptr) {
      // BCE with logits, the numerically stable form torch uses: max(x, 0) - x y + log(1 + exp(-|x|))
      const float x = acc[0], y = targets[b];
      atomicAdd(loss_sum, (fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)))) * loss_scale);
    } else if (labels != nullptr && loss_sum != nullptr) {
      const int y = (int)labels[b];
      float ly = 0.f;
      for (int c = 0; c < C; ++c) if (c == y) ly = acc[c];
      atomicAdd(loss_sum, (mx + logf(se) - ly) * loss_scale);
    }
  }
}

// dlogits = (softmax - onehot) * gscale (or a caller-provided dlogits); dWc += dlogits^T z ; dbc += dlogits ;
// dpre = (Wc^T dlogits) * dropmask * (1 - pooled^2)  -> bf16 [B][H]
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ pooled, const float* __restrict__ logits,
                                                       const float* __restrict__ dlogits_in,
                                                       const long long* __restrict__ labels, const float* __restrict__ Wc,
                                                       float* __restrict__ dWc, float* __restrict__ dbc,
                                                       h16* __restrict__ dpre, int B, int H, int C, float gscale,
                                                       uint32_t drop_thresh, uint32_t drop_seed, uint32_t drop_stream,
                                                       float drop_scale, const float* __restrict__ targets) {
  H16_SATURATE();
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float dl[MAXC];
  if (dlogits_in != nullptr) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) dl[c] = (c < C) ? dlogits_in[(size_t)b * C + c] : 0.f;
  } else if (targets != nullptr) {   // d BCE-with-logits / d logit = sigmoid(x) - y
#pragma unroll
    for (int c = 0; c < MAXC; ++c) dl[c] = 0.f;
    const float x = logits[b];
    dl[0] = (1.f / (1.f + expf(-x)) - targets[b]) * gscale;
  } else {
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      dl[c] = (c < C) ? logits[(size_t)b * C + c] : -INFINITY;
      mx = fmaxf(mx, dl[c]);
    }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      dl[c] = (c < C) ? expf(dl[c] - mx) : 0.f;
      se += dl[c];
    }
    const int y = (int)labels[b];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) dl[c] = (dl[c] / se - (c == y ? 1.f : 0.f)) * gscale;
  }
  for (int n = lane; n < H; n += 64) {
    const float t = pooled[(size_t)b * H + n];
    float keep = 1.f;
    if (drop_thresh != 0u) keep = dropout_keep(drop_seed, drop_stream, (uint32_t)(b * H + n), drop_thresh) ? drop_scale : 0.f;
    const float z = t * keep;
    float dz = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) {
        dz += dl[c] * Wc[(size_t)c * H + n];
        atomicAdd(dWc + (size_t)c * H + n, dl[c] * z);
      }
    dpre[(size_t)b * H + n] = (h16)(dz * keep * (1.f - t * t));
  }
  if (lane == 0)
    for (int c = 0; c < C; ++c) atomicAdd(dbc + c, dl[c]);
}

// dpre_bf16 = bf16(dpooled * (1 - pooled^2))   (VaultModel path: gradient arrives at pooler_output)
__global__ void tanh_bwd_kernel(const float* __restrict__ pooled, const float* __restrict__ dpooled,
                                h16* __restrict__ dpre, long long n) {
  H16_SATURATE();
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256ll) {
    const float t = pooled[i];
    dpre[i] = (h16)(dpooled[i] * (1.f - t * t));
  }
}

// y_bf16 = gelu(x)   /   dx = dy * gelu'(x)      (MLP task heads: Linear - LayerNorm - GELU - Linear on the pooled output)
__global__ void gelu_fwd_kernel(const float* __restrict__ x, h16* __restrict__ y, long long n) {
  H16_SATURATE();
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256ll) y[i] = (h16)gelu_f(x[i]);
}
__global__ void gelu_fwd_f32_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256ll) y[i] = gelu_f(x[i]);
}
__global__ void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256ll)
    dx[i] = dy[i] * dgelu_f(x[i]);
}

}  // namespace

extern "C" int vault_gelu_fwd(const float* x, void* y_bf16, long long n, void* stream) {
  if (!x || !y_bf16 || n <= 0) return VAULT_EINVAL;
  const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     reinterpret_cast<h16*>(y_bf16), n);
  return (int)hipGetLastError();
}

extern "C" int vault_gelu_fwd_f32(const float* x, float* y, long long n, void* stream) {
  if (!x || !y || n <= 0) return VAULT_EINVAL;
  const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(gelu_fwd_f32_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, n);
  return (int)hipGetLastError();
}

extern "C" int vault_gelu_bwd(const float* x, const float* dy, float* dx, long long n, void* stream) {
  if (!x || !dy || !dx || n <= 0) return VAULT_EINVAL;
  const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, dy, dx, n);
  return (int)hipGetLastError();
}

extern "C" int vault_head_fwd(const vault_head_args* a, void* stream) {
  if (!a || !a->pre || !a->pooled || a->B <= 0 || a->C > MAXC || a->C < 0) return VAULT_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->C > 0 && (!a->Wc || !a->bc || !a->logits)) return VAULT_EINVAL;
  if (a->loss_kind < 0 || a->loss_kind > 1 || (a->loss_kind == 1 && a->loss_sum && (a->C != 1 || !a->targets))) return VAULT_EINVAL;
  hipLaunchKernelGGL(head_fwd_kernel, dim3((a->B + 3) / 4), dim3(256), 0, st, a->pre, a->Wc, a->bc,
                     reinterpret_cast<const long long*>(a->labels), a->pooled, a->logits, a->loss_sum, a->B, a->H, a->C,
                     a->loss_scale, a->drop_thresh, a->drop_seed, a->drop_stream, a->drop_scale,
                     a->loss_kind == 1 ? a->targets : nullptr);
  return (int)hipGetLastError();
}

extern "C" int vault_head_bwd(const vault_head_args* a, void* stream) {
  if (!a || !a->pooled || !a->dpre_bf16 || a->B <= 0 || a->C > MAXC || a->C <= 0) return VAULT_EINVAL;
  if (a->loss_kind < 0 || a->loss_kind > 1) return VAULT_EINVAL;
  if (a->dlogits == nullptr && a->loss_kind == 1 && (a->C != 1 || a->targets == nullptr || a->logits == nullptr)) return VAULT_EINVAL;
  if (a->dlogits == nullptr && a->loss_kind == 0 && (a->labels == nullptr || a->logits == nullptr)) return VAULT_EINVAL;
  hipLaunchKernelGGL(head_bwd_kernel, dim3((a->B + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     a->pooled, a->logits, a->dlogits, reinterpret_cast<const long long*>(a->labels), a->Wc, a->dWc,
                     a->dbc, reinterpret_cast<h16*>(a->dpre_bf16), a->B, a->H, a->C, a->grad_scale, a->drop_thresh,
                     a->drop_seed, a->drop_stream, a->drop_scale, a->loss_kind == 1 ? a->targets : nullptr);
  return (int)hipGetLastError();
}

extern "C" int vault_tanh_bwd(const float* pooled, const float* dpooled, void* dpre_bf16, long long n, void* stream) {
  if (!pooled || !dpooled || !dpre_bf16 || n <= 0) return VAULT_EINVAL;
  const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pooled, dpooled,
                     reinterpret_cast<h16*>(dpre_bf16), n);
  return (int)hipGetLastError();
}
