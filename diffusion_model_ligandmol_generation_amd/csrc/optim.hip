// Optimiser step of the trainer (train.py:196-197): global-norm gradient clipping + AdamW over ALL parameters in
// two launches.  The parameter gradients of one backward pass are slices of one flat buffer (backward.py), so the
// norm is one reduction over that buffer and the update is one multi-tensor kernel driven by a table of pointers.
#include "common.h"

struct AdamWTensor { float* param; float* exp_avg; float* exp_avg_sq; int64_t grad_offset; int64_t numel; };

#define OPT_CHUNK 2048   // elements per workgroup of the update kernel (256 lanes x 8)

__global__ __launch_bounds__(256) void k_sumsq_partial(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  float s = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 4 <= n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
      s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    } else {
      for (int64_t j = i; j < n; ++j) s += g[j] * g[j];
    }
  }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// norm2[0] = total norm, norm2[1] = clip coefficient min(1, max_norm / (norm + 1e-6)) (1 when max_norm <= 0)
__global__ __launch_bounds__(256) void k_sumsq_final(const float* __restrict__ part, int n, float max_norm, float* __restrict__ norm2) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float nrm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    norm2[0] = nrm;
    float c = 1.0f;
    if (max_norm > 0.f) { c = max_norm / (nrm + 1e-6f); c = c < 1.0f ? c : 1.0f; }
    norm2[1] = c;
  }
}
struct AdamWScalars { float lr, beta1, beta2, eps, weight_decay, bias1, sqrt_bias2; };
__global__ __launch_bounds__(256) void k_adamw(const AdamWTensor* __restrict__ tab, const int32_t* __restrict__ chunk_tensor,
                                               const int32_t* __restrict__ chunk_start, float* __restrict__ flat,
                                               const float* __restrict__ norm2, AdamWScalars a) {
  const AdamWTensor t = tab[chunk_tensor[blockIdx.x]];
  const int64_t i0 = (int64_t)chunk_start[blockIdx.x] * OPT_CHUNK;
  const float coef = norm2[1];
  const float step_size = a.lr / a.bias1, decay = 1.0f - a.lr * a.weight_decay;
  float* g = flat + t.grad_offset;
#pragma unroll
  for (int k = 0; k < OPT_CHUNK / 256; ++k) {
    const int64_t i = i0 + k * 256 + threadIdx.x;
    if (i < t.numel) {
      const float gr = g[i] * coef;                       // clip_grad_norm_ scales the gradient in place
      g[i] = gr;
      float p = t.param[i] * decay;                       // decoupled weight decay
      float m = t.exp_avg[i];
      m = m + (gr - m) * (1.0f - a.beta1);                // lerp
      const float v = t.exp_avg_sq[i] * a.beta2 + (gr * gr) * (1.0f - a.beta2);
      const float denom = sqrtf(v) / a.sqrt_bias2 + a.eps;
      p -= step_size * (m / denom);
      t.param[i] = p; t.exp_avg[i] = m; t.exp_avg_sq[i] = v;
    }
  }
}

extern "C" size_t ligdiff_adamw_workspace_bytes(void) { return 1024 * 4 + 256; }

extern "C" int ligdiff_adamw_step(const void* table_dev, const int32_t* chunk_tensor_dev, const int32_t* chunk_start_dev,
                                  int32_t n_chunks, float* flat_grad, int64_t flat_numel, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, float bias_correction1, float bias_correction2_sqrt,
                                  float max_norm, int32_t compute_norm, float* norm2_dev, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (!table_dev || !chunk_tensor_dev || !chunk_start_dev || n_chunks <= 0 || !flat_grad || flat_numel <= 0 || !norm2_dev ||
      !workspace || workspace_bytes < ligdiff_adamw_workspace_bytes() || !(bias_correction1 > 0.f) ||
      !(bias_correction2_sqrt > 0.f))
    return LIGDIFF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (compute_norm) {   // one norm per step: a caller that updates several step groups computes it with the first
    float* part = reinterpret_cast<float*>(workspace);
    int nb = (int)((flat_numel + 4095) / 4096);
    nb = nb < 1 ? 1 : nb > 1024 ? 1024 : nb;
    hipLaunchKernelGGL(k_sumsq_partial, dim3(nb), dim3(256), 0, s, flat_grad, flat_numel, part);
    hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(256), 0, s, part, nb, max_norm, norm2_dev);
  }
  AdamWScalars a{lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt};
  hipLaunchKernelGGL(k_adamw, dim3(n_chunks), dim3(256), 0, s, reinterpret_cast<const AdamWTensor*>(table_dev), chunk_tensor_dev,
                     chunk_start_dev, flat_grad, norm2_dev, a);
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}
