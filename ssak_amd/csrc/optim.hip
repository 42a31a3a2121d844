// a11: optimizer tail of the train step -- global grad-norm, clip, AdamW, bf16 weight shadow.  gfx950.
//
// Stands behind clip_grad_norm_ -> torch.optim.AdamW.step -> zero_grad as driven by HF Trainer
// (docker/transformers_modified/trainer.py:1827-1855; optim="adamw_torch", lr 1e-4, wd 0.0,
// ssak/train/transformers/wav2vec_train.py:353-384).  One pass over flat fp32 buffers (p, g, m, v):
// 16 B/param read + 12 B/param written (+2 B bf16 shadow) -- HBM-bound.  The clip coefficient is read from
// device memory (sum of squares produced by k_sumsq), so the step needs no host synchronisation.
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 q = reinterpret_cast<const float4*>(g)[i];
    s += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[blockIdx.x] = s;  // per-workgroup partial; summed in a fixed order by sumsq_final_kernel
}
__global__ __launch_bounds__(1024) void sumsq_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out,
                                                           int add) {
  __shared__ float red[16];
  float s = threadIdx.x < n ? partial[threadIdx.x] : 0.f;
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = add ? out[0] + s : s;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16* __restrict__ shadow, long n, const float* __restrict__ gnorm_sq,
                                                    float max_norm, float grad_scale, float lr, float beta1, float beta2,
                                                    float eps, float wd, float bc1, float bc2_sqrt) {
  float coef = grad_scale;
  if (gnorm_sq && max_norm > 0.f) {
    const float nrm = sqrtf(gnorm_sq[0]) * grad_scale;
    coef *= fminf(1.f, max_norm / (nrm + 1e-6f));
  }
  const float step_size = lr / bc1;
  const float decay = 1.f - lr * wd;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pe = &pp.x;
    const float* ge = &gg.x;
    float* me = &mm.x;
    float* ve = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = ge[k] * coef;
      me[k] = beta1 * me[k] + (1.f - beta1) * gk;
      ve[k] = beta2 * ve[k] + (1.f - beta2) * gk * gk;
      const float denom = sqrtf(ve[k]) / bc2_sqrt + eps;
      pe[k] = pe[k] * decay - step_size * (me[k] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (shadow) {
      const bf16x4 t = {(bf16)pp.x, (bf16)pp.y, (bf16)pp.z, (bf16)pp.w};
      reinterpret_cast<bf16x4*>(shadow)[i] = t;
    }
  }
  if (blockIdx.x == 0) {
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
      const float gk = g[i] * coef;
      const float mk = beta1 * m[i] + (1.f - beta1) * gk;
      const float vk = beta2 * v[i] + (1.f - beta2) * gk * gk;
      m[i] = mk;
      v[i] = vk;
      const float pk = p[i] * decay - step_size * (mk / (sqrtf(vk) / bc2_sqrt + eps));
      p[i] = pk;
      if (shadow) shadow[i] = (bf16)pk;
    }
  }
}

}  // namespace

int k_sumsq(const float* g, long n, float* out, float* partial, hipStream_t st, bool add) {
  if (n <= 0) return SSAK_OK;
  const int blocks = (int)fmin(1024.0, (double)ssak_cdiv(n, 1024));
  ProfScope prof_scope(PROF_SUMSQ, (double)n * 4.0, st);
  sumsq_kernel<<<blocks, 256, 0, st>>>(g, n, partial);
  SSAK_LAUNCH_CHECK();
  sumsq_final_kernel<<<1, 1024, 0, st>>>(partial, blocks, out, add ? 1 : 0);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

int k_adamw(float* p, const float* g, float* m, float* v, bf16* shadow, long n, const float* gnorm_sq, float max_norm,
            float grad_scale, float lr, float beta1, float beta2, float eps, float wd, int step, hipStream_t st) {
  if (n <= 0) return SSAK_OK;
  SSAK_REQUIRE(step >= 1, "adamw: step is 1-based");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  // p, g, m, v read (16 B) + p, m, v written (12 B) + the bf16 shadow (2 B) per parameter (SURVEY.md 8d counted reads only)
  ProfScope prof_scope(PROF_ADAMW, (double)n * (shadow ? 30.0 : 28.0), st);
  adamw_kernel<<<(int)fmin(4096.0, (double)ssak_cdiv(n, 1024)), 256, 0, st>>>(p, g, m, v, shadow, n, gnorm_sq, max_norm,
                                                                              grad_scale, lr, beta1, beta2, eps, wd, bc1, bc2s);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

extern "C" int ssak_grad_sumsq(const float* grads, long n, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(grads && out && n > 0, "grad_sumsq: bad arguments");
  SSAK_REQUIRE(((uintptr_t)grads & 15) == 0, "grad_sumsq: buffer must be 16-byte aligned");
  SSAK_REQUIRE(workspace && workspace_bytes >= 1024 * sizeof(float), "grad_sumsq: workspace of 4096 bytes needed");
  return k_sumsq(grads, n, out, (float*)workspace, (hipStream_t)stream);
}

extern "C" int ssak_grad_sumsq_add(const float* grads, long n, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(grads && out && n > 0, "grad_sumsq_add: bad arguments");
  SSAK_REQUIRE(((uintptr_t)grads & 15) == 0, "grad_sumsq_add: buffer must be 16-byte aligned");
  SSAK_REQUIRE(workspace && workspace_bytes >= 1024 * sizeof(float), "grad_sumsq_add: workspace of 4096 bytes needed");
  return k_sumsq(grads, n, out, (float*)workspace, (hipStream_t)stream, true);
}

extern "C" int ssak_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                               long n, const float* gnorm_sq, float max_norm, float grad_scale, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, void* stream) {
  SSAK_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0, "adamw: bad arguments");
  SSAK_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0 &&
                   ((uintptr_t)shadow_bf16 & 7) == 0, "adamw: buffers must be 16-byte aligned");
  return k_adamw(params, grads, exp_avg, exp_avg_sq, (bf16*)shadow_bf16, n, gnorm_sq, max_norm, grad_scale, lr, beta1,
                 beta2, eps, weight_decay, step, (hipStream_t)stream);
}
