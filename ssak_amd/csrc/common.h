// Shared device/host helpers for the ssak_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ssak_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define SSAK_WAVE 64

// status plumbing: every C-ABI entry returns an int; the message is kept per thread
void ssak_set_error(const char* fmt, ...);
#define SSAK_REQUIRE(cond, ...)              \
  do {                                       \
    if (!(cond)) {                           \
      ssak_set_error(__VA_ARGS__);           \
      return SSAK_ERR_INVALID;               \
    }                                        \
  } while (0)
#define SSAK_LAUNCH_CHECK()                                               \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      ssak_set_error("%s:%d launch: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return SSAK_ERR_LAUNCH;                                             \
    }                                                                     \
  } while (0)
#define SSAK_HIP(call)                                                    \
  do {                                                                    \
    hipError_t e__ = (call);                                              \
    if (e__ != hipSuccess) {                                              \
      ssak_set_error("%s:%d %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      return SSAK_ERR_LAUNCH;                                             \
    }                                                                     \
  } while (0)

static inline int ssak_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block reductions for blockDim.x <= 1024 (<= 16 waves); `red` is >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = -INFINITY;
  for (int i = 0; i < nw; ++i) r = fmaxf(r, red[i]);
  return r;
}
// exact (erf) GELU, the activation of both the conv feature encoder and the FFN
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// counter-based RNG for dropout masks: the forward and backward kernels recompute the same bit
// from (seed, stream, element index); no mask tensor is stored.
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint32_t stream, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1) + ((uint64_t)stream << 32);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 16);
}
__device__ __forceinline__ bool keep_bit(uint64_t seed, uint32_t stream, uint64_t idx, uint32_t thresh) {
  // keep with probability 1-p where thresh = p * 2^32 (thresh == 0 -> always keep)
  return hash_u32(seed, stream, idx) >= thresh;
}
#endif
