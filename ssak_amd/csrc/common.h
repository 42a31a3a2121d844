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
// Wave-wide reductions by DPP (quad swaps, row half mirror / mirror, row_bcast15 / 31, then v_readlane of lane 63): six
// VALU-rate steps; the __shfl_xor butterfly they replace went through ds_bpermute (LDS crossbar latency on every step),
// which made the one-wave-per-row normalisation kernels latency-bound.  The result is uniform (an SGPR broadcast).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                                ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1, 0xf>(0.f, v);   // quad_perm [1,0,3,2]
  v += dpp_f<0x4E, 0xf>(0.f, v);   // quad_perm [2,3,0,1]
  v += dpp_f<0x141, 0xf>(0.f, v);  // row_half_mirror
  v += dpp_f<0x140, 0xf>(0.f, v);  // row_mirror: every lane holds its 16-lane row total
  v += dpp_f<0x142, 0xa>(0.f, v);  // row_bcast15 into rows 1 and 3
  v += dpp_f<0x143, 0xc>(0.f, v);  // row_bcast31 into rows 2 and 3: lane 63 holds the wave total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f<0xB1, 0xf>(v, v));
  v = fmaxf(v, dpp_f<0x4E, 0xf>(v, v));
  v = fmaxf(v, dpp_f<0x141, 0xf>(v, v));
  v = fmaxf(v, dpp_f<0x140, 0xf>(v, v));
  v = fmaxf(v, dpp_f<0x142, 0xa>(v, v));
  v = fmaxf(v, dpp_f<0x143, 0xc>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is also a memory fence: the compiler puts `s_waitcnt vmcnt(0)`
// in front of it, so a wave with global stores (or prefetches) in flight sits out their round trip.  Where the barrier only
// hands LDS contents from wave to wave, this is enough.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
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
// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, far below bf16 resolution): 2 transcendentals +
// ~10 VALU ops instead of the branchy libm erff -- GELU sits in GEMM epilogues and in the conv0 store pass,
// where the VALU, not HBM, was the limiter.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
// GELU (erf form, the activation of the conv feature encoder and of the FFN): x * Phi(x) with the normal CDF written as
// a logistic function of an odd polynomial, Phi(x) = 1 / (1 + exp2(q(x))), q(x) = -log2(e) * x (c0 + c1 x^2 + c2 x^4 +
// c3 x^6), x clamped to [-6, 6].  Coefficients fitted to logit(Phi) for max |Phi error| = 1.6e-5 (fp32 evaluation
// included; bf16 resolution is 3.9e-3), i.e. the same function as the reference's exact GELU at every precision this
// engine stores.  7 VALU + exp + rcp per element instead of 15 + 2 for the Abramowitz-Stegun erf: the GELU / GELU'
// GEMM epilogues were VALU-bound (tools/probes/p8_probe.hip: 18 us of epilogue after a 19 us K = 768 main loop).  The
// two-element forms keep everything but the transcendentals in packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32).
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define SSAK_PHI_C0 (-2.302147388458252f)
#define SSAK_PHI_C1 (-0.10512793809175491f)
#define SSAK_PHI_C2 (0.00039503577863797545f)
#define SSAK_PHI_C3 (5.9617443184833974e-05f)
__device__ __forceinline__ f32x2 phi2(f32x2 x) {
  const f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -6.f, 6.f), __builtin_amdgcn_fmed3f(x[1], -6.f, 6.f)};
  const f32x2 s = xc * xc;
  f32x2 q = __builtin_elementwise_fma(s, (f32x2){SSAK_PHI_C3, SSAK_PHI_C3}, (f32x2){SSAK_PHI_C2, SSAK_PHI_C2});
  q = __builtin_elementwise_fma(q, s, (f32x2){SSAK_PHI_C1, SSAK_PHI_C1});
  q = __builtin_elementwise_fma(q, s, (f32x2){SSAK_PHI_C0, SSAK_PHI_C0});
  q = q * xc;
  const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])} + (f32x2){1.f, 1.f};
  return (f32x2){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x2 gelu2(f32x2 x) { return x * phi2(x); }
__device__ __forceinline__ f32x2 gelu_grad2(f32x2 x) {
  // d/dx [x Phi(x)] = Phi(x) + x phi(x), phi(x) = exp(-x^2 / 2) / sqrt(2 pi)
  const f32x2 t = x * x * (f32x2){-0.72134752044448170f, -0.72134752044448170f};
  const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  return __builtin_elementwise_fma(x * (f32x2){0.39894228040143268f, 0.39894228040143268f}, e, phi2(x));
}
__device__ __forceinline__ float gelu_f(float x) { return gelu2((f32x2){x, x})[0]; }
__device__ __forceinline__ float gelu_grad_f(float x) { return gelu_grad2((f32x2){x, x})[0]; }
// ---- storage-type helpers -----------------------------------------------------------------------------------------
// Every activation kernel is a template over the element type T it loads and stores: bf16 in the production engine, float
// in the fp32-exact verification mode (ssak_w2v2_config.exact).  Both modes run the same kernel source; what changes with T
// is the width of a chunk in memory and -- for T = float -- the exact erf GELU instead of the 1.6e-5 logistic fit.
template <typename T>
struct Chunk8;  // 8 consecutive elements as they sit in memory
template <>
struct Chunk8<bf16> {
  uint4 q;
};
template <>
struct Chunk8<float> {
  float4 a, b;
};
template <typename T>
struct Chunk4;
template <>
struct Chunk4<bf16> {
  uint2 q;
};
template <>
struct Chunk4<float> {
  float4 a;
};
template <typename T>
__device__ __forceinline__ Chunk8<T> ld8(const T* p) {
  return *reinterpret_cast<const Chunk8<T>*>(p);
}
template <typename T>
__device__ __forceinline__ void st8(T* p, const Chunk8<T>& c) {
  *reinterpret_cast<Chunk8<T>*>(p) = c;
}
template <typename T>
__device__ __forceinline__ Chunk4<T> ld4(const T* p) {
  return *reinterpret_cast<const Chunk4<T>*>(p);
}
template <typename T>
__device__ __forceinline__ void st4(T* p, const Chunk4<T>& c) {
  *reinterpret_cast<Chunk4<T>*>(p) = c;
}
__device__ __forceinline__ void chunk_to_f(const Chunk8<bf16>& c, float* f) {
  const uint32_t w[4] = {c.q.x, c.q.y, c.q.z, c.q.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void chunk_to_f(const Chunk8<float>& c, float* f) {
  *reinterpret_cast<float4*>(f) = c.a;
  *reinterpret_cast<float4*>(f + 4) = c.b;
}
__device__ __forceinline__ void chunk_to_f(const Chunk4<bf16>& c, float* f) {
  const uint32_t w[2] = {c.q.x, c.q.y};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void chunk_to_f(const Chunk4<float>& c, float* f) { *reinterpret_cast<float4*>(f) = c.a; }
template <typename T>
__device__ __forceinline__ Chunk8<T> f_to_chunk8(const float* f);
template <>
__device__ __forceinline__ Chunk8<bf16> f_to_chunk8<bf16>(const float* f) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16x2 t = {(bf16)f[2 * i], (bf16)f[2 * i + 1]};
    w[i] = __builtin_bit_cast(uint32_t, t);
  }
  Chunk8<bf16> c;
  c.q = make_uint4(w[0], w[1], w[2], w[3]);
  return c;
}
template <>
__device__ __forceinline__ Chunk8<float> f_to_chunk8<float>(const float* f) {
  Chunk8<float> c;
  c.a = *reinterpret_cast<const float4*>(f);
  c.b = *reinterpret_cast<const float4*>(f + 4);
  return c;
}
template <typename T>
__device__ __forceinline__ Chunk4<T> f_to_chunk4(const float* f);
template <>
__device__ __forceinline__ Chunk4<bf16> f_to_chunk4<bf16>(const float* f) {
  const bf16x2 t0 = {(bf16)f[0], (bf16)f[1]}, t1 = {(bf16)f[2], (bf16)f[3]};
  Chunk4<bf16> c;
  c.q = make_uint2(__builtin_bit_cast(uint32_t, t0), __builtin_bit_cast(uint32_t, t1));
  return c;
}
template <>
__device__ __forceinline__ Chunk4<float> f_to_chunk4<float>(const float* f) {
  Chunk4<float> c;
  c.a = *reinterpret_cast<const float4*>(f);
  return c;
}
// GELU / GELU' at the accuracy of the storage type
template <typename T>
__device__ __forceinline__ float gelu_s(float x) {
  if constexpr (sizeof(T) == 4) return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
  else return gelu_f(x);
}
template <typename T>
__device__ __forceinline__ float gelu_grad_s(float x) {
  if constexpr (sizeof(T) == 4)
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
  else return gelu_grad_f(x);
}
// Development switches read from the environment exist only in builds made with -DSSAK_DEV (`make DEV=1`); in the release
// library the expression is a null constant and the code behind it folds away.
#ifdef SSAK_DEV
#define SSAK_DEV_ENV(name) getenv(name)
#else
#define SSAK_DEV_ENV(name) (static_cast<const char*>(nullptr))
#endif
// Counter-based RNG for dropout masks: the forward and backward kernels recompute the same bits from (seed, site, row,
// column); no mask tensor is stored.  Round 5 form -- ONE full-rate integer multiply per element:
//     word(row, col) = rowkey(seed, site, row) * colmul(col)  (mod 2^32),      keep iff word >= thresh16 << 16
// rowkey = the "lowbias32" mixer of the row index under (seed, site), forced odd: one evaluation per row of a row kernel / per
// accumulator row of an epilogue / per (utterance, head, query) of the attention kernels; colmul = the same mixer of the column
// index under a fixed key, forced odd: a function of the column alone, which a kernel keeps in registers (row kernels, the
// attention kernels' keys) or in LDS next to its tiles.  The top bits of an odd x odd product are multiply-shift hashing in
// both arguments; measured on 4 096 x 3 072 masks (oracle/dropout_hash.py `quality_report`, tests/test_oracle.py): drop rate
// within 1.5 sigma for p = 0.05 .. 0.5, |rho| < 1.1e-3 between neighbouring columns / rows / diagonals / sites / seeds, row and
// column drop counts binomial.  Rounds 1-4 spent a two-multiply hash per element PAIR plus field extraction (7 VALU per
// element in the row kernels and epilogues, 5.5 in the attention kernels); this form is mul + compare + select = 3.
// Every site uses it: element-wise sites with (row, col) of their [M, C] tensor, attention with row = (b * nh + h) * F + q and
// col = key.
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint32_t stream, uint64_t idx) {
  const uint32_t key = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B1u) ^ (stream * 0x85EBCA77u) ^
                       ((uint32_t)(idx >> 32) * 0xC2B2AE3Du);
  uint32_t h = (uint32_t)idx ^ key;
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}
constexpr uint64_t DROP_COL_SEED = 0x5EED0C01A11CE5ull;
constexpr uint32_t DROP_COL_STREAM = 0x51u;
__device__ __forceinline__ uint32_t drop_rowkey(uint64_t seed, uint32_t stream, uint64_t row) { return hash_u32(seed, stream, row) | 1u; }
__device__ __forceinline__ uint32_t drop_colmul(uint32_t col) { return hash_u32(DROP_COL_SEED, DROP_COL_STREAM, (uint64_t)col) | 1u; }
// colmul of the first DROP_TABLE_N columns as a compile-time table in the code object's read-only data (each translation unit
// that uses it carries its own 64 KB copy: device symbols do not link across objects without relocatable device code).  Row
// kernels and GEMM epilogues read their columns' entries like they read a bias slice; a site with more columns than the table
// is refused by its launcher.
constexpr int DROP_TABLE_N = 16384;
constexpr uint32_t drop_colmul_host(uint32_t col) {
  const uint32_t key = (uint32_t)DROP_COL_SEED ^ ((uint32_t)(DROP_COL_SEED >> 32) * 0x9E3779B1u) ^ (DROP_COL_STREAM * 0x85EBCA77u);
  uint32_t h = col ^ key;
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h | 1u;
}
// (read through 16-byte vector loads in gemm_common.h and norm_act.hip's load_colmul: the alignment is part of the type)
struct alignas(16) DropColmulTable {
  uint32_t v[DROP_TABLE_N];
  constexpr DropColmulTable() : v() {
    for (int i = 0; i < DROP_TABLE_N; ++i) v[i] = drop_colmul_host((uint32_t)i);
  }
};
static_assert(alignof(DropColmulTable) >= 16, "g_drop_colmul is read as uint4");
#define SSAK_DEFINE_DROP_TABLE static __device__ const DropColmulTable g_drop_colmul = DropColmulTable();
// keep test on a word; thi = thresh16 << 16 (thresh16 = round(p * 65536): the realised drop probability is thresh16 / 65536)
__device__ __forceinline__ bool drop_keep(uint32_t rowkey, uint32_t colmul, uint32_t thi) { return rowkey * colmul >= thi; }
// the whole thing for one element (debug / exact-mode / cold paths: two hashes per call)
__device__ __forceinline__ bool keep_bit(uint64_t seed, uint32_t stream, uint64_t row, uint32_t col, uint32_t thresh16) {
  return drop_keep(drop_rowkey(seed, stream, row), drop_colmul(col), thresh16 << 16);
}
#endif
