// Internal C++ launch functions shared between the kernel files and the engine (not part of the C ABI).
#pragma once
#include "common.h"

// Per-launch timing slots (api.cpp; include/ssak_hip.h: ssak_prof_*).  The kernel classes below own the first slots; every
// GEMM instantiation registers a slot under its own name (as rocprofv3 prints it) the first time it is launched.
enum : int {
  PROF_ATTN_FWD = 0, PROF_ATTN_BWD, PROF_LN_FWD, PROF_LN_BWD, PROF_CONV0, PROF_ADAMW, PROF_SUMSQ, PROF_CTC, PROF_WAVE_NORM,
  PROF_ROWWISE, PROF_POSCONV_W, PROF_SOFTMAX, PROF_POSCONV_DIRECT, PROF_POSCONV_WGRAD, PROF_LOGMEL,
  PROF_CLASS_SLOTS
};
constexpr int PROF_MAX_SLOTS = 128;
int ssak_prof_register(const char* name, int bound);  // -> slot id (idempotent per name); thread-safe
bool ssak_prof_wanted(int slot, hipStream_t st);
// Brackets everything launched on `st` during its lifetime when the slot is being profiled; `work` = the ALGORITHMIC flops
// (MFMA-bound slots) or bytes (HBM / latency-bound slots) of what it covers.
class ProfScope {
 public:
  ProfScope(int slot, double work, hipStream_t st);
  ~ProfScope();
  ProfScope(const ProfScope&) = delete;
  ProfScope& operator=(const ProfScope&) = delete;

 private:
  hipStream_t st_;
  hipEvent_t e0_ = nullptr;
  int slot_;
  double work_;
  bool on_;
};

struct DropSpec {
  uint64_t seed = 0;
  uint32_t stream = 0;
  float p = 0.f;
};

// Deferred second stages of the two-stage column reductions (LayerNorm dgamma / dbeta / bias sums, column sums of bf16
// matrices, the GEMM epilogue's per-tile-row sums): out[n] += sum_k partial[k * stride + n].  Each used to be its own 4 us
// launch -- 50 per train step, a fifth of all launches.  While a sink is installed (the engine's backward does, per pair of
// encoder layers) the producers queue the job instead and k_reduce_flush runs all of them as ONE launch; the partial buffers
// must stay untouched until then.
struct ReduceJob {
  const float* partial;
  float* out;
  long stride;
  int slots, ncols;
};
struct ReduceSink {
  static constexpr int CAP = 32;
  ReduceJob jobs[CAP];
  int n = 0;
  bool push(const float* partial, long stride, int slots, int ncols, float* out) {
    if (n >= CAP) return false;
    jobs[n++] = ReduceJob{partial, out, stride, slots, ncols};
    return true;
  }
};
extern thread_local ReduceSink* g_reduce_sink;
int k_reduce_flush(ReduceSink& sink, hipStream_t st);

// norm_act.hip -- row / element-wise kernels, templates over the activation storage type T (bf16: production engine,
// float: fp32-exact verification mode); explicitly instantiated for both in norm_act.hip
template <typename T>
int k_layernorm_fwd_t(const T* y, const T* res, const float* gamma, const float* beta, T* r_out, T* out,
                      float* mean, float* rstd, int M, int C, float eps, const DropSpec& pre, const DropSpec& post,
                      hipStream_t st, const DropSpec& mid = DropSpec(), bool post_gelu = false);
template <typename T>
int k_layernorm_bwd_t(const T* g1, const T* g2, const T* r, const float* mean, const float* rstd,
                      const float* gamma, const T* g_res, T* dr, T* dy, float* dgamma, float* dbeta,
                      float* partial /*[LN_BWD_BLOCKS*3*C] scratch*/, int M, int C, const DropSpec& pre, const DropSpec& post,
                      hipStream_t st, const DropSpec& mid = DropSpec(), float* dy_colsum = nullptr /*[C] += column sums of dy*/,
                      const float* post_gelu_beta = nullptr /*forward was gelu(LN(x)): beta of that LN*/);
#ifndef SSAK_LN_BWD_BLOCKS
#define SSAK_LN_BWD_BLOCKS 768
#endif
#ifndef SSAK_LN_FWD_BLOCKS
#define SSAK_LN_FWD_BLOCKS 768
#endif
constexpr int LN_FWD_BLOCKS = SSAK_LN_FWD_BLOCKS;  // workgroups of the multi-row LayerNorm forward (3 per CU, three waves per SIMD, ~5 rows per wave at the train shape: 768 / 1024 / 2048 workgroups measured 473 / 490 / 488 us per step)
constexpr int LN_BWD_BLOCKS = SSAK_LN_BWD_BLOCKS;  // 3 waves per SIMD (the kernel takes 159 VGPRs): bytes in flight bound this kernel (one row + one prefetched row per wave)
template <typename T>
int k_softmax_fwd_t(const T* S, T* P, T* Pd, const int32_t* klens, int rows, int cols, int ld,
                    int rows_per_batch, const DropSpec& drop, hipStream_t st);
template <typename T>
int k_softmax_bwd_t(const T* dPd, const T* P, T* dS, int rows, int cols, int ld, const DropSpec& drop,
                    hipStream_t st);
// out[n] += column sums of X; with a scratch of >= min(64, ceil(M/32)) * N floats the sum is two-stage and deterministic
// (without one: float atomics).  rowmask / flens / F: only masked, non-padding rows (SpecAugment embedding gradient).
template <typename T>
int k_colsum_t(const T* X, long ld, int M, int N, float* out, hipStream_t st, float* scratch = nullptr, size_t scratch_floats = 0,
               const uint8_t* rowmask = nullptr, const int32_t* flens = nullptr, int F = 0);
int k_cast_f32_bf16(const float* in, bf16* out, long n, hipStream_t st);
template <typename T>
int k_specaug_fwd_t(T* h, const uint8_t* mask, const int32_t* flens, const float* embed, int B, int F, int C,
                    hipStream_t st);
template <typename T>
int k_specaug_bwd_t(T* dh, const uint8_t* mask, const int32_t* flens, float* dembed, int B, int F, int C,
                    hipStream_t st, float* scratch = nullptr, size_t scratch_floats = 0);
template <typename T>
int k_gelu_grad_mul_t(const T* dy, const T* pre, T* out, long n, hipStream_t st);
template <typename T>
int k_add_t(const T* a, const T* b, T* out, long n, hipStream_t st);
// bf16-named forms used by the other kernel files (sb_head.hip, ...)
int k_layernorm_fwd(const bf16* y, const bf16* res, const float* gamma, const float* beta, bf16* r_out, bf16* out,
                    float* mean, float* rstd, int M, int C, float eps, const DropSpec& pre, const DropSpec& post,
                    hipStream_t st, const DropSpec& mid = DropSpec(), bool post_gelu = false);
int k_layernorm_bwd(const bf16* g1, const bf16* g2, const bf16* r, const float* mean, const float* rstd,
                    const float* gamma, const bf16* g_res, bf16* dr, bf16* dy, float* dgamma, float* dbeta,
                    float* partial, int M, int C, const DropSpec& pre, const DropSpec& post,
                    hipStream_t st, const DropSpec& mid = DropSpec(), float* dy_colsum = nullptr,
                    const float* post_gelu_beta = nullptr);
int k_colsum(const bf16* X, long ld, int M, int N, float* out, hipStream_t st, float* scratch = nullptr, size_t scratch_floats = 0,
             const uint8_t* rowmask = nullptr, const int32_t* flens = nullptr, int F = 0);

// conv_frontend.hip
size_t k_conv0_stats_doubles(int B, int T0, int C);
template <typename T>
int k_conv0_gn_gelu_t(const float* x, const float* w, const float* gamma, const float* beta, T* out, double* stats,
                      int B, int Tn, int T0, int C, int ksize, int stride, hipStream_t st, bool raw_input = false);
// dw [C][ksize] += sum over frames of d[b,t,c] * x[b, stride*t + k] (conv0 weight gradient of the layer-norm feature encoder,
// Cin = 1); scratch >= k_conv0_wgrad_scratch_floats() floats; deterministic (per-workgroup partials, fixed-order sum)
size_t k_conv0_wgrad_scratch_floats(int B, int C, int ksize);
template <typename DT>
int k_conv0_wgrad_t(const DT* d, const float* x, float* dw, float* scratch, int B, int T, int T0, int C, int ksize, int stride,
                  hipStream_t st);
template <typename T>
int k_conv0_bias_t(const float* x, const float* w, const float* bias, T* out, int B, int Tn, int T0, int C, int ksize,
                   int stride, hipStream_t st);
template <typename T_>
int k_col2im_t(const T_* dxcol, T_* dx, int B, int Tin, int Tout, int C, int k, int s, hipStream_t st);
int k_sum_slabs(const float* slabs, int nb, long n, float* out, hipStream_t st);
size_t k_conv0_bwd_scratch_floats(int B, int T0, int C);
template <typename DT>
int k_conv0_gn_gelu_bwd_t(const float* x, const float* w, const float* gamma, const float* beta, const DT* dy,
                        const double* sums, float* scratch, float* dw, float* dgamma, float* dbeta, int B, int T, int T0, int C,
                        hipStream_t st);
template <typename T>
int k_conv_weight_rearrange_t(const float* w, T* out, int Co, int Ci, int k, hipStream_t st);
template <typename T>
int k_posconv_prepare_t(const float* g, const float* v, T* w_fwd, T* w_bwd, float* norms, int H, int G, int K,
                        hipStream_t st);
int k_posconv_weight_bwd(const float* dw, const float* g, const float* v, const float* norms, float* dg, float* dv,
                         int H, int G, int K, hipStream_t st);
template <typename T>
int k_posconv_pack_t(const T* h, T* pg, int B, int F, int H, int G, int K, hipStream_t st);

// attention.hip (fused, head_dim 64)
bool k_attention_supported(int H, int nh);
int k_attention_fwd(const bf16* qkv, bf16* ctx, float* lse, const int32_t* klens, int B, int F, int nh, int H,
                    const DropSpec& drop, hipStream_t st);
// bias_part / bias_grad (both or neither): the kernels leave the column sums of their rows of dqkv in bias_part
// [k_attention_bwd_bias_floats] and bias_grad [3H] += their total (through the caller's ReduceSink when there is one) -- the q|k|v bias gradient
size_t k_attention_bwd_bias_floats(int B, int F, int H);  // 0: the kernels' block sizes differ (development switch), sum dqkv's columns separately
int k_attention_bwd(const bf16* qkv, const bf16* ctx, const float* lse, const int32_t* klens, const bf16* dctx, float* delta,
                    bf16* dqkv, int B, int F, int nh, int H, const DropSpec& drop, int mode /* SSAK_ATTN_BWD_* */, hipStream_t st,
                    float* bias_part = nullptr, float* bias_grad = nullptr);
int k_colsum_rows(const float* partial, int slots, int N, float* out, hipStream_t st);  // out[n] += sum over slots of partial[slot][n]

// posconv.hip: the grouped positional convolution as a direct convolution (input window resident in LDS)
bool k_posconv_direct_supported(int H, int G, int K);
int k_posconv_frag_weights(const bf16* w, bf16* wfrag, int H, int G, int K, hipStream_t st);
int k_posconv_direct(const bf16* x, long rows_per_group, int row0, const bf16* w, const float* bias, bf16* out, bf16* pre, int B, int F,
                     int H, int G, int K, int gelu, hipStream_t st);

// ... and its weight gradient as a direct contraction over time (x, dy packed by k_posconv_pack_t; dwf [G][K * cg][cg] fp32)
size_t k_posconv_wgrad_scratch_floats(int H, int G, int K);
int k_posconv_wgrad_direct(const bf16* x, const bf16* dy, long rows_per_group, long rows, int lead, float* dwf, float* scratch, int H, int G,
                           int K, hipStream_t st);

// whisper_frontend.hip
// (T_ = bf16, or float in the fp32-exact mode)
template <typename T_>
int k_mel_to_cl_t(const float* mel, T_* cl, int B, int C, int T, int RS, int lead, hipStream_t st);
template <typename T_>
int k_add_rowvec_t(const T_* x, const T_* pos, T_* out, int B, int F, int H, hipStream_t st);
template <typename T_>
int k_copy_rows_padded_t(const T_* src, T_* dst, int B, int F, int RS, int H, hipStream_t st);
template <typename T_>
int k_col2im_k3s2_t(const T_* dxcol, const T_* pre, T_* out, int B, int F, int Tin, int RS1, int H, hipStream_t st);
int k_conv_wgrad_unrearrange(const float* dwr, float* g, int Co, int Ci, int k, hipStream_t st);

// optim.hip
// out[0] = sum g^2: 1024 per-workgroup partials in `partial` (>= 1024 floats), then one fixed-order pass -- deterministic
int k_sumsq(const float* g, long n, float* out /*[1]*/, float* partial, hipStream_t st, bool add = false /*out[0] += */);
int k_adamw(float* p, const float* g, float* m, float* v, bf16* shadow, long n, const float* gnorm_sq, float max_norm,
            float grad_scale, float lr, float beta1, float beta2, float eps, float wd, int step, hipStream_t st);

// gemm_p8.hip: (256|192|128)x256-tile phase-interleaved GEMM; `params` is gemm_common.h's GemmParams with
// tiles_m / tiles_n counted for bm x 256 tiles
int ssak_gemm_p8_launch(const void* params, int bm, int a_km, int b_km, hipStream_t st);
// transpose.hip: dst[i] [C][R] = src[i] [R][C]^T, n matrices in one launch
int k_transpose_bf16_batched(int n, const bf16* const* src, bf16* const* dst, const int* R, const int* C, hipStream_t st);

// ticket counters of the persistent GEMMs (eight per-XCD counters + the count of finished workgroups), one slot per (device, stream)
int ssak_gemm_ticket_slot(hipStream_t st, int** out);
// gemm_p4.hip: the same tiles with four waves and a hand-scheduled main loop (K-contiguous operands, K % 64 == 0, N % 256 == 0)
bool ssak_gemm_p4_supports(const void* params, int bm, int a_km, int b_km);
int ssak_gemm_p4_launch(const void* params, int bm, int b_km, hipStream_t st);
// B-direct form: params->B = the fragment-ordered copy (k_gemm_fragment_b_batched), params->ext_b its bytes
int ssak_gemm_p8bd_launch(const void* params, int bm, hipStream_t st);
size_t k_gemm_fragment_b_bytes(int N, int K);
int k_gemm_fragment_b_batched(int n, const void* const* B, const long* ldb, const int* N, const int* K, const int* b_km, void* const* out,
                              hipStream_t st);
int ssak_gemm_p8_launch_grouped(const void* params, int n, const void* const* A, const void* const* B, void* const* C, const int* M,
                                const int* N, const long* lda, const long* ldb, const long* ldc, const uint32_t* ext_a,
                                const uint32_t* ext_b, int a_km, int b_km, hipStream_t st);
