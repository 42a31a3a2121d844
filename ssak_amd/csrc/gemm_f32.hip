// fp32 GEMM of the fp32-exact verification mode (ssak_w2v2_config.exact): same descriptor, operand layouts, batch strides and
// fused epilogues as ssak_gemm_bf16 (include/ssak_hip.h), with float operands and results and the products on the fp32 matrix
// pipe (v_mfma_f32_32x32x2_f32).  It stands behind the same nn.Linear / Conv1d / attention products of
// transformers.Wav2Vec2ForCTC (modeling_wav2vec2.py) -- the reference computes them in fp32 (USE_MIXED_PRECISION = False,
// ssak/train/transformers/wav2vec_train.py:191-192) -- so that the engine's sequencing, layouts and row kernels can be checked
// against the reference at fp32 tolerances (1e-4), which the bf16 production path cannot offer.
//
// Not a performance kernel: 64x64 tiles, 16-deep K steps staged through LDS with element-wise bounds checks (any M, N, K and
// leading dimension; overlapping-row "Toeplitz" A operands; K-major operands), one 32x32 MFMA tile per wave.  fp32 MFMA peaks
// at 157 TFLOP/s on MI355X; this kernel reaches a few per cent of it, which is plenty for a B = 2 parity run.
#include "kernels.h"

namespace {

struct F32Params {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* aux_in;
  float* aux_out;
  float* colsum;
  int M, N, K;
  long lda, ldb, ldc;
  int nb2;
  long sa1, sa2, sb1, sb2, sc1, sc2, bias_s2;
  float alpha;
  int epilogue, accumulate;
  uint32_t drop_thresh, drop_stream;
  float drop_scale;
  uint64_t drop_seed;
};

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TM = 64, TN = 64, TK = 16;

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const F32Params p) {
  __shared__ float As[TK][TM + 1];
  __shared__ float Bs[TK][TN + 1];
  const int z = blockIdx.z, z1 = z / p.nb2, z2 = z % p.nb2;
  const float* A = p.A + z1 * p.sa1 + z2 * p.sa2;
  const float* Bm = p.B + z1 * p.sb1 + z2 * p.sb2;
  const long coff = z1 * p.sc1 + z2 * p.sc2;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < p.K; k0 += TK) {
    // stage a TM x TK slab of op(A) and a TN x TK slab of op(B): As[k][m], Bs[k][n]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int r, k;
      if (A_KM) {  // stored [K, M]: consecutive threads walk m
        r = tid & 63;
        k = (tid >> 6) + 4 * i;
      } else {  // stored [M, K]: consecutive threads walk k
        k = tid & 15;
        r = (tid >> 4) + 16 * i;
      }
      const int m = m0 + r, kk = k0 + k;
      float v = 0.f;
      if (m < p.M && kk < p.K) v = A_KM ? A[(long)kk * p.lda + m] : A[(long)m * p.lda + kk];
      As[k][r] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int r, k;
      if (B_KM) {
        r = tid & 63;
        k = (tid >> 6) + 4 * i;
      } else {
        k = tid & 15;
        r = (tid >> 4) + 16 * i;
      }
      const int n = n0 + r, kk = k0 + k;
      float v = 0.f;
      if (n < p.N && kk < p.K) v = B_KM ? Bm[(long)kk * p.ldb + n] : Bm[(long)n * p.ldb + kk];
      Bs[k][r] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; kk += 2) {
      // 32x32x2: lane l supplies A[row l % 32][k = l / 32] and B[k = l / 32][col l % 32]
      const float a = As[kk + (lane >> 5)][wm + (lane & 31)];
      const float b = Bs[kk + (lane >> 5)][wn + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // accumulator layout: acc[i] = C[row = 8 * (i / 4) + 4 * (l / 32) + i % 4][col = l % 32]
  const int n = n0 + wn + (lane & 31);
  float cs = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = m0 + wm + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
    if (m >= p.M || n >= p.N) continue;
    const long o = coff + (long)m * p.ldc + n;
    float v = acc[i] * p.alpha;
    if (p.bias) v += p.bias[z2 * p.bias_s2 + n];
    if (p.epilogue == SSAK_EPI_GELU) {
      if (p.aux_out) p.aux_out[o] = v;
      v = gelu_s<float>(v);
    } else if (p.epilogue == SSAK_EPI_MUL_GELU_GRAD) {
      v *= gelu_grad_s<float>(p.aux_in[o]);
    } else if (p.epilogue == SSAK_EPI_MUL_AUX) {
      v *= p.aux_in[o];
    }
    float gd = 0.f;
    if (p.epilogue == SSAK_EPI_GELU_SAVE_GRAD) {
      gd = gelu_grad_s<float>(v);
      v = gelu_s<float>(v);
    }
    if (p.drop_thresh) {
      const bool keep = keep_bit(p.drop_seed, p.drop_stream, (uint64_t)z * p.M + m, (uint32_t)n, p.drop_thresh);  // (row, column) of the output: common.h
      v = keep ? v * p.drop_scale : 0.f;
      gd = keep ? gd * p.drop_scale : 0.f;
    }
    if (p.epilogue == SSAK_EPI_GELU_SAVE_GRAD && p.aux_out) p.aux_out[o] = gd;
    cs += v;
    p.C[o] = p.accumulate ? p.C[o] + v : v;
  }
  if (p.colsum && n < p.N) atomicAdd(p.colsum + n, cs);
}

uint32_t thresh_of(float pr) { return pr <= 0.f ? 0u : (uint32_t)fminf(65535.f, roundf(pr * 65536.f)); }

}  // namespace

extern "C" int ssak_gemm_f32(const ssak_gemm_desc* d, const void* A, const void* B, void* C, const float* bias, const void* aux_in,
                             void* aux_out, void* stream) {
  SSAK_REQUIRE(d && A && B && C, "gemm_f32: null pointer");
  SSAK_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->nb1 >= 1 && d->nb2 >= 1, "gemm_f32: bad shape %d x %d x %d", d->M, d->N, d->K);
  SSAK_REQUIRE((d->epilogue != SSAK_EPI_MUL_GELU_GRAD && d->epilogue != SSAK_EPI_MUL_AUX) || aux_in, "gemm_f32: this epilogue needs aux_in");
  SSAK_REQUIRE(!d->colsum || (aux_out && d->epilogue != SSAK_EPI_GELU && d->nb1 * d->nb2 == 1), "gemm_f32: colsum needs aux_out [N], no GELU, no batches");
  F32Params p;
  p.A = (const float*)A;
  p.B = (const float*)B;
  p.C = (float*)C;
  p.bias = bias;
  p.aux_in = (const float*)aux_in;
  p.aux_out = d->colsum ? nullptr : (float*)aux_out;
  p.colsum = d->colsum ? (float*)aux_out : nullptr;
  p.M = d->M;
  p.N = d->N;
  p.K = d->K;
  p.lda = d->lda;
  p.ldb = d->ldb;
  p.ldc = d->ldc;
  p.nb2 = d->nb2;
  p.sa1 = d->sa1;
  p.sa2 = d->sa2;
  p.sb1 = d->sb1;
  p.sb2 = d->sb2;
  p.sc1 = d->sc1;
  p.sc2 = d->sc2;
  p.bias_s2 = d->bias_s2;
  p.alpha = d->alpha;
  p.epilogue = d->epilogue;
  p.accumulate = d->accumulate;
  p.drop_thresh = thresh_of(d->drop_p);
  if (d->epilogue == SSAK_EPI_MUL_AUX) p.drop_thresh = 0;  // the float factor carries mask AND scale; drop_p only matters to the bf16 form's codes
  p.drop_scale = p.drop_thresh ? 1.f / (1.f - (float)p.drop_thresh / 65536.f) : 1.f;
  p.drop_stream = d->drop_stream;
  p.drop_seed = d->drop_seed;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(ssak_cdiv(d->N, TN), ssak_cdiv(d->M, TM), d->nb1 * d->nb2);
  SSAK_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm_f32: too many tiles in one grid dimension");
  if (!d->a_kmajor && !d->b_kmajor)
    gemm_f32_kernel<false, false><<<grid, 256, 0, st>>>(p);
  else if (!d->a_kmajor && d->b_kmajor)
    gemm_f32_kernel<false, true><<<grid, 256, 0, st>>>(p);
  else if (d->a_kmajor && !d->b_kmajor)
    gemm_f32_kernel<true, false><<<grid, 256, 0, st>>>(p);
  else
    gemm_f32_kernel<true, true><<<grid, 256, 0, st>>>(p);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
