// Feature-encoder front end and positional-convolution weight plumbing.  gfx950.
//
//  * conv0 (C_in = 1, k = 10, s = 5) + GroupNorm(C groups = per-channel over time) + GELU, written
//    channels-last in bf16 (transformers modeling_wav2vec2.py:302-323).  The fp32 pre-norm activation
//    (65 MB / utterance at 10 s) is never materialised: pass 1 recomputes the 10-tap dot product to get the
//    per-channel statistics (fp64 atomics), pass 2 recomputes it again, normalises, applies GELU and writes
//    bf16 once.  HBM-bound on the 32.7 MB / utterance bf16 store.
//  * Conv1d weight re-layout [Co, Ci, k] -> [Co, k, Ci] bf16: with channels-last activations a strided
//    Conv1d is then a GEMM whose A rows overlap (lda = stride * Ci), no im2col.
//  * Positional conv (:326-368): weight_norm(dim=2) materialisation w = g v / ||v|| into the forward GEMM
//    layout [H][K][H/G] and the flipped/transposed layout of the input-gradient GEMM, its backward to
//    (g, v), and the zero-padded per-group activation packing that turns the grouped conv into batched GEMMs.
#include <cstdlib>

#include "kernels.h"

namespace {

constexpr int KS0 = 10, ST0 = 5;  // conv0 taps / stride
constexpr int FR_STATS = 1024, FR_APPLY = 128;  // frames per workgroup (statistics pass: few, contended fp64 atomics)

template <typename OT, bool APPLY, int FR0>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    OT* __restrict__ out, double* __restrict__ stats,
                                                    const double* __restrict__ sums, int T, int T0, int C) {
  constexpr int NS0 = (FR0 - 1) * ST0 + KS0;
  __shared__ float xs[NS0];
  __shared__ float red[256][9];
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * FR0;
  const int nq = C >> 2;        // channel quads
  const int fl = 256 / nq;      // frame lanes
  const int q = threadIdx.x % nq, fli = threadIdx.x / nq;
  const int nfr = min(FR0, T0 - f0);
  const float* xb = x + (size_t)b * T;
  for (int i = threadIdx.x; i < NS0; i += 256) {
    const int s = f0 * ST0 + i;
    xs[i] = (s < T) ? xb[s] : 0.f;
  }
  float wr[4][KS0];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < KS0; ++k) wr[j][k] = w[(q * 4 + j) * KS0 + k];
  float mu[4], rs[4], ga[4], be[4];
  if (APPLY) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mu[j] = (float)sums[((size_t)b * C + q * 4 + j) * 2];  // (mean, 1 / sqrt(var + eps)) per (utterance, channel)
      rs[j] = (float)sums[((size_t)b * C + q * 4 + j) * 2 + 1];
      ga[j] = gamma[q * 4 + j];
      be[j] = beta[q * 4 + j];
    }
  }
  __syncthreads();
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int f = fli; f < nfr; f += fl) {
    float xv[KS0];
#pragma unroll
    for (int k = 0; k < KS0; ++k) xv[k] = xs[f * ST0 + k];
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < KS0; ++k) a = fmaf(wr[j][k], xv[k], a);
      v[j] = a;
    }
    if (APPLY) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = gelu_s<OT>((v[j] - mu[j]) * rs[j] * ga[j] + be[j]);
      st4<OT>(out + ((size_t)b * T0 + f0 + f) * C + q * 4, f_to_chunk4<OT>(o));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1[j] += v[j];
        s2[j] = fmaf(v[j], v[j], s2[j]);
      }
    }
  }
  if (!APPLY) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[threadIdx.x][j] = s1[j];
      red[threadIdx.x][4 + j] = s2[j];
    }
    __syncthreads();
    if (fli == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double a = 0.0, c = 0.0;
        for (int l = 0; l < fl; ++l) {
          a += red[l * nq + q][j];
          c += red[l * nq + q][4 + j];
        }
        // per-workgroup partial; conv0_stats_finalize_kernel sums them in a fixed order
        double* dst = stats + (((size_t)b * gridDim.x + blockIdx.x) * C + q * 4 + j) * 2;
        dst[0] = a;
        dst[1] = c;
      }
    }
  }
}

// conv0 (+ bias) only, bf16 channels-last: the layer-norm feature encoder (XLSR) normalises over channels per frame
// afterwards (LayerNorm + GELU row kernel), so no time statistics are needed here.
template <typename OT>
__global__ __launch_bounds__(256) void conv0_bias_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, OT* __restrict__ out, int T,
                                                         int T0, int C) {
  constexpr int FR0 = FR_APPLY, NS0 = (FR0 - 1) * ST0 + KS0;
  __shared__ float xs[NS0];
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * FR0;
  const int nq = C >> 2, fl = 256 / nq;
  const int q = threadIdx.x % nq, fli = threadIdx.x / nq;
  const int nfr = min(FR0, T0 - f0);
  const float* xb = x + (size_t)b * T;
  for (int i = threadIdx.x; i < NS0; i += 256) {
    const int s = f0 * ST0 + i;
    xs[i] = (s < T) ? xb[s] : 0.f;
  }
  float wr[4][KS0], bb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bb[j] = bias ? bias[q * 4 + j] : 0.f;
#pragma unroll
    for (int k = 0; k < KS0; ++k) wr[j][k] = w[(q * 4 + j) * KS0 + k];
  }
  __syncthreads();
  for (int f = fli; f < nfr; f += fl) {
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = bb[j];
#pragma unroll
      for (int k = 0; k < KS0; ++k) a = fmaf(wr[j][k], xs[f * ST0 + k], a);
      o[j] = a;
    }
    st4<OT>(out + ((size_t)b * T0 + f0 + f) * C + q * 4, f_to_chunk4<OT>(o));
  }
}

// conv0 apply pass on the matrix cores (bf16 output, C = 512).  The VALU form above spends ~25 instructions per output
// element -- ten fmas for the taps, the affine, the GELU -- and with 524 M elements per step of 32 utterances that is what
// bounds it (tools/probes/valu_rate.hip: an fma is 4 cycles per wave; 344 us measured, 1.07 GB of stores would take ~200 us):
// it is arithmetic-bound, not store-bound.  Here the ten-tap dot product is ONE pair of 16x16x32 MFMAs per 16 channels x 16
// frames with fp32-grade accuracy from a three-term bf16 split, x = xh + xl, w = wh + wl (hi / lo bf16 parts):
//   MFMA 1: K = [wh (10, padded to 16) | wh (16)] x [xh (16) | xl (16)]  = wh xh + wh xl
//   MFMA 2: K = [wl (16) | 0]                     x [xh (16) | 0]        = wl xh          (wl xl ~ 2^-18 relative: dropped)
// so the VALU only runs y = v a_c + b_c (the GroupNorm folded into one fma) and the GELU, in packed fp32.
//   workgroup = 128 frames of one utterance, 4 waves x 128 channels; the frames' [xh | xl] rows are built once per workgroup
//   in LDS (64 B per frame), the weight fragments once per wave in registers.  Orientation out^T[channel][frame] = W X^T: a
//   lane then holds four consecutive channels of one frame, which go through a wave-private, XOR-swizzled LDS tile
//   ([16 frames][128 channels], 8-byte writes) and leave as 16-byte stores of whole 256-byte row segments.
#ifndef C0M_HALF_SWAP
#define C0M_HALF_SWAP 1  // (0: the round-2 store tile, for A/B builds)
#endif
constexpr int C0M_FR = 128;
#ifndef C0M_SUB
#define C0M_SUB 4  // blocks of C0M_FR frames per workgroup
#endif
constexpr int C0M_NS = (C0M_FR - 1) * ST0 + KS0;  // 645 input samples
typedef __attribute__((ext_vector_type(2))) float c0_f32x2;

__global__ __launch_bounds__(256) void conv0_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         bf16* __restrict__ out, const double* __restrict__ sums, int T, int T0) {
  constexpr int C = 512;
  __shared__ float xs[C0M_NS + 3];
  __shared__ __attribute__((aligned(16))) float ab[2][C];                 // y = v * ab[0][c] + ab[1][c]
  __shared__ __attribute__((aligned(16))) char ximg[C0M_FR * 64];        // per frame: xh[0..9], 0 x 6 | xl[0..9], 0 x 6 (bf16)
  __shared__ __attribute__((aligned(16))) char ostage[4][16 * 256];      // per wave: [16 frames][128 channels] bf16, 16-byte chunk ch of row r at ch ^ r
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, lc = lane & 15;
  const int b = blockIdx.y;
  const float* xb = x + (size_t)b * T;
  // A workgroup walks C0M_SUB consecutive blocks of 128 frames: the weight fragments (64 scattered loads + a hi / lo split per
  // lane) and the GroupNorm table are set up once per C0M_SUB x 128 frames instead of once per 128 -- the set-up was a third of a
  // workgroup's life (20 us for ~7 us of main-loop issue).  The samples of the next block are fetched while this one is computed.
  auto load_xs = [&](int f0) {
    for (int i = threadIdx.x; i < C0M_NS; i += 256) {
      const int sidx = f0 * ST0 + i;
      xs[i] = (sidx < T) ? xb[sidx] : 0.f;
    }
  };
  load_xs(blockIdx.x * C0M_SUB * C0M_FR);
  for (int c = threadIdx.x; c < C; c += 256) {
    const float mu = (float)sums[((size_t)b * C + c) * 2], rs = (float)sums[((size_t)b * C + c) * 2 + 1];
    const float a = rs * gamma[c];
    ab[0][c] = a;
    ab[1][c] = beta[c] - mu * a;
  }
  // weight fragments (A operands: row = channel 128 wave + 16 ct + lc, k = 8 g + j):
  //   A1 = [wh | wh]: g even -> taps 0..7, g odd -> taps 8, 9, then zeros;   A2 = [wl | 0]: the same for g < 2, zeros above
  bf16x8 wa1[8], wa2[8];
#pragma unroll
  for (int ct = 0; ct < 8; ++ct) {
    const float* wc = w + (size_t)(128 * wave + 16 * ct + lc) * KS0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = (g & 1) * 8 + j;
      const float v = k < KS0 ? wc[k] : 0.f;
      const bf16 hi = (bf16)v;
      const bf16 lo = (bf16)(v - (float)hi);
      wa1[ct][j] = hi;
      wa2[ct][j] = g < 2 ? lo : (bf16)0.f;
    }
  }
  __syncthreads();
  char* const ost = ostage[wave];
  const bf16x8 zero8 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
  for (int sub = 0; sub < C0M_SUB; ++sub) {
  const int f0 = (blockIdx.x * C0M_SUB + sub) * C0M_FR;
  if (f0 >= T0) break;  // uniform
  {  // [xh | xl] rows: thread = (frame, half)
    const int f = threadIdx.x >> 1, half = threadIdx.x & 1;
    bf16x8 r0, r1;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float v = k < KS0 ? xs[f * ST0 + k] : 0.f;
      const bf16 hi = (bf16)v;
      const bf16 e = half ? (bf16)(v - (float)hi) : hi;
      if (k < 8)
        r0[k] = e;
      else
        r1[k - 8] = e;
    }
    *reinterpret_cast<bf16x8*>(ximg + f * 64 + half * 32) = r0;
    *reinterpret_cast<bf16x8*>(ximg + f * 64 + half * 32 + 16) = r1;
  }
  __syncthreads();  // the frame rows are built: xs is free for the next block's samples, which land under this block's arithmetic
  // (into registers now, into xs behind the main loop: a store to LDS here would wait for the loads on the spot)
  constexpr int C0M_XPT = (C0M_NS + 255) / 256;
  float nx[C0M_XPT];
  const bool more = sub + 1 < C0M_SUB && f0 + C0M_FR < T0;
  if (more) {
#pragma unroll
    for (int u = 0; u < C0M_XPT; ++u) {
      const int i = threadIdx.x + 256 * u, sidx = (f0 + C0M_FR) * ST0 + i;
      nx[u] = (i < C0M_NS && sidx < T) ? xb[sidx] : 0.f;
    }
  }
  for (int ft = 0; ft < C0M_FR / 16; ++ft) {
    if (f0 + 16 * ft >= T0) break;  // uniform
    // B operands: column = frame 16 ft + lc, k = 8 g + j of the frame's 64-byte row
    const bf16x8 xb1 = *reinterpret_cast<const bf16x8*>(ximg + (16 * ft + lc) * 64 + g * 16);
    const bf16x8 xb2 = g < 2 ? xb1 : zero8;
    // Stage-major over the eight channel tiles: as one chain per tile (affine -> polynomial -> exp -> rcp -> store) the
    // compiler emitted the eight GELU chains one after the other, each waiting on its own transcendental latencies.
    c0_f32x2 y[16];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa1[ct], xb1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa2[ct], xb2, acc, 0, 0, 0);
      // acc[r] = conv(channel 128 wave + 16 ct + 4 g + r, frame 16 ft + lc)
      const int c0 = 128 * wave + 16 * ct + 4 * g;
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(&ab[0][c0]);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(&ab[1][c0]);
      y[2 * ct] = (c0_f32x2){acc[0], acc[1]} * (c0_f32x2){a4[0], a4[1]} + (c0_f32x2){b4[0], b4[1]};
      y[2 * ct + 1] = (c0_f32x2){acc[2], acc[3]} * (c0_f32x2){a4[2], a4[3]} + (c0_f32x2){b4[2], b4[3]};
    }
    {  // GELU = y * Phi(y), Phi = 1 / (1 + exp2(q(y))) (common.h: phi2), every stage over all sixteen pairs
      c0_f32x2 xc[16], q[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) xc[i] = (c0_f32x2){__builtin_amdgcn_fmed3f(y[i][0], -6.f, 6.f), __builtin_amdgcn_fmed3f(y[i][1], -6.f, 6.f)};
      c0_f32x2 sq[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) sq[i] = xc[i] * xc[i];
#pragma unroll
      for (int i = 0; i < 16; ++i) q[i] = __builtin_elementwise_fma(sq[i], (c0_f32x2){SSAK_PHI_C3, SSAK_PHI_C3}, (c0_f32x2){SSAK_PHI_C2, SSAK_PHI_C2});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) q[i] = __builtin_elementwise_fma(q[i], sq[i], (c0_f32x2){SSAK_PHI_C1, SSAK_PHI_C1});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) q[i] = __builtin_elementwise_fma(q[i], sq[i], (c0_f32x2){SSAK_PHI_C0, SSAK_PHI_C0});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) q[i] = q[i] * xc[i];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) q[i] = (c0_f32x2){__builtin_amdgcn_exp2f(q[i][0]), __builtin_amdgcn_exp2f(q[i][1])} + (c0_f32x2){1.f, 1.f};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) q[i] = (c0_f32x2){__builtin_amdgcn_rcpf(q[i][0]), __builtin_amdgcn_rcpf(q[i][1])};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
        const c0_f32x2 o0 = y[2 * ct] * q[2 * ct], o1 = y[2 * ct + 1] * q[2 * ct + 1];
        const bf16x4 o = {(bf16)o0[0], (bf16)o0[1], (bf16)o1[0], (bf16)o1[1]};
        // row lc of the wave's tile, channels 16 ct + 4 g .. + 3: 16-byte chunk 2 ct + (g >> 1), half (g & 1).  The sixteen
        // lanes of a store group (one g, lc = 0..15) write 8 bytes each = all 32 banks exactly when their 8-byte slots differ:
        // the chunk swizzle alone sends rows lc and lc + 8 to the same slot (2-way conflict on every store: rocprofv3 counted
        // SQ_LDS_BANK_CONFLICT = 24 % of the LDS cycles), so rows 8..15 also swap the two halves of a chunk (undone at the read)
        *reinterpret_cast<bf16x4*>(ost + lc * 256 + (((2 * ct + (g >> 1)) ^ lc) << 4) + (((g & 1) ^ (C0M_HALF_SWAP ? lc >> 3 : 0)) << 3)) = o;
      }
    }
    // the tile leaves as whole 256-byte row segments: lane = (row 4 pass + g, chunk lc)
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = 4 * pass + g;
      f32x4 v = *reinterpret_cast<const f32x4*>(ost + r * 256 + ((lc ^ r) << 4));
      if (C0M_HALF_SWAP && pass >= 2) v = (f32x4){v[2], v[3], v[0], v[1]};  // rows 8..15 were stored with their chunk halves swapped
      const int frame = f0 + 16 * ft + r;
      if (frame < T0) *reinterpret_cast<f32x4*>(out + ((size_t)b * T0 + frame) * C + 128 * wave + 8 * lc) = v;
    }
  }
  if (more) {
#pragma unroll
    for (int u = 0; u < C0M_XPT; ++u)
      if (threadIdx.x + 256 * u < C0M_NS) xs[threadIdx.x + 256 * u] = nx[u];
  }
  __syncthreads();  // every wave is done with this block's frame rows (and sees the next block's samples)
  }
}

// GroupNorm statistics of the conv0 output WITHOUT evaluating the convolution: y[c,t] = sum_k w[c,k] x[5t+k], hence
//   sum_t y[c,t]   = sum_k w[c,k] S[k],              S[k]    = sum_t x[5t+k]
//   sum_t y[c,t]^2 = sum_{k,k'} w[c,k] w[c,k'] R[k,k'],  R[k,k'] = sum_t x[5t+k] x[5t+k']
// i.e. 65 moments of the input per utterance (fp64) instead of a 512-channel convolution pass: the statistics pass drops
// from 144 us to a few us.  Output sums[b][c] = (mean_t y, 1 / sqrt(var_t y + eps)), fixed summation order.
// Two more moments, sum x and sum x^2 over ALL samples of the utterance (slots NMOM - 2, NMOM - 1), fold the feature extractor's
// zero-mean / unit-variance normalisation (a1: x_n = (x - mu) / sqrt(var + 1e-7), transformers feature_extraction_wav2vec2.py:
// 78-97) into this GroupNorm for full-length utterances: conv0 is linear and bias-free, so conv(x_n) = (conv(x) - mu W_c) / sigma,
// and GroupNorm over time is invariant under a per-channel shift and scale except for its epsilon:
//     GN(conv(x_n)) = (y - mean_t y) / sqrt(var_t y + 1e-5 sigma^2),      y = conv(x) on the RAW waveform.
// conv0_channel_stats_kernel uses 1e-5 sigma^2 as the epsilon of rstd = 1 / sqrt(var_t y + eps) and the apply pass runs on raw
// samples unchanged: the train step needs no normalisation pass at all (it was two launches and two passes over the waveform).
constexpr int NMOM = KS0 + KS0 * (KS0 + 1) / 2 + 2;
__global__ __launch_bounds__(256) void conv0_moments_kernel(const float* __restrict__ x, int T, int T0, double* __restrict__ partial) {
  constexpr int FR0 = FR_STATS, NS0 = (FR0 - 1) * ST0 + KS0;
  __shared__ float xs[NS0];
  __shared__ double red[4][NMOM];
  const int b = blockIdx.y, f0 = blockIdx.x * FR0;
  const int nfr = min(FR0, T0 - f0);
  const float* xb = x + (size_t)b * T;
  for (int i = threadIdx.x; i < NS0; i += 256) {
    const int s = f0 * ST0 + i;
    xs[i] = (s < T) ? xb[s] : 0.f;
  }
  __syncthreads();
  double m[NMOM];
#pragma unroll
  for (int i = 0; i < NMOM; ++i) m[i] = 0.0;
  {
    // every sample once: this workgroup's frames start at samples [5 f0, 5 (f0 + nfr)); the last workgroup takes the tail up to T
    // (the last few samples of an utterance lie behind its last window: outside xs, straight from memory)
    const int own = (blockIdx.x + 1 == gridDim.x) ? T - f0 * ST0 : nfr * ST0;
    for (int i = threadIdx.x; i < own; i += 256) {
      const double v = (double)(i < NS0 ? xs[i] : xb[f0 * ST0 + i]);
      m[NMOM - 2] += v;
      m[NMOM - 1] += v * v;
    }
  }
  for (int f = threadIdx.x; f < nfr; f += 256) {
    double xv[KS0];
#pragma unroll
    for (int k = 0; k < KS0; ++k) xv[k] = (double)xs[f * ST0 + k];
    int idx = KS0;
#pragma unroll
    for (int k = 0; k < KS0; ++k) {
      m[k] += xv[k];
#pragma unroll
      for (int k2 = k; k2 < KS0; ++k2) m[idx++] += xv[k] * xv[k2];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Sum every moment over the wave's 64 lanes.  One butterfly per moment is 67 x 6 exchanges of a double (two ds_bpermute
  // each: 804 per wave, and that, not the arithmetic, was the kernel -- 64 us per launch).  Instead the lanes split the LIST at
  // every step: of c live entries a lane keeps the lower or the upper half (by the step's lane bit), sends the other half to
  // its partner and adds what the partner sent -- 34 + 17 + 9 + 5 + 3 + 2 = 70 exchanges, after which lane L holds two complete
  // sums, those of moments j + 2 b1 + 3 b2 + 5 b4 + 9 b8 + 17 b16 + 34 b32 (b = L's bits), j = 0, 1.
  {
    constexpr int CNT[7] = {NMOM, 34, 17, 9, 5, 3, 2};
    static_assert(NMOM == 67, "the halving schedule below is written for 67 moments");
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int off = 32 >> s, c = CNT[s], h = CNT[s + 1];
      const bool up = (lane & off) != 0;
#pragma unroll
      for (int i = 0; i < h; ++i) {
        const double lo = m[i], hi = (i + h < c) ? m[i + h] : 0.0;
        const double keep = up ? hi : lo, send = up ? lo : hi;
        m[i] = keep + __shfl_xor(send, off);
      }
    }
    // the moments this lane's m[0] and m[1] belong to; valid if every level's index is inside its list
    bool ok0 = true, ok1 = true;
    int i0 = 0, i1 = 1;
#pragma unroll
    for (int s = 5; s >= 0; --s) {
      const int add = (lane & (32 >> s)) ? CNT[s + 1] : 0;
      i0 += add;
      i1 += add;
      ok0 = ok0 && i0 < CNT[s];
      ok1 = ok1 && i1 < CNT[s];
    }
    if (ok0) red[wave][i0] = m[0];
    if (ok1) red[wave][i1] = m[1];
  }
  __syncthreads();
  if (threadIdx.x < NMOM)
    partial[((size_t)b * gridDim.x + blockIdx.x) * NMOM + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void conv0_channel_stats_kernel(const double* __restrict__ partial, int nblk, const float* __restrict__ w,
                                                                  int C, double* __restrict__ sums, int T, int T0, int fold_norm) {
  __shared__ double mom[NMOM];
  const int b = blockIdx.y;
  if (threadIdx.x < NMOM) {
    double s = 0.0;
    for (int k = 0; k < nblk; ++k) s += partial[((size_t)b * nblk + k) * NMOM + threadIdx.x];
    mom[threadIdx.x] = s;
  }
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double wk[KS0];
#pragma unroll
  for (int k = 0; k < KS0; ++k) wk[k] = (double)w[c * KS0 + k];
  double s1 = 0.0, s2 = 0.0;
  int idx = KS0;
#pragma unroll
  for (int k = 0; k < KS0; ++k) {
    s1 += wk[k] * mom[k];
#pragma unroll
    for (int k2 = k; k2 < KS0; ++k2) s2 += (k2 == k ? 1.0 : 2.0) * wk[k] * wk[k2] * mom[idx++];
  }
  double eps = 1e-5;
  if (fold_norm) {  // raw waveform in: the normalisation's sigma^2 rescales GroupNorm's epsilon (see NMOM above)
    const double mu = mom[NMOM - 2] / T, sig2 = fmax(mom[NMOM - 1] / T - mu * mu, 0.0) + 1e-7;
    eps *= sig2;
  }
  // the epsilon stays OUTSIDE the clamp: a near-dead channel of quiet audio has var_y far below 1e-5 and 1e-5 sigma^2 is then
  // the whole denominator (adding it to sum y^2 instead lost it to the clamp at 0)
  const double m = s1 / T0;
  sums[((size_t)b * C + c) * 2] = m;
  sums[((size_t)b * C + c) * 2 + 1] = 1.0 / sqrt(fmax(s2 / T0 - m * m, 0.0) + eps);
}

__global__ void conv0_stats_finalize_kernel(const double* __restrict__ partial, int nblk, int C, int T0, double* __restrict__ sums) {
  // grid (ceil(C/256), B): sums[b][c] = (mean, 1 / sqrt(var + 1e-5)) from the nblk workgroup partials (sum y, sum y^2), fixed order
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int b = blockIdx.y;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < nblk; ++k) {
    s1 += partial[(((size_t)b * nblk + k) * C + c) * 2];
    s2 += partial[(((size_t)b * nblk + k) * C + c) * 2 + 1];
  }
  const double m = s1 / T0;
  sums[((size_t)b * C + c) * 2] = m;
  sums[((size_t)b * C + c) * 2 + 1] = 1.0 / sqrt(fmax(s2 / T0 - m * m, 0.0) + 1e-5);
}

// ---- feature-encoder backward (--no_freeze, ssak/train/transformers/wav2vec_train.py:326-327 off) -------------------
// input gradient of a channels-last Conv1d(k, s, no padding) from the column form dxcol [B, Tout, k, C]:
//   dx[u] = sum over taps kk with (u - kk) % s == 0 and t = (u - kk) / s in [0, Tout) of dxcol[t][kk]
template <typename T_>
__global__ void col2im_kernel(const T_* __restrict__ dxcol, T_* __restrict__ dx, int Tin, int Tout, int C, int k, int s,
                              long n8) {
  const int hc = C >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long row = i / hc;
    const int c = (int)(i % hc);
    const long b = row / Tin;
    const int u = (int)(row % Tin);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < k; ++kk) {
      const int d = u - kk;
      if (d < 0 || d % s) continue;
      const int t = d / s;
      if (t >= Tout) continue;
      float q[8];
      chunk_to_f(ld8<T_>(dxcol + ((((long)b * Tout + t) * k + kk) * hc + c) * 8), q);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += q[j];
    }
    st8<T_>(dx + 8 * i, f_to_chunk8<T_>(acc));
  }
}

// out[e] = sum_b slabs[b][e]   (per-utterance weight-gradient slabs -> one gradient, fixed order)
__global__ void sum_slabs_kernel(const float* __restrict__ slabs, int nb, long n, float* __restrict__ out) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += slabs[(long)b * n + e];
    out[e] = s;
  }
}

// conv0 + GroupNorm + GELU backward, recomputing the 10-tap dot products like the forward does.
//   PASS 0: per (b, c) partial sums of g = dy * gelu'(y) and g * xhat over this workgroup's frames
//   PASS 1: dv = gamma * rstd * (g - mean_t(g) - xhat * mean_t(g xhat)); partial dW[c][tap] = sum_t dv * x[5t + tap]
template <int PASS, typename DT>
__global__ __launch_bounds__(256) void conv0_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const DT* __restrict__ dy, const double* __restrict__ sums,
                                                        const double* __restrict__ gsums, float* __restrict__ partial, int T,
                                                        int T0, int C) {
  constexpr int FR0 = FR_STATS, NS0 = (FR0 - 1) * ST0 + KS0;
  __shared__ float xs[NS0];
  extern __shared__ float red[];  // [256][NACC + 1]
  constexpr int NACC = PASS == 0 ? 8 : 4 * KS0;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * FR0;
  const int nq = C >> 2, fl = 256 / nq;
  const int q = threadIdx.x % nq, fli = threadIdx.x / nq;
  const int nfr = min(FR0, T0 - f0);
  const float* xb = x + (size_t)b * T;
  for (int i = threadIdx.x; i < NS0; i += 256) {
    const int s = f0 * ST0 + i;
    xs[i] = (s < T) ? xb[s] : 0.f;
  }
  float wr[4][KS0], mu[4], rs[4], ga[4], be[4], m1[4], m2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = q * 4 + j;
#pragma unroll
    for (int k = 0; k < KS0; ++k) wr[j][k] = w[c * KS0 + k];
    mu[j] = (float)sums[((size_t)b * C + c) * 2];
    rs[j] = (float)sums[((size_t)b * C + c) * 2 + 1];
    ga[j] = gamma[c];
    be[j] = beta[c];
    m1[j] = PASS == 1 ? (float)(gsums[((size_t)b * C + c) * 2] / T0) : 0.f;
    m2[j] = PASS == 1 ? (float)(gsums[((size_t)b * C + c) * 2 + 1] / T0) : 0.f;
  }
  __syncthreads();
  float acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) acc[a] = 0.f;
  for (int f = fli; f < nfr; f += fl) {
    float xv[KS0];
#pragma unroll
    for (int k = 0; k < KS0; ++k) xv[k] = xs[f * ST0 + k];
    float d4[4];
    chunk_to_f(ld4<DT>(dy + ((size_t)b * T0 + f0 + f) * C + q * 4), d4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < KS0; ++k) v = fmaf(wr[j][k], xv[k], v);
      const float xh = (v - mu[j]) * rs[j];
      const float g = d4[j] * gelu_grad_s<DT>(xh * ga[j] + be[j]);
      if (PASS == 0) {
        acc[j] += g;
        acc[4 + j] = fmaf(g, xh, acc[4 + j]);
      } else {
        const float dv = ga[j] * rs[j] * (g - m1[j] - xh * m2[j]);
#pragma unroll
        for (int k = 0; k < KS0; ++k) acc[j * KS0 + k] = fmaf(dv, xv[k], acc[j * KS0 + k]);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < NACC; ++a) red[threadIdx.x * (NACC + 1) + a] = acc[a];
  __syncthreads();
  if (fli == 0) {
    float* dst = partial + ((size_t)b * gridDim.x + blockIdx.x) * (size_t)(C / 4) * NACC + (size_t)q * NACC;
    for (int a = 0; a < NACC; ++a) {
      float s = 0.f;
      for (int l = 0; l < fl; ++l) s += red[(l * nq + q) * (NACC + 1) + a];
      dst[a] = s;
    }
  }
}

// gsums[b][c][0|1] = sum over workgroups of the PASS-0 partials (layout [b][blk][q][8]: g for 4 channels, then g*xhat)
__global__ void conv0_bwd_gsums_kernel(const float* __restrict__ partial, int nblk, int C, double* __restrict__ gsums) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;  // (c, which)
  if (e >= 2 * C) return;
  const int b = blockIdx.y, c = e >> 1, which = e & 1;
  const int q = c >> 2, j = c & 3;
  double s = 0.0;
  for (int k = 0; k < nblk; ++k) s += partial[(((size_t)b * nblk + k) * (C / 4) + q) * 8 + which * 4 + j];
  gsums[((size_t)b * C + c) * 2 + which] = s;
}
// dgamma[c] += sum_b gsums[b][c][1], dbeta[c] += sum_b gsums[b][c][0]
__global__ void conv0_bwd_affine_kernel(const double* __restrict__ gsums, int B, int C, float* __restrict__ dgamma,
                                        float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, bb = 0.0;
  for (int b = 0; b < B; ++b) {
    bb += gsums[((size_t)b * C + c) * 2];
    a += gsums[((size_t)b * C + c) * 2 + 1];
  }
  dgamma[c] += (float)a;
  dbeta[c] += (float)bb;
}
// dW0[c][tap] += sum over (b, workgroup) of the PASS-1 partials (layout [slab][q][4*KS0])
__global__ void conv0_bwd_dw_kernel(const float* __restrict__ partial, int nslab, int C, float* __restrict__ dw) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;  // c * KS0 + tap
  if (e >= C * KS0) return;
  const int c = e / KS0, tap = e % KS0;
  const int q = c >> 2, j = c & 3;
  float s = 0.f;
  for (int k = 0; k < nslab; ++k) s += partial[((size_t)k * (C / 4) + q) * (4 * KS0) + j * KS0 + tap];
  dw[e] += s;
}

template <typename OT>
__global__ void conv_w_rearrange_kernel(const float* __restrict__ w, OT* __restrict__ out, int Co, int Ci, int k) {
  const long n = (long)Co * Ci * k;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int kk = (int)(e % k);
    const int ci = (int)((e / k) % Ci);
    const int co = (int)(e / ((long)k * Ci));
    out[((long)co * k + kk) * Ci + ci] = (OT)w[e];
  }
}

// Per-tap sums over the [H][cg] rows of v[o][c][k] (k fastest): ||v_k||^2, or the dot product with the weight gradient
// dwf[group][k][c][n] (o = group * cg + n).  One workgroup per (group, c) pair, a thread per tap k: the reads of v are
// coalesced over k, each thread walks the cg contiguous n of its dwf row; partial[(group * cg + c)][k], then a fixed-order
// reduction.  (The first version used 256 two-wave workgroups striding over all rows with an uncoalesced dwf gather and a
// single-workgroup serial finalize: 89 + 60 us for 19 MB.)
__global__ void posconv_colnorm_kernel(const float* __restrict__ v, const float* __restrict__ dwf, int G, int K, int cg,
                                       float* __restrict__ partial) {
  const int k = threadIdx.x;
  if (k >= K) return;
  const int g = blockIdx.x / cg, c = blockIdx.x % cg;
  float s = 0.f;
  const float* vp = v + ((long)(g * cg) * cg + c) * K + k;  // + n * cg * K
  if (dwf) {
    const float* dp = dwf + (((long)g * K + k) * cg + c) * cg;  // + n
    // (unrolled: the loads of eight rows in flight per thread; one at a time this kernel ran at HBM latency, 34 us for 19 MB)
#pragma unroll 8
    for (int n = 0; n < cg; ++n) s = fmaf(vp[(long)n * cg * K], dp[n], s);
  } else {
#pragma unroll 8
    for (int n = 0; n < cg; ++n) {
      const float a = vp[(long)n * cg * K];
      s = fmaf(a, a, s);
    }
  }
  partial[(size_t)blockIdx.x * K + k] = s;
}
// out[k] = sum over nblk partial rows, fixed order; workgroup = 64 taps x 16 row groups
__global__ __launch_bounds__(1024) void posconv_colnorm_finalize_kernel(const float* __restrict__ partial, int nblk, int K, float* __restrict__ out) {
  __shared__ float red[16][65];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + cx;
  float s = 0.f;
  if (k < K) {
#pragma unroll 8
    for (int b = ry; b < nblk; b += 16) s += partial[(size_t)b * K + k];  // (eight loads in flight: two workgroups do all of it)
  }
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && k < K) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cx];
    out[k] = t;
  }
}

// w = g[k] v / ||v_k|| in the two bf16 layouts the GEMMs read: wf[o][k][c] (forward / dX operand) and
// wb[group][c][K-1-k][n] (o = group * cg + n).  Both are (c, k) <-> (k, c|n) transposes of v[o][c][k]: a workgroup stages
// one [cg][K] slab through LDS -- slab o for wf (blocks [0, H)), the slab of (group, c) over its cg output channels for wb
// (blocks [H, 2H)) -- so that global reads and writes are both contiguous (the element-wise version scattered 2-byte
// stores: 48 us for 38 MB).
template <typename OT>
__global__ __launch_bounds__(256) void posconv_materialize_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                                  const float* __restrict__ nsq, OT* __restrict__ wf,
                                                                  OT* __restrict__ wb, int H, int cg, int K) {
  extern __shared__ float slab[];  // [cg][K + 1]
  const int KP = K + 1;
  const bool back = (int)blockIdx.x >= H;
  const int id = back ? blockIdx.x - H : blockIdx.x;
  if (!back) {
    const float* src = v + (long)id * cg * K;  // v[o = id][c][k]
    for (int j = threadIdx.x; j < cg * K; j += 256) {
      const int c = j / K, k = j % K;
      slab[c * KP + k] = g[k] * src[j] * rsqrtf(nsq[k]);
    }
    __syncthreads();
    OT* dst = wf + (long)id * K * cg;  // wf[o][k][c]
    for (int j = threadIdx.x; j < K * cg; j += 256) dst[j] = (OT)slab[(j % cg) * KP + j / cg];
  } else {
    const int grp = id / cg, c = id % cg;
    for (int j = threadIdx.x; j < cg * K; j += 256) {
      const int nn = j / K, k = j % K;
      slab[nn * KP + k] = g[k] * v[(((long)grp * cg + nn) * cg + c) * K + k] * rsqrtf(nsq[k]);
    }
    __syncthreads();
    OT* dst = wb + (long)id * K * cg;  // wb[group][c][kk][n], kk = K - 1 - k
    for (int j = threadIdx.x; j < K * cg; j += 256) dst[j] = (OT)slab[(j % cg) * KP + (K - 1 - j / cg)];
  }
}

// weight-norm backward: dv[o][c][k] += g[k] / ||v_k|| (d - v dot[k] / ||v_k||^2), dg[k] += dot[k] / ||v_k||, with
// d = dwf[group][k][c][n].  One workgroup per (group, c): the dwf rows (contiguous over n) go through LDS and leave
// as rows contiguous over k.
__global__ __launch_bounds__(256) void posconv_wbwd_kernel(const float* __restrict__ dwf, const float* __restrict__ g,
                                                           const float* __restrict__ v, const float* __restrict__ nsq,
                                                           const float* __restrict__ dot, float* __restrict__ dg,
                                                           float* __restrict__ dv, int H, int cg, int K) {
  extern __shared__ float slab[];  // [K][cg + 1]
  const int CP = cg + 1;
  const int grp = blockIdx.x / cg, c = blockIdx.x % cg;
  for (int j = threadIdx.x; j < K * cg; j += 256) {
    const int k = j / cg, nn = j % cg;
    slab[k * CP + nn] = dwf[(((long)grp * K + k) * cg + c) * cg + nn];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < cg * K; j += 256) {
    const int nn = j / K, k = j % K;
    const long e = (((long)grp * cg + nn) * cg + c) * K + k;
    const float inv = rsqrtf(nsq[k]);
    dv[e] += g[k] * inv * (slab[k * CP + nn] - v[e] * dot[k] * inv * inv);
  }
  if (blockIdx.x == 0)
    for (int k = threadIdx.x; k < K; k += 256) dg[k] += dot[k] * rsqrtf(nsq[k]);
}

template <typename T>
__global__ void posconv_pack_kernel(const T* __restrict__ h, T* __restrict__ pg, int B, int F, int H, int G,
                                    int lead, int RS, long rows_total) {
  const int cg = H / G, cpr = cg >> 3;
  const long n = rows_total * G * cpr;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(e % cpr);
    const long r = (e / cpr) % rows_total;
    const int g = (int)(e / ((long)cpr * rows_total));
    const long rr = r - lead;
    Chunk8<T> val = {};
    if (rr >= 0) {
      const int b = (int)(rr / RS), t = (int)(rr % RS);
      if (b < B && t < F) val = ld8<T>(h + ((long)b * F + t) * H + g * cg + cc * 8);
    }
    st8<T>(pg + ((long)g * rows_total + r) * cg + cc * 8, val);
  }
}

}  // namespace

namespace {

// conv0 weight gradient for the layer-norm feature encoder (the gradient w.r.t. the conv output is materialised there):
// dw[c][k] = sum_{b,t} d[b,t,c] * x[b, stride t + k].  Workgroup = a contiguous range of frames of one utterance; thread =
// two channels, 2 * ksize accumulators in registers; the frame's taps are read once per workgroup into LDS.
constexpr int C0W_BLOCKS_PER_UTT = 32;
template <typename DT>
__global__ __launch_bounds__(256) void conv0_wgrad_kernel(const DT* __restrict__ d, const float* __restrict__ x, float* __restrict__ partial,
                                                          int T, int T0, int C, int ksize, int stride) {
  __shared__ float xs[64 * 5 + 16];  // the samples of 64 frames (stride 5, kernel 10)
  const int b = blockIdx.y, blk = blockIdx.x;
  const int per = (T0 + C0W_BLOCKS_PER_UTT - 1) / C0W_BLOCKS_PER_UTT;
  const int t_lo = blk * per, t_hi = min(T0, t_lo + per);
  float acc[2][10];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[u][k] = 0.f;
  const float* xb = x + (size_t)b * T;
  for (int t0 = t_lo; t0 < t_hi; t0 += 64) {
    const int nt = min(64, t_hi - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < nt * stride + ksize; i += 256) {
      const long xi = (long)t0 * stride + i;
      xs[i] = xi < T ? xb[xi] : 0.f;
    }
    __syncthreads();
    for (int c2 = threadIdx.x; c2 < C / 2; c2 += 256)  // (C <= 512: one pass)
      for (int tt = 0; tt < nt; ++tt) {
        const DT* dp = d + ((size_t)b * T0 + t0 + tt) * C + 2 * c2;
        const float d0 = (float)dp[0], d1 = (float)dp[1];
#pragma unroll
        for (int k = 0; k < 10; ++k)
          if (k < ksize) {
            const float xv = xs[tt * stride + k];
            acc[0][k] = fmaf(d0, xv, acc[0][k]);
            acc[1][k] = fmaf(d1, xv, acc[1][k]);
          }
      }
  }
  float* out = partial + ((size_t)b * gridDim.x + blk) * C * ksize;
  const int c2 = threadIdx.x;
  if (c2 < C / 2)
#pragma unroll
    for (int k = 0; k < 10; ++k)
      if (k < ksize) {
        out[(2 * c2) * ksize + k] = acc[0][k];
        out[(2 * c2 + 1) * ksize + k] = acc[1][k];
      }
}
__global__ void conv0_wgrad_sum_kernel(const float* __restrict__ partial, int nslab, int n, float* __restrict__ dw) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  for (int k = 0; k < nslab; ++k) s += partial[(size_t)k * n + e];
  dw[e] += s;
}

}  // namespace

size_t k_conv0_wgrad_scratch_floats(int B, int C, int ksize) { return (size_t)B * C0W_BLOCKS_PER_UTT * C * ksize; }

template <typename DT>
int k_conv0_wgrad_t(const DT* d, const float* x, float* dw, float* scratch, int B, int T, int T0, int C, int ksize, int stride,
                    hipStream_t st) {
  SSAK_REQUIRE(ksize <= 10 && stride * 64 + ksize <= 64 * 5 + 16 && C <= 512 && (C & 1) == 0,
               "conv0_wgrad: kernel %d / stride %d / C %d outside what is built (k <= 10, stride <= 5, C <= 512)", ksize, stride, C);
  conv0_wgrad_kernel<DT><<<dim3(C0W_BLOCKS_PER_UTT, B), 256, 0, st>>>(d, x, scratch, T, T0, C, ksize, stride);
  SSAK_LAUNCH_CHECK();
  conv0_wgrad_sum_kernel<<<ssak_cdiv(C * ksize, 256), 256, 0, st>>>(scratch, B * C0W_BLOCKS_PER_UTT, C * ksize, dw);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

size_t k_conv0_stats_doubles(int B, int T0, int C) {
  // [B][C] (mean, rstd) | per-workgroup partials: 2C (convolution-pass statistics) or NMOM (input moments) doubles each
  return (size_t)B * 2 * C + (size_t)B * ssak_cdiv(T0, FR_STATS) * std::max(2 * C, NMOM);
}

template <typename OT>
int k_conv0_gn_gelu_t(const float* x, const float* w, const float* gamma, const float* beta, OT* out, double* stats,
                      int B, int T, int T0, int C, int ksize, int stride, hipStream_t st, bool raw_input) {
  SSAK_REQUIRE(ksize == KS0 && stride == ST0, "conv0: only kernel 10 / stride 5 is built (got %d/%d)", ksize, stride);
  SSAK_REQUIRE((C & 3) == 0 && C <= 1024 && 256 % (C / 4) == 0, "conv0: C=%d must divide into 256 threads as quads", C);
  SSAK_REQUIRE(T0 == (T - KS0) / ST0 + 1 && T0 > 0, "conv0: T0 mismatch");
  // stats layout: [B][C] (mean, rstd) | [B][nblk][2C] per-workgroup partials   (k_conv0_stats_doubles(B, T0, C) doubles)
  const int nblk = ssak_cdiv(T0, FR_STATS);
  ProfScope prof_scope(PROF_CONV0, (double)B * ((double)T * 4.0 + (double)T0 * C * sizeof(OT)), st);  // waveform in, channels-last out
  double* sums = stats;
  double* partial = stats + (size_t)B * 2 * C;
  static const bool direct_stats = SSAK_DEV_ENV("SSAK_CONV0_DIRECT_STATS") != nullptr;  // development: the convolution-pass statistics
  SSAK_REQUIRE(!(raw_input && direct_stats), "conv0: the folded normalisation needs the moment statistics");
  if (direct_stats) {
    conv0_kernel<OT, false, FR_STATS><<<dim3(nblk, B), 256, 0, st>>>(x, w, gamma, beta, out, partial, nullptr, T, T0, C);
    SSAK_LAUNCH_CHECK();
    conv0_stats_finalize_kernel<<<dim3(ssak_cdiv(C, 256), B), 256, 0, st>>>(partial, nblk, C, T0, sums);
    SSAK_LAUNCH_CHECK();
  } else {  // 65 input moments per utterance, then the channels' sums in closed form (NMOM <= 2 C doubles per partial slot)
    conv0_moments_kernel<<<dim3(nblk, B), 256, 0, st>>>(x, T, T0, partial);
    SSAK_LAUNCH_CHECK();
    conv0_channel_stats_kernel<<<dim3(ssak_cdiv(C, 256), B), 256, 0, st>>>(partial, nblk, w, C, sums, T, T0, raw_input ? 1 : 0);
    SSAK_LAUNCH_CHECK();
  }
  static const bool no_mfma = SSAK_DEV_ENV("SSAK_CONV0_VALU") != nullptr;  // development: the VALU apply pass
  if constexpr (sizeof(OT) == 2) {
    if (C == 512 && !no_mfma) {
      conv0_mfma_kernel<<<dim3(ssak_cdiv(T0, C0M_FR * C0M_SUB), B), 256, 0, st>>>(x, w, gamma, beta, (bf16*)out, sums, T, T0);
      SSAK_LAUNCH_CHECK();
      return SSAK_OK;
    }
  }
  conv0_kernel<OT, true, FR_APPLY><<<dim3(ssak_cdiv(T0, FR_APPLY), B), 256, 0, st>>>(x, w, gamma, beta, out, nullptr, sums, T, T0, C);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <typename OT>
int k_conv0_bias_t(const float* x, const float* w, const float* bias, OT* out, int B, int T, int T0, int C, int ksize,
                   int stride, hipStream_t st) {
  SSAK_REQUIRE(ksize == KS0 && stride == ST0, "conv0: only kernel 10 / stride 5 is built (got %d/%d)", ksize, stride);
  SSAK_REQUIRE((C & 3) == 0 && C <= 1024 && 256 % (C / 4) == 0, "conv0: C=%d must divide into 256 threads as quads", C);
  ProfScope prof_scope(PROF_CONV0, (double)B * ((double)T * 4.0 + (double)T0 * C * sizeof(OT)), st);
  conv0_bias_kernel<OT><<<dim3(ssak_cdiv(T0, FR_APPLY), B), 256, 0, st>>>(x, w, bias, out, T, T0, C);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <typename OT>
int k_conv_weight_rearrange_t(const float* w, OT* out, int Co, int Ci, int k, hipStream_t st) {
  conv_w_rearrange_kernel<OT><<<min(2048, ssak_cdiv((long)Co * Ci * k, 256)), 256, 0, st>>>(w, out, Co, Ci, k);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <typename OT>
int k_posconv_prepare_t(const float* g, const float* v, OT* w_fwd, OT* w_bwd, float* norms, int H, int G, int K,
                        hipStream_t st) {
  SSAK_REQUIRE(K <= 1024 && H % G == 0 && ((H / G) & 7) == 0, "posconv: K=%d <= 1024 and (H/G)=%d %% 8 == 0 required", K, H / G);
  const int cg = H / G;
  // norms layout: [K] ||v||^2 | [K] dot scratch | [H][K] per-workgroup partials
  float* partial = norms + 2 * K;
  ProfScope prof_scope(PROF_POSCONV_W, (double)H * cg * K * 8.0, st);  // v read (fp32), two bf16 layouts written
  posconv_colnorm_kernel<<<H, ((K + 63) / 64) * 64, 0, st>>>(v, nullptr, H / cg, K, cg, partial);  // H = groups * cg workgroups
  SSAK_LAUNCH_CHECK();
  posconv_colnorm_finalize_kernel<<<ssak_cdiv(K, 64), 1024, 0, st>>>(partial, H, K, norms);
  SSAK_LAUNCH_CHECK();
  SSAK_REQUIRE((size_t)cg * (K + 1) * sizeof(float) <= 64 * 1024, "posconv: (H/G) x K slab does not fit in LDS");
  posconv_materialize_kernel<OT><<<2 * H, 256, (size_t)cg * (K + 1) * sizeof(float), st>>>(g, v, norms, w_fwd, w_bwd, H, cg, K);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

int k_posconv_weight_bwd(const float* dw, const float* g, const float* v, const float* norms, float* dg, float* dv,
                         int H, int G, int K, hipStream_t st) {
  // `dg` doubles as the scratch for dot[k] would alias the output; use the tail of `norms` (2K floats allocated)
  const int cg = H / G;
  float* dot = const_cast<float*>(norms) + K;
  float* partial = const_cast<float*>(norms) + 2 * K;
  ProfScope prof_scope(PROF_POSCONV_W, (double)H * cg * K * 12.0, st);  // dw and v read, dv written (fp32)
  posconv_colnorm_kernel<<<H, ((K + 63) / 64) * 64, 0, st>>>(v, dw, H / cg, K, cg, partial);
  SSAK_LAUNCH_CHECK();
  posconv_colnorm_finalize_kernel<<<ssak_cdiv(K, 64), 1024, 0, st>>>(partial, H, K, dot);
  SSAK_LAUNCH_CHECK();
  SSAK_REQUIRE((size_t)K * (cg + 1) * sizeof(float) <= 64 * 1024, "posconv: K x (H/G) slab does not fit in LDS");
  posconv_wbwd_kernel<<<H, 256, (size_t)K * (cg + 1) * sizeof(float), st>>>(dw, g, v, norms, dot, dg, dv, H, cg, K);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <typename T>
int k_posconv_pack_t(const T* h, T* pg, int B, int F, int H, int G, int K, hipStream_t st) {
  const int lead = K / 2, RS = F + K;
  const long rows_total = lead + (long)B * RS + K;
  const long n = rows_total * G * (H / G / 8);
  ProfScope prof_scope(PROF_ROWWISE, (double)B * F * H * 2.0 * sizeof(T), st);
  posconv_pack_kernel<T><<<min(4096, ssak_cdiv(n, 256)), 256, 0, st>>>(h, pg, B, F, H, G, lead, RS, rows_total);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <typename T_>
int k_col2im_t(const T_* dxcol, T_* dx, int B, int Tin, int Tout, int C, int k, int s, hipStream_t st) {
  SSAK_REQUIRE((C & 7) == 0, "col2im: C must be a multiple of 8");
  const long n8 = (long)B * Tin * C / 8;
  col2im_kernel<T_><<<min(8192, ssak_cdiv(n8, 256)), 256, 0, st>>>(dxcol, dx, Tin, Tout, C, k, s, n8);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

int k_sum_slabs(const float* slabs, int nb, long n, float* out, hipStream_t st) {
  sum_slabs_kernel<<<min(4096, ssak_cdiv(n, 256)), 256, 0, st>>>(slabs, nb, n, out);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

size_t k_conv0_bwd_scratch_floats(int B, int T0, int C) {
  const size_t nblk = ssak_cdiv(T0, FR_STATS);
  return (size_t)B * nblk * (C / 4) * (4 * KS0) + (size_t)B * C * 2 * 2 /* gsums as doubles */ + 64;
}

// dy: gradient w.r.t. the conv0 block output (post GELU) [B, T0, C] bf16; sums: the forward's [B][2C] statistics
template <typename DT>
int k_conv0_gn_gelu_bwd_t(const float* x, const float* w, const float* gamma, const float* beta, const DT* dy,
                          const double* sums, float* scratch, float* dw, float* dgamma, float* dbeta, int B, int T, int T0, int C,
                          hipStream_t st) {
  SSAK_REQUIRE((C & 3) == 0 && C <= 1024 && 256 % (C / 4) == 0, "conv0_bwd: C=%d must divide into 256 threads as quads", C);
  const int nblk = ssak_cdiv(T0, FR_STATS);
  double* gsums = reinterpret_cast<double*>(scratch);  // [B][C][2]
  float* partial = scratch + (size_t)B * C * 2 * 2 + 16;
  dim3 grid(nblk, B);
  conv0_bwd_kernel<0, DT><<<grid, 256, 256 * 9 * sizeof(float), st>>>(x, w, gamma, beta, dy, sums, nullptr, partial, T, T0, C);
  SSAK_LAUNCH_CHECK();
  conv0_bwd_gsums_kernel<<<dim3(ssak_cdiv(2 * C, 256), B), 256, 0, st>>>(partial, nblk, C, gsums);
  SSAK_LAUNCH_CHECK();
  conv0_bwd_affine_kernel<<<ssak_cdiv(C, 256), 256, 0, st>>>(gsums, B, C, dgamma, dbeta);
  SSAK_LAUNCH_CHECK();
  constexpr int lds1 = 256 * (4 * KS0 + 1) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    SSAK_HIP(hipFuncSetAttribute((const void*)conv0_bwd_kernel<1, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
    attr_done = true;
  }
  conv0_bwd_kernel<1, DT><<<grid, 256, lds1, st>>>(x, w, gamma, beta, dy, sums, gsums, partial, T, T0, C);
  SSAK_LAUNCH_CHECK();
  conv0_bwd_dw_kernel<<<ssak_cdiv(C * KS0, 256), 256, 0, st>>>(partial, B * nblk, C, dw);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

#define SSAK_INSTANTIATE_CONV_KERNELS(T)                                                                                       \
  template int k_conv0_gn_gelu_t<T>(const float*, const float*, const float*, const float*, T*, double*, int, int, int, int,   \
                                    int, int, hipStream_t, bool);                                                              \
  template int k_conv0_bias_t<T>(const float*, const float*, const float*, T*, int, int, int, int, int, int, hipStream_t);     \
  template int k_conv_weight_rearrange_t<T>(const float*, T*, int, int, int, hipStream_t);                                     \
  template int k_posconv_prepare_t<T>(const float*, const float*, T*, T*, float*, int, int, int, hipStream_t);                 \
  template int k_posconv_pack_t<T>(const T*, T*, int, int, int, int, int, hipStream_t);                                        \
  template int k_col2im_t<T>(const T*, T*, int, int, int, int, int, int, hipStream_t);                                        \
  template int k_conv0_wgrad_t<T>(const T*, const float*, float*, float*, int, int, int, int, int, int, hipStream_t);         \
  template int k_conv0_gn_gelu_bwd_t<T>(const float*, const float*, const float*, const float*, const T*, const double*,      \
                                        float*, float*, float*, float*, int, int, int, int, hipStream_t);
SSAK_INSTANTIATE_CONV_KERNELS(bf16)
SSAK_INSTANTIATE_CONV_KERNELS(float)

// exported for per-op parity tests (a3: first layer of the feature encoder)
extern "C" int ssak_conv0_gn_gelu(const float* x, const float* w, const float* gamma, const float* beta, void* out_bf16, void* workspace,
                                  size_t workspace_bytes, int B, int T, int C, void* stream) {
  SSAK_REQUIRE(x && w && gamma && beta && out_bf16 && workspace, "conv0_gn_gelu: null pointer");
  SSAK_REQUIRE(T >= KS0, "conv0_gn_gelu: T=%d shorter than the kernel", T);
  const int T0 = (T - KS0) / ST0 + 1;
  SSAK_REQUIRE(workspace_bytes >= k_conv0_stats_doubles(B, T0, C) * sizeof(double), "conv0_gn_gelu: workspace too small");
  return k_conv0_gn_gelu_t<bf16>(x, w, gamma, beta, (bf16*)out_bf16, (double*)workspace, B, T, T0, C, KS0, ST0, (hipStream_t)stream, false);
}
// the same on RAW full-length waveforms: the feature extractor's zero-mean / unit-variance normalisation (a1) folded into the
// GroupNorm statistics (conv_frontend.hip: NMOM) -- out == ssak_conv0_gn_gelu(ssak_wave_normalize(x)) to fp32 rounding
extern "C" int ssak_conv0_gn_gelu_raw(const float* x, const float* w, const float* gamma, const float* beta, void* out_bf16, void* workspace,
                                      size_t workspace_bytes, int B, int T, int C, void* stream) {
  SSAK_REQUIRE(x && w && gamma && beta && out_bf16 && workspace, "conv0_gn_gelu_raw: null pointer");
  SSAK_REQUIRE(T >= KS0, "conv0_gn_gelu_raw: T=%d shorter than the kernel", T);
  const int T0 = (T - KS0) / ST0 + 1;
  SSAK_REQUIRE(workspace_bytes >= k_conv0_stats_doubles(B, T0, C) * sizeof(double), "conv0_gn_gelu_raw: workspace too small");
  return k_conv0_gn_gelu_t<bf16>(x, w, gamma, beta, (bf16*)out_bf16, (double*)workspace, B, T, T0, C, KS0, ST0, (hipStream_t)stream, true);
}
extern "C" size_t ssak_conv0_workspace_bytes(int B, int T, int C) {
  if (T < KS0) return 0;
  return k_conv0_stats_doubles(B, (T - KS0) / ST0 + 1, C) * sizeof(double);
}
