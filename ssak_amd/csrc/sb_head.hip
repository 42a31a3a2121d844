// f4: the row-wise pieces of the SpeechBrain recipe's acoustic head.  gfx950.
//
// The recipe (ssak/train/speechbrain/wav2vec_train.py:39-56, modules of
// ssak/train/speechbrain/fr/hyperparameters_wav2vec_finetune_cv-fr.yaml:87-137) runs, on top of the wav2vec2 hidden states,
//   layer_norm over (frames x features) per utterance  ->  3 x [Linear -> BatchNorm1d -> LeakyReLU -> Dropout]  ->  Linear
//   -> log-softmax -> CTC,   Adadelta on the head, Adam on wav2vec2.
// The Linears are the library's GEMM, log-softmax + CTC is ssak_ctc_loss_fwd_bwd; this file holds what is left:
//   * utterance normalisation (F.layer_norm(x, x.shape[1:]), no affine) forward / backward, fp32 (waveform) or bf16 (features)
//   * BatchNorm1d over all B*T rows (+ LeakyReLU + dropout fused into the apply pass) forward / backward, running statistics
//   * the Adadelta update with the global-norm clip coefficient read on the device.
// All of it is HBM-bound streaming: statistics are two-stage (per-workgroup partials, fixed-order finalisation in double) so
// results do not depend on scheduling; x is read twice and y written once per direction (6 B per element forward).
#include "kernels.h"

namespace {
SSAK_DEFINE_DROP_TABLE

uint32_t drop_thresh(float p) { return p <= 0.f ? 0u : (uint32_t)fminf(65535.f, roundf(p * 65536.f)); }
float drop_scale(float p) { return p <= 0.f ? 1.f : 1.f / (1.f - (float)drop_thresh(p) / 65536.f); }

constexpr int BN_ROW_BLOCKS = 512;  // row groups of the partial-statistics passes

__device__ __forceinline__ void load8(const bf16* p, float* v) {
  const bf16x8 q = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (float)q[k];
}
__device__ __forceinline__ void store8(bf16* p, const float* v) {
  bf16x8 q;
#pragma unroll
  for (int k = 0; k < 8; ++k) q[k] = (bf16)v[k];
  *reinterpret_cast<bf16x8*>(p) = q;
}

// ---------------------------------------------------------------- BatchNorm1d over rows
// Thread mapping of the row-streaming kernels: a row is cpr = C / 8 chunks of 8 columns (16 B); a workgroup of NT threads
// covers cpb = min(cpr, NT) chunks (grid.y column blocks when a row is wider) and `lanes` = NT / cpb rows at a time, so that
// narrow and wide rows both keep every wave busy and each wave reads whole contiguous row segments.  Rows are taken four at
// a time with the loads issued first (memory-level parallelism: these passes are pure HBM streaming).
struct RowMap {
  int cc, rl, lanes, tcc, cpb;
  bool active;
  __device__ __forceinline__ RowMap(int C, int NT) {
    const int cpr = C >> 3;
    cpb = cpr < NT ? cpr : NT;
    lanes = NT / cpb;
    tcc = threadIdx.x % cpb;
    rl = threadIdx.x / cpb;
    cc = blockIdx.y * cpb + tcc;
    active = rl < lanes && cc < cpr;
  }
};
constexpr int BN_STAT_THREADS = 1024;

// sums of up to two per-column quantities over this workgroup's rows -> partial[blockIdx.x][0..1][c] (lanes reduced through LDS)
__device__ __forceinline__ void bn_block_reduce_store(const RowMap& m, float (&s)[8], float (&q)[8], float* __restrict__ partial,
                                                      int C, float* red) {
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    __syncthreads();
    if (m.rl < m.lanes) {
#pragma unroll
      for (int k = 0; k < 8; ++k) red[(k * m.lanes + m.rl) * m.cpb + m.tcc] = which ? q[k] : s[k];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < m.cpb * 8; t += blockDim.x) {
      const int k = t / m.cpb, tc = t % m.cpb;
      float v = 0.f;
      for (int l = 0; l < m.lanes; ++l) v += red[(k * m.lanes + l) * m.cpb + tc];
      const int c = (blockIdx.y * m.cpb + tc) * 8 + k;
      if (c < C) partial[((long)blockIdx.x * 2 + which) * C + c] = v;
    }
  }
}

// partial[rb][0][c] = sum x, partial[rb][1][c] = sum x^2 over the rows of workgroup rb (fp32: a few dozen terms per partial)
__global__ __launch_bounds__(BN_STAT_THREADS) void bn_stats_partial_kernel(const bf16* __restrict__ x, long ld, int M, int C,
                                                                           float* __restrict__ partial) {
  extern __shared__ float red[];
  const RowMap m(C, BN_STAT_THREADS);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (m.active) {
    const int stride = gridDim.x * m.lanes;
    for (int r = blockIdx.x * m.lanes + m.rl; r < M; r += 4 * stride) {
      bf16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ru = r + u * stride;
        v[u] = ru < M ? *reinterpret_cast<const bf16x8*>(x + (long)ru * ld + m.cc * 8) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float f = (float)v[u][k];
          s[k] += f;
          q[k] = fmaf(f, f, q[k]);
        }
    }
  }
  bn_block_reduce_store(m, s, q, partial, C, red);
}
// Column totals of the two partial planes: workgroup = 64 columns x 16 slot groups, fixed-order sums in double.
// (One thread per column walking all the partials serially was 150 us of dependent-latency loads.)
__device__ __forceinline__ void bn_partial_totals(const float* __restrict__ partial, int nb, int C, double& s, double& q,
                                                  double (*red)[16][64]) {
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  double ps = 0, pq = 0;
  if (c < C) {
#pragma unroll 4
    for (int b = ry; b < nb; b += 16) {
      ps += (double)partial[((long)b * 2) * C + c];
      pq += (double)partial[((long)b * 2 + 1) * C + c];
    }
  }
  red[0][ry][cx] = ps;
  red[1][ry][cx] = pq;
  __syncthreads();
  s = 0, q = 0;
  if (ry == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s += red[0][r][cx];
      q += red[1][r][cx];
    }
  }
}
// mean / rstd of the batch (biased variance), running statistics updated with the unbiased one (torch.nn.BatchNorm1d)
__global__ __launch_bounds__(1024) void bn_stats_final_kernel(const float* __restrict__ partial, int nb, int M, int C, float eps,
                                                              float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                              float* __restrict__ run_mean, float* __restrict__ run_var) {
  __shared__ double red[2][16][64];
  double s, q;
  bn_partial_totals(partial, nb, C, s, q, red);
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if ((threadIdx.x >> 6) != 0 || c >= C) return;
  const double mu = s / M;
  double var = q / M - mu * mu;
  var = var > 0 ? var : 0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    const double unb = M > 1 ? var * ((double)M / (M - 1)) : var;
    run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mu);
    run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
  }
}
__global__ __launch_bounds__(256) void bn_running_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                                         int C, float eps, float* __restrict__ mean, float* __restrict__ rstd) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  mean[c] = run_mean[c];
  rstd[c] = (float)(1.0 / sqrt((double)run_var[c] + (double)eps));
}
// synchronised BatchNorm (data parallel): the two column totals leave as doubles, are summed over the ranks by the caller
// (one all-reduce of 2 C doubles) and come back through the *_from_sums kernels with the global row count
__global__ __launch_bounds__(1024) void bn_sums_final_kernel(const float* __restrict__ partial, int nb, int M, int C,
                                                             double* __restrict__ sums) {
  __shared__ double red[2][16][64];
  double s, q;
  bn_partial_totals(partial, nb, C, s, q, red);
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if ((threadIdx.x >> 6) != 0 || c >= C) return;
  sums[c] = s;
  sums[C + c] = q;
  if (c == 0) sums[2 * C] = (double)M;  // the row count rides in the same all-reduce
}
__global__ __launch_bounds__(256) void bn_stats_from_sums_kernel(const double* __restrict__ sums, int C, float eps, float momentum,
                                                                 float* __restrict__ mean, float* __restrict__ rstd,
                                                                 float* __restrict__ run_mean, float* __restrict__ run_var) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double M = sums[2 * C];
  const double mu = sums[c] / M;
  double var = sums[C + c] / M - mu * mu;
  var = var > 0 ? var : 0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    const double unb = M > 1 ? var * (M / (M - 1)) : var;
    run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mu);
    run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
  }
}
__global__ __launch_bounds__(256) void bn_coef_from_sums_kernel(const double* __restrict__ sums, int C, float* __restrict__ coef) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double M = sums[2 * C];
  coef[c] = (float)(sums[c] / M);
  coef[C + c] = (float)(sums[C + c] / M);
}
// y = dropout(leaky_relu(gamma * (x - mean) * rstd + beta)); mask bit = common.h drop_keep(rowkey(seed, stream, r), colmul(c))
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16* __restrict__ x, long ldx, bf16* __restrict__ y, long ldy, int M,
                                                       int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                                                       uint64_t seed, uint32_t stream, uint32_t thresh, float dscale) {
  const RowMap m(C, 256);
  if (!m.active) return;
  const int c0 = m.cc * 8;
  uint32_t cm[8];  // dropout column multipliers of this thread's 8 columns (common.h)
#pragma unroll
  for (int k = 0; k < 8; ++k) cm[k] = thresh ? g_drop_colmul.v[c0 + k] : 1u;
  float mu[8], rs[8], gm[8], bt[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    mu[k] = mean[c0 + k];
    rs[k] = rstd[c0 + k];
    gm[k] = gamma[c0 + k];
    bt[k] = beta[c0 + k];
  }
  const int stride = gridDim.x * m.lanes;
  for (int r = blockIdx.x * m.lanes + m.rl; r < M; r += 4 * stride) {
    bf16x8 in[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ru = r + u * stride;
      if (ru < M) in[u] = *reinterpret_cast<const bf16x8*>(x + (long)ru * ldx + c0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ru = r + u * stride;
      if (ru >= M) break;
      const uint32_t rk = thresh ? drop_rowkey(seed, stream, (uint64_t)ru) : 1u;
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float z = fmaf(((float)in[u][k] - mu[k]) * rs[k], gm[k], bt[k]);  // the expression the backward re-evaluates (same sign of z)
        z = z > 0.f ? z : z * slope;
        if (thresh) z = drop_keep(rk, cm[k], thresh << 16) ? z * dscale : 0.f;
        v[k] = z;
      }
      store8(y + (long)ru * ldy + c0, v);
    }
  }
}
// g = dy * mask * scale * leaky'(z);  partial[rb][0][c] = sum g, partial[rb][1][c] = sum g * xhat
__global__ __launch_bounds__(BN_STAT_THREADS) void bn_bwd_partial_kernel(const bf16* __restrict__ dy, long ldy, const bf16* __restrict__ x,
                                                                         long ldx, int M, int C, const float* __restrict__ mean,
                                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                         const float* __restrict__ beta, float slope, uint64_t seed,
                                                                         uint32_t stream, uint32_t thresh, float dscale,
                                                                         float* __restrict__ partial) {
  extern __shared__ float red[];
  const RowMap m(C, BN_STAT_THREADS);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (m.active) {
    const int c0 = m.cc * 8;
    uint32_t cm[8];  // dropout column multipliers of this thread's 8 columns (common.h)
#pragma unroll
    for (int k = 0; k < 8; ++k) cm[k] = thresh ? g_drop_colmul.v[c0 + k] : 1u;
    float mu[8], rs[8], gm[8], bt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      mu[k] = mean[c0 + k];
      rs[k] = rstd[c0 + k];
      gm[k] = gamma[c0 + k];
      bt[k] = beta[c0 + k];
    }
    const int stride = gridDim.x * m.lanes;
    for (int r = blockIdx.x * m.lanes + m.rl; r < M; r += 2 * stride) {
      bf16x8 xv[2], dv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ru = r + u * stride;
        if (ru < M) {
          xv[u] = *reinterpret_cast<const bf16x8*>(x + (long)ru * ldx + c0);
          dv[u] = *reinterpret_cast<const bf16x8*>(dy + (long)ru * ldy + c0);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ru = r + u * stride;
        if (ru >= M) break;
        const uint32_t rk = thresh ? drop_rowkey(seed, stream, (uint64_t)ru) : 1u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float xh = ((float)xv[u][k] - mu[k]) * rs[k];
          const float z = fmaf(xh, gm[k], bt[k]);
          float g = z > 0.f ? (float)dv[u][k] : (float)dv[u][k] * slope;
          if (thresh) g = drop_keep(rk, cm[k], thresh << 16) ? g * dscale : 0.f;
          s[k] += g;
          q[k] = fmaf(g, xh, q[k]);
        }
      }
    }
  }
  bn_block_reduce_store(m, s, q, partial, C, red);
}
// dbeta = sum g, dgamma = sum g xhat; coef[0][c] = dbeta / M, coef[1][c] = dgamma / M for the dx pass
__global__ __launch_bounds__(1024) void bn_bwd_final_kernel(const float* __restrict__ partial, int nb, int M, int C,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ coef, double* __restrict__ sums) {
  __shared__ double red[2][16][64];
  double s, q;
  bn_partial_totals(partial, nb, C, s, q, red);
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if ((threadIdx.x >> 6) != 0 || c >= C) return;
  dbeta[c] = (float)s;
  dgamma[c] = (float)q;
  if (coef) {
    coef[c] = (float)(s / M);
    coef[C + c] = (float)(q / M);
  }
  if (sums) {
    sums[c] = s;
    sums[C + c] = q;
    if (c == 0) sums[2 * C] = (double)M;
  }
}
// dx = gamma * rstd * (g - mean(g) - xhat * mean(g xhat))
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16* __restrict__ dy, long ldy, const bf16* __restrict__ x, long ldx,
                                                           bf16* __restrict__ dx, long lddx, int M, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ coef, float slope, uint64_t seed,
                                                           uint32_t stream, uint32_t thresh, float dscale) {
  const RowMap m(C, 256);
  if (!m.active) return;
  const int c0 = m.cc * 8;
  uint32_t cm[8];  // dropout column multipliers of this thread's 8 columns (common.h)
#pragma unroll
  for (int k = 0; k < 8; ++k) cm[k] = thresh ? g_drop_colmul.v[c0 + k] : 1u;
  float mu[8], rs[8], gm[8], bt[8], c1[8], c2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    mu[k] = mean[c0 + k];
    rs[k] = rstd[c0 + k];
    gm[k] = gamma[c0 + k];
    bt[k] = beta[c0 + k];
    c1[k] = coef[c0 + k];
    c2[k] = coef[C + c0 + k];
  }
  const int stride = gridDim.x * m.lanes;
  for (int r = blockIdx.x * m.lanes + m.rl; r < M; r += 2 * stride) {
    bf16x8 xv[2], dv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ru = r + u * stride;
      if (ru < M) {
        xv[u] = *reinterpret_cast<const bf16x8*>(x + (long)ru * ldx + c0);
        dv[u] = *reinterpret_cast<const bf16x8*>(dy + (long)ru * ldy + c0);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ru = r + u * stride;
      if (ru >= M) break;
      const uint32_t rk = thresh ? drop_rowkey(seed, stream, (uint64_t)ru) : 1u;
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = ((float)xv[u][k] - mu[k]) * rs[k];
        const float z = fmaf(xh, gm[k], bt[k]);
        float g = z > 0.f ? (float)dv[u][k] : (float)dv[u][k] * slope;
        if (thresh) g = drop_keep(rk, cm[k], thresh << 16) ? g * dscale : 0.f;
        v[k] = gm[k] * rs[k] * (g - c1[k] - xh * c2[k]);
      }
      store8(dx + (long)ru * lddx + c0, v);
    }
  }
}

// ---------------------------------------------------------------- utterance normalisation
constexpr int UN_BLOCKS = 64;  // partial-sum workgroups per utterance

template <typename T>
__device__ __forceinline__ void loadv(const T* p, long i, float* v);  // 8 (bf16) or 4 (fp32) elements at chunk i
template <>
__device__ __forceinline__ void loadv<bf16>(const bf16* p, long i, float* v) { load8(p + i * 8, v); }
template <>
__device__ __forceinline__ void loadv<float>(const float* p, long i, float* v) {
  const float4 q = reinterpret_cast<const float4*>(p)[i];
  v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
}
template <typename T>
__device__ __forceinline__ void storev(T* p, long i, const float* v);
template <>
__device__ __forceinline__ void storev<bf16>(bf16* p, long i, const float* v) { store8(p + i * 8, v); }
template <>
__device__ __forceinline__ void storev<float>(float* p, long i, const float* v) {
  reinterpret_cast<float4*>(p)[i] = make_float4(v[0], v[1], v[2], v[3]);
}
template <typename T>
struct Chunk {
  static constexpr int N = sizeof(T) == 2 ? 8 : 4;
};

// partial[b][blk][0..1] = (sum a, sum a*b) over this workgroup's share of utterance b; B2 == nullptr: (sum a, sum a^2)
template <typename T>
__global__ __launch_bounds__(256) void un_partial_kernel(const T* __restrict__ A, const T* __restrict__ B2, long n,
                                                         float* __restrict__ partial) {
  __shared__ float red[16];
  constexpr int N = Chunk<T>::N;
  const long base = (long)blockIdx.y * n;
  const long nch = (n % N == 0) ? n / N : 0;  // rows of other lengths are not 16-byte aligned: element by element
  float s = 0.f, q = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nch; i += (long)gridDim.x * 256) {
    float a[8], b[8];
    loadv<T>(A + base, i, a);
    if (B2) loadv<T>(B2 + base, i, b);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      s += a[k];
      q = fmaf(a[k], B2 ? b[k] : a[k], q);
    }
  }
  for (long i = nch * N + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float a = (float)A[base + i], b = B2 ? (float)B2[base + i] : a;
    s += a;
    q = fmaf(a, b, q);
  }
  s = block_sum(s, red);
  q = block_sum(q, red);
  if (threadIdx.x == 0) {
    partial[((long)blockIdx.y * gridDim.x + blockIdx.x) * 2] = s;
    partial[((long)blockIdx.y * gridDim.x + blockIdx.x) * 2 + 1] = q;
  }
}
// forward: y = (x - mean) * rstd;  stats[b] = (mean, rstd)
template <typename T>
__global__ __launch_bounds__(256) void un_fwd_apply_kernel(const T* __restrict__ x, T* __restrict__ y, long n, float eps,
                                                           const float* __restrict__ partial, int nparts,
                                                           float* __restrict__ stats) {
  constexpr int N = Chunk<T>::N;
  __shared__ float mr[2];
  if (threadIdx.x == 0) {
    double s = 0, q = 0;
    for (int i = 0; i < nparts; ++i) {
      s += (double)partial[((long)blockIdx.y * nparts + i) * 2];
      q += (double)partial[((long)blockIdx.y * nparts + i) * 2 + 1];
    }
    const double mu = s / n;
    double var = q / n - mu * mu;
    var = var > 0 ? var : 0;
    mr[0] = (float)mu;
    mr[1] = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.x == 0 && stats) {
      stats[blockIdx.y * 2] = mr[0];
      stats[blockIdx.y * 2 + 1] = mr[1];
    }
  }
  __syncthreads();
  const float mu = mr[0], rs = mr[1];
  const long base = (long)blockIdx.y * n;
  const long nch = (n % N == 0) ? n / N : 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nch; i += (long)gridDim.x * 256) {
    float a[8];
    loadv<T>(x + base, i, a);
#pragma unroll
    for (int k = 0; k < N; ++k) a[k] = (a[k] - mu) * rs;
    storev<T>(y + base, i, a);
  }
  for (long i = nch * N + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[base + i] = (T)(((float)x[base + i] - mu) * rs);
}
// backward: dx = rstd * (dy - mean(dy) - y * mean(dy * y))   (y is the forward's output = xhat)
template <typename T>
__global__ __launch_bounds__(256) void un_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                           long n, const float* __restrict__ partial, int nparts,
                                                           const float* __restrict__ stats) {
  constexpr int N = Chunk<T>::N;
  __shared__ float cc[2];
  if (threadIdx.x == 0) {
    double s = 0, q = 0;
    for (int i = 0; i < nparts; ++i) {
      s += (double)partial[((long)blockIdx.y * nparts + i) * 2];
      q += (double)partial[((long)blockIdx.y * nparts + i) * 2 + 1];
    }
    cc[0] = (float)(s / n);
    cc[1] = (float)(q / n);
  }
  __syncthreads();
  const float c1 = cc[0], c2 = cc[1], rs = stats[blockIdx.y * 2 + 1];
  const long base = (long)blockIdx.y * n;
  const long nch = (n % N == 0) ? n / N : 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nch; i += (long)gridDim.x * 256) {
    float d[8], h[8];
    loadv<T>(dy + base, i, d);
    loadv<T>(y + base, i, h);
#pragma unroll
    for (int k = 0; k < N; ++k) d[k] = rs * (d[k] - c1 - h[k] * c2);
    storev<T>(dx + base, i, d);
  }
  for (long i = nch * N + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    dx[base + i] = (T)(rs * ((float)dy[base + i] - c1 - (float)y[base + i] * c2));
}

// ---------------------------------------------------------------- Adadelta
__global__ __launch_bounds__(256) void adadelta_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq,
                                                       float* __restrict__ acc, bf16* __restrict__ shadow, long n,
                                                       const float* __restrict__ gnorm_sq, float max_norm, float grad_scale,
                                                       float lr, float rho, float eps, float wd) {
  float coef = grad_scale;
  if (gnorm_sq && max_norm > 0.f) coef *= fminf(1.f, max_norm / (sqrtf(gnorm_sq[0]) * grad_scale + 1e-6f));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float pk = p[i];
    float gk = g[i] * coef;
    if (wd != 0.f) gk = fmaf(wd, pk, gk);
    const float s = rho * sq[i] + (1.f - rho) * gk * gk;
    const float a0 = acc[i];
    const float delta = sqrtf(a0 + eps) / sqrtf(s + eps) * gk;
    sq[i] = s;
    acc[i] = rho * a0 + (1.f - rho) * delta * delta;
    pk -= lr * delta;
    p[i] = pk;
    if (shadow) shadow[i] = (bf16)pk;
  }
}

// launch geometry matching RowMap
struct BnGrid {
  int nb, cb_stat, ab, cb_apply;
  size_t lds;
  BnGrid(int M, int C) {
    const int cpr = C / 8;
    const int cpb_s = cpr < BN_STAT_THREADS ? cpr : BN_STAT_THREADS, lanes_s = BN_STAT_THREADS / cpb_s;
    cb_stat = ssak_cdiv(cpr, cpb_s);
    nb = ssak_cdiv(M, lanes_s) < BN_ROW_BLOCKS ? ssak_cdiv(M, lanes_s) : BN_ROW_BLOCKS;
    lds = (size_t)8 * lanes_s * cpb_s * sizeof(float);
    const int cpb_a = cpr < 256 ? cpr : 256, lanes_a = 256 / cpb_a;
    cb_apply = ssak_cdiv(cpr, cpb_a);
    const int want = ssak_cdiv(M, lanes_a * 4);
    ab = want < 1 ? 1 : (want > 4096 ? 4096 : want);
  }
};

template <typename T>
int utt_norm_fwd(const T* x, T* y, int B, long n, float eps, float* stats, float* ws, hipStream_t st) {
  un_partial_kernel<T><<<dim3(UN_BLOCKS, B), 256, 0, st>>>(x, (const T*)nullptr, n, ws);
  SSAK_LAUNCH_CHECK();
  const int ab = (int)fmin(512.0, (double)ssak_cdiv(n, 256 * Chunk<T>::N));
  un_fwd_apply_kernel<T><<<dim3(ab, B), 256, 0, st>>>(x, y, n, eps, ws, UN_BLOCKS, stats);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
template <typename T>
int utt_norm_bwd(const T* dy, const T* y, T* dx, int B, long n, const float* stats, float* ws, hipStream_t st) {
  un_partial_kernel<T><<<dim3(UN_BLOCKS, B), 256, 0, st>>>(dy, y, n, ws);
  SSAK_LAUNCH_CHECK();
  const int ab = (int)fmin(512.0, (double)ssak_cdiv(n, 256 * Chunk<T>::N));
  un_bwd_apply_kernel<T><<<dim3(ab, B), 256, 0, st>>>(dy, y, dx, n, ws, UN_BLOCKS, stats);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

}  // namespace

extern "C" size_t ssak_utt_norm_workspace_bytes(int B) { return (size_t)(B > 0 ? B : 0) * UN_BLOCKS * 2 * sizeof(float); }

extern "C" int ssak_utt_norm_fwd(const void* x, void* y, int B, long n, int is_bf16, float eps, float* stats, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(x && y && workspace && B > 0 && n > 0, "utt_norm_fwd: bad arguments");
  SSAK_REQUIRE(workspace_bytes >= ssak_utt_norm_workspace_bytes(B), "utt_norm_fwd: workspace too small");
  SSAK_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "utt_norm_fwd: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  return is_bf16 ? utt_norm_fwd<bf16>((const bf16*)x, (bf16*)y, B, n, eps, stats, (float*)workspace, st)
                 : utt_norm_fwd<float>((const float*)x, (float*)y, B, n, eps, stats, (float*)workspace, st);
}

extern "C" int ssak_utt_norm_bwd(const void* dy, const void* y, void* dx, int B, long n, int is_bf16, const float* stats,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(dy && y && dx && stats && workspace && B > 0 && n > 0, "utt_norm_bwd: bad arguments");
  SSAK_REQUIRE(workspace_bytes >= ssak_utt_norm_workspace_bytes(B), "utt_norm_bwd: workspace too small");
  SSAK_REQUIRE((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx) & 15) == 0, "utt_norm_bwd: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  return is_bf16 ? utt_norm_bwd<bf16>((const bf16*)dy, (const bf16*)y, (bf16*)dx, B, n, stats, (float*)workspace, st)
                 : utt_norm_bwd<float>((const float*)dy, (const float*)y, (float*)dx, B, n, stats, (float*)workspace, st);
}

extern "C" size_t ssak_batchnorm_workspace_bytes(int C) { return (size_t)(BN_ROW_BLOCKS * 2 + 2) * (C > 0 ? C : 0) * sizeof(float); }

extern "C" int ssak_batchnorm_stats(const void* x, int M, int C, double* sums, void* workspace, size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(x && sums && M > 0 && C > 0 && C % 8 == 0 && ((uintptr_t)x & 15) == 0, "batchnorm_stats: bad arguments");
  SSAK_REQUIRE(workspace && workspace_bytes >= ssak_batchnorm_workspace_bytes(C), "batchnorm_stats: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const BnGrid gd(M, C);
  bn_stats_partial_kernel<<<dim3(gd.nb, gd.cb_stat), BN_STAT_THREADS, gd.lds, st>>>((const bf16*)x, C, M, C, (float*)workspace);
  SSAK_LAUNCH_CHECK();
  bn_sums_final_kernel<<<ssak_cdiv(C, 64), 1024, 0, st>>>((const float*)workspace, gd.nb, M, C, sums);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

extern "C" int ssak_batchnorm_act_fwd(const void* x, void* y, int M, int C, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, float momentum, float eps, int training,
                                      float leaky_slope, float drop_p, uint64_t seed, uint32_t drop_stream, float* save_mean,
                                      float* save_rstd, const double* global_sums, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  SSAK_REQUIRE(x && y && gamma && beta && save_mean && save_rstd && M > 0 && C > 0, "batchnorm_act_fwd: bad arguments");
  SSAK_REQUIRE(C % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, "batchnorm_act_fwd: C must be a multiple of 8, rows 16-byte aligned");
  SSAK_REQUIRE(!(drop_p > 0.f) || C <= DROP_TABLE_N, "batchnorm_act_fwd: dropout sites are built for at most %d columns", DROP_TABLE_N);
  SSAK_REQUIRE(training || (running_mean && running_var), "batchnorm_act_fwd: evaluation needs the running statistics");
  SSAK_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "batchnorm_act_fwd: running_mean and running_var go together");
  SSAK_REQUIRE(!global_sums || training, "batchnorm_act_fwd: global sums are a training-mode input");
  hipStream_t st = (hipStream_t)stream;
  const BnGrid gd(M, C);
  const int nb = gd.nb;
  if (training && global_sums) {
    bn_stats_from_sums_kernel<<<ssak_cdiv(C, 256), 256, 0, st>>>(global_sums, C, eps, momentum, save_mean, save_rstd,
                                                                 running_mean, running_var);
    SSAK_LAUNCH_CHECK();
  } else if (training) {
    SSAK_REQUIRE(workspace && workspace_bytes >= ssak_batchnorm_workspace_bytes(C), "batchnorm_act_fwd: workspace too small");
    bn_stats_partial_kernel<<<dim3(nb, gd.cb_stat), BN_STAT_THREADS, gd.lds, st>>>((const bf16*)x, C, M, C, (float*)workspace);
    SSAK_LAUNCH_CHECK();
    bn_stats_final_kernel<<<ssak_cdiv(C, 64), 1024, 0, st>>>((const float*)workspace, nb, M, C, eps, momentum, save_mean, save_rstd,
                                                              running_mean, running_var);
    SSAK_LAUNCH_CHECK();
  } else {
    bn_running_kernel<<<ssak_cdiv(C, 256), 256, 0, st>>>(running_mean, running_var, C, eps, save_mean, save_rstd);
    SSAK_LAUNCH_CHECK();
  }
  const float p = training ? drop_p : 0.f;
  bn_apply_kernel<<<dim3(gd.ab, gd.cb_apply), 256, 0, st>>>((const bf16*)x, C, (bf16*)y, C, M, C, save_mean, save_rstd, gamma, beta, leaky_slope,
                                               seed, drop_stream, drop_thresh(p), drop_scale(p));
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

// Backward in one call (global_sums == NULL, local_sums_out == NULL), or in two around the caller's all-reduce:
//   call 1: dx == NULL, local_sums_out != NULL  -> dgamma / dbeta (local) and the two local totals as doubles
//   call 2: global_sums != NULL                 -> dx from the global totals / global row count; dgamma / dbeta are not touched
extern "C" int ssak_batchnorm_act_bwd(const void* dy, const void* x, void* dx, int M, int C, const float* gamma, const float* beta,
                                      const float* save_mean, const float* save_rstd, float leaky_slope, float drop_p, uint64_t seed,
                                      uint32_t drop_stream, float* dgamma, float* dbeta, double* local_sums_out,
                                      const double* global_sums, void* workspace, size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(dy && x && gamma && beta && save_mean && save_rstd && M > 0 && C > 0, "batchnorm_act_bwd: bad arguments");
  SSAK_REQUIRE(dx || local_sums_out, "batchnorm_act_bwd: nothing to compute");
  SSAK_REQUIRE(global_sums || (dgamma && dbeta), "batchnorm_act_bwd: dgamma / dbeta needed");
  SSAK_REQUIRE(!global_sums || dx, "batchnorm_act_bwd: global sums come with dx");
  SSAK_REQUIRE(C % 8 == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0,
               "batchnorm_act_bwd: C must be a multiple of 8, rows 16-byte aligned");
  SSAK_REQUIRE(!(drop_p > 0.f) || C <= DROP_TABLE_N, "batchnorm_act_bwd: dropout sites are built for at most %d columns", DROP_TABLE_N);
  SSAK_REQUIRE(workspace && workspace_bytes >= ssak_batchnorm_workspace_bytes(C), "batchnorm_act_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const BnGrid gd(M, C);
  const int nb = gd.nb;
  float* partial = (float*)workspace;
  float* coef = partial + (size_t)BN_ROW_BLOCKS * 2 * C;
  const uint32_t th = drop_thresh(drop_p);
  const float ds = drop_scale(drop_p);
  if (global_sums) {
    bn_coef_from_sums_kernel<<<ssak_cdiv(C, 256), 256, 0, st>>>(global_sums, C, coef);
    SSAK_LAUNCH_CHECK();
  } else {
    bn_bwd_partial_kernel<<<dim3(nb, gd.cb_stat), BN_STAT_THREADS, gd.lds, st>>>((const bf16*)dy, C, (const bf16*)x, C, M, C, save_mean, save_rstd, gamma, beta,
                                                       leaky_slope, seed, drop_stream, th, ds, partial);
    SSAK_LAUNCH_CHECK();
    bn_bwd_final_kernel<<<ssak_cdiv(C, 64), 1024, 0, st>>>(partial, nb, M, C, dgamma, dbeta, dx ? coef : nullptr, local_sums_out);
    SSAK_LAUNCH_CHECK();
  }
  if (dx) {
    bn_bwd_apply_kernel<<<dim3(gd.ab, gd.cb_apply), 256, 0, st>>>((const bf16*)dy, C, (const bf16*)x, C, (bf16*)dx, C, M, C, save_mean, save_rstd,
                                                     gamma, beta, coef, leaky_slope, seed, drop_stream, th, ds);
    SSAK_LAUNCH_CHECK();
  }
  return SSAK_OK;
}

extern "C" int ssak_adadelta_step(float* params, const float* grads, float* square_avg, float* acc_delta, void* shadow_bf16, long n,
                                  const float* gnorm_sq, float max_norm, float grad_scale, float lr, float rho, float eps,
                                  float weight_decay, void* stream) {
  SSAK_REQUIRE(params && grads && square_avg && acc_delta && n > 0, "adadelta: bad arguments");
  adadelta_kernel<<<(int)fmin(2048.0, (double)ssak_cdiv(n, 256)), 256, 0, (hipStream_t)stream>>>(
      params, grads, square_avg, acc_delta, (bf16*)shadow_bf16, n, gnorm_sq, max_norm, grad_scale, lr, rho, eps, weight_decay);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

namespace {
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16* __restrict__ in, float* __restrict__ out, long n) {
  const long n8 = n >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const bf16x8 q = reinterpret_cast<const bf16x8*>(in)[i];
    reinterpret_cast<float4*>(out)[2 * i] = make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
    reinterpret_cast<float4*>(out)[2 * i + 1] = make_float4((float)q[4], (float)q[5], (float)q[6], (float)q[7]);
  }
  if (blockIdx.x == 0)
    for (long i = (n8 << 3) + threadIdx.x; i < n; i += blockDim.x) out[i] = (float)in[i];
}
}  // namespace

extern "C" int ssak_cast_bf16_f32(const void* src_bf16, float* dst, long n, void* stream) {
  SSAK_REQUIRE(src_bf16 && dst && n > 0 && (((uintptr_t)src_bf16 | (uintptr_t)dst) & 15) == 0, "cast_bf16_f32: bad arguments");
  cast_bf16_f32_kernel<<<(int)fmin(4096.0, (double)ssak_cdiv(n, 2048)), 256, 0, (hipStream_t)stream>>>((const bf16*)src_bf16, dst, n);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

extern "C" int ssak_cast_f32_bf16(const float* src, void* dst_bf16, long n, void* stream) {
  SSAK_REQUIRE(src && dst_bf16 && n > 0, "cast_f32_bf16: bad arguments");
  return k_cast_f32_bf16(src, (bf16*)dst_bf16, n, (hipStream_t)stream);
}

extern "C" size_t ssak_colsum_workspace_bytes(int N) { return (size_t)64 * (N > 0 ? N : 0) * sizeof(float); }

extern "C" int ssak_colsum_bf16(const void* X, long ld, int M, int N, float* out, void* workspace, size_t workspace_bytes,
                                void* stream) {
  SSAK_REQUIRE(X && out && workspace && M > 0 && N > 0, "colsum_bf16: bad arguments");
  SSAK_REQUIRE(workspace_bytes >= ssak_colsum_workspace_bytes(N), "colsum_bf16: workspace too small");
  SSAK_HIP(hipMemsetAsync(out, 0, (size_t)N * sizeof(float), (hipStream_t)stream));
  return k_colsum((const bf16*)X, ld, M, N, out, (hipStream_t)stream, (float*)workspace, workspace_bytes / sizeof(float));
}
