// Row-wise and element-wise kernels around the GEMMs of the encoder (all HBM-bound).  gfx950.
//
//  * residual + dropout + LayerNorm forward / backward   (transformers modeling_wav2vec2.py:429-434, :591-608,
//    :631-654, :667-726 -- feature projection LN, encoder LN, per-layer post-/pre-LN)
//  * attention softmax forward / backward with key-padding mask and dropout (:438-463)
//  * column sums (bias gradients), dtype casts, SpecAugment scatter (:1272-1316)
// One wave owns one row; 16-byte vector accesses; statistics in fp32; two-pass variance as torch does.
// Dropout masks are recomputed from (seed, stream, element index) in the backward kernels.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

SSAK_DEFINE_DROP_TABLE

constexpr int ROW_THREADS = 256;  // 4 waves = 4 rows per workgroup
// the dropout column multipliers of a lane's chunk (common.h: word = rowkey * colmul): VEC consecutive table entries
template <int VEC>
__device__ __forceinline__ void load_colmul(uint32_t (&cm)[VEC], int col0) {
#pragma unroll
  for (int k = 0; k < VEC; k += 4) {
    const uint4 t = *reinterpret_cast<const uint4*>(g_drop_colmul.v + col0 + k);
    cm[k] = t.x, cm[k + 1] = t.y, cm[k + 2] = t.z, cm[k + 3] = t.w;
  }
}

// chunk of VEC elements of type T (common.h: Chunk8 / Chunk4)
template <typename T, int VEC>
using ChunkT = typename std::conditional<VEC == 8, Chunk8<T>, Chunk4<T>>::type;
template <typename T, int VEC>
__device__ __forceinline__ ChunkT<T, VEC> f_to_chunk(const float* f) {
  if constexpr (VEC == 8) return f_to_chunk8<T>(f);
  else return f_to_chunk4<T>(f);
}

template <typename T>
struct LnFwdParams {
  const T* y;         // [M,C] branch output (may be null -> r = res)
  const T* res;       // [M,C] residual (may be null)
  const float* gamma;
  const float* beta;
  T* r_out;           // [M,C] r = res + drop(y)   (may be null)
  T* out;             // [M,C] LN(r) (then optional post-dropout); null -> no LN (plain residual add)
  float* mean;
  float* rstd;
  int M, C;
  float eps;
  uint64_t seed;
  uint32_t pre_stream, pre_thresh, post_stream, post_thresh;
  float pre_scale, post_scale;
  uint32_t mid_stream, mid_thresh;  // dropout applied to the SUM res + drop(y) (stable-LN encoder input)
  float mid_scale;
  int post_gelu;                    // out = gelu(LN(r))  (layer-norm feature-encoder conv layers)
};

template <typename T, int NCH>
__global__ __launch_bounds__(ROW_THREADS) void ln_fwd_kernel(const LnFwdParams<T> p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (ROW_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int nch = p.C >> 3;
  float v[NCH][8];
  float s = 0.f;
  // dropout words = row key (one hash per site and row) x column multiplier (table): common.h
  const bool any_drop = (p.pre_thresh | p.mid_thresh | p.post_thresh) != 0;  // (uniform)
  const uint32_t rk_pre = p.pre_thresh ? drop_rowkey(p.seed, p.pre_stream, (uint64_t)row) : 1u;
  const uint32_t rk_mid = p.mid_thresh ? drop_rowkey(p.seed, p.mid_stream, (uint64_t)row) : 1u;
  const uint32_t rk_post = p.post_thresh ? drop_rowkey(p.seed, p.post_stream, (uint64_t)row) : 1u;
  const uint32_t thi_pre = p.pre_thresh << 16, thi_mid = p.mid_thresh << 16, thi_post = p.post_thresh << 16;
  uint32_t cm[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[i][k] = 0.f;
    if (ch < nch) {
      const size_t o = (size_t)row * p.C + ch * 8;
      if (any_drop) load_colmul<8>(cm[i], ch * 8);
      if (p.y) {
        chunk_to_f(ld8<T>(p.y + o), v[i]);
        if (p.pre_thresh) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[i][k] = drop_keep(rk_pre, cm[i][k], thi_pre) ? v[i][k] * p.pre_scale : 0.f;
        }
      }
      if (p.res) {
        float r[8];
        chunk_to_f(ld8<T>(p.res + o), r);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[i][k] += r[k];
      }
      if (p.mid_thresh) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[i][k] = drop_keep(rk_mid, cm[i][k], thi_mid) ? v[i][k] * p.mid_scale : 0.f;
      }
      if (p.r_out) {
        // round through the storage type so that forward and backward see the same LN input
        const Chunk8<T> q = f_to_chunk8<T>(v[i]);
        st8<T>(p.r_out + o, q);
        chunk_to_f(q, v[i]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[i][k];
    }
  }
  if (!p.out) return;
  const float mean = wave_sum(s) / (float)p.C;
  float q2 = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (lane + 64 * i < nch) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float d = v[i][k] - mean;
        q2 += d * d;
      }
    }
  const float rstd = rsqrtf(wave_sum(q2) / (float)p.C + p.eps);
  if (lane == 0 && p.mean) {
    p.mean[row] = mean;
    p.rstd[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      const size_t o = (size_t)row * p.C + ch * 8;
      float g[8], b[8], w[8];
      *reinterpret_cast<float4*>(g) = *reinterpret_cast<const float4*>(p.gamma + ch * 8);
      *reinterpret_cast<float4*>(g + 4) = *reinterpret_cast<const float4*>(p.gamma + ch * 8 + 4);
      *reinterpret_cast<float4*>(b) = *reinterpret_cast<const float4*>(p.beta + ch * 8);
      *reinterpret_cast<float4*>(b + 4) = *reinterpret_cast<const float4*>(p.beta + ch * 8 + 4);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        w[k] = (v[i][k] - mean) * rstd * g[k] + b[k];
        if (p.post_gelu) w[k] = gelu_s<T>(w[k]);
        if (p.post_thresh) w[k] = drop_keep(rk_post, cm[i][k], thi_post) ? w[k] * p.post_scale : 0.f;
      }
      st8<T>(p.out + o, f_to_chunk8<T>(w));
    }
  }
}

// The encoder's hot shapes (C = 768: three chunks of 4 per lane, every lane busy; C = 1024: two chunks of 8), SPECIALISED like
// ln_bwd_kernel: which operands, outputs and dropout sites exist is a compile-time bit set (1 branch y, 2 residual, 4 pre-,
// 8 sum-, 16 post-dropout, 32 r_out), a wave strides over rows and keeps gamma, beta and the dropout column multipliers in
// registers for all of them (the one-row-per-wave form above reads them again for every row: 144 of its 336 bytes of loads per
// lane and row at C = 768), and the next row's operands are fetched while this row's two wave reductions run.
template <typename T, int NCH, int VEC, int SPEC>
__global__ __launch_bounds__(ROW_THREADS) void ln_fwd_rows_kernel(const LnFwdParams<T> p) {
  using VecT = ChunkT<T, VEC>;
  constexpr bool HAS_Y = (SPEC & 1) != 0, HAS_RES = (SPEC & 2) != 0, D_PRE = (SPEC & 4) != 0, D_MID = (SPEC & 8) != 0,
                 D_POST = (SPEC & 16) != 0, HAS_ROUT = (SPEC & 32) != 0, DROP = (SPEC & 28) != 0;
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * (ROW_THREADS / 64) + (threadIdx.x >> 6);
  const int rstep = gridDim.x * (ROW_THREADS / 64);
  float gm[NCH][VEC], bt[NCH][VEC];
  uint32_t cm[DROP ? NCH : 1][VEC];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c0 = (lane + 64 * i) * VEC;
#pragma unroll
    for (int k = 0; k < VEC; ++k) gm[i][k] = p.gamma[c0 + k], bt[i][k] = p.beta[c0 + k];
    if (DROP) load_colmul<VEC>(cm[i], c0);
  }
  const uint32_t thi_pre = p.pre_thresh << 16, thi_mid = p.mid_thresh << 16, thi_post = p.post_thresh << 16;
  const float inv_c = 1.f / (float)p.C;
  VecT ny[NCH], nr[NCH];
  auto fetch = [&](int row) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const size_t o = (size_t)row * p.C + (lane + 64 * i) * VEC;
      if (HAS_Y) ny[i] = *reinterpret_cast<const VecT*>(p.y + o);
      if (HAS_RES) nr[i] = *reinterpret_cast<const VecT*>(p.res + o);
    }
  };
  if (wid < p.M) fetch(wid);
  for (int row = wid; row < p.M; row += rstep) {
    VecT cy[NCH], cr[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) cy[i] = ny[i], cr[i] = nr[i];
    if (row + rstep < p.M) fetch(row + rstep);
    const uint32_t rk_pre = D_PRE ? drop_rowkey(p.seed, p.pre_stream, (uint64_t)row) : 1u;
    const uint32_t rk_mid = D_MID ? drop_rowkey(p.seed, p.mid_stream, (uint64_t)row) : 1u;
    const uint32_t rk_post = D_POST ? drop_rowkey(p.seed, p.post_stream, (uint64_t)row) : 1u;
    float v[NCH][VEC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const size_t o = (size_t)row * p.C + (lane + 64 * i) * VEC;
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[i][k] = 0.f;
      if (HAS_Y) {
        chunk_to_f(cy[i], v[i]);
        if (D_PRE) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) v[i][k] = drop_keep(rk_pre, cm[DROP ? i : 0][k], thi_pre) ? v[i][k] * p.pre_scale : 0.f;
        }
      }
      if (HAS_RES) {
        float r[VEC];
        chunk_to_f(cr[i], r);
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[i][k] += r[k];
      }
      if (D_MID) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[i][k] = drop_keep(rk_mid, cm[DROP ? i : 0][k], thi_mid) ? v[i][k] * p.mid_scale : 0.f;
      }
      if (HAS_ROUT) {
        // round through the storage type so that forward and backward see the same LN input
        const VecT q = f_to_chunk<T, VEC>(v[i]);
        *reinterpret_cast<VecT*>(p.r_out + o) = q;
        chunk_to_f(q, v[i]);
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) s += v[i][k];
    }
    const float mean = wave_sum(s) * inv_c;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float d = v[i][k] - mean;
        q2 += d * d;
      }
    const float rstd = rsqrtf(wave_sum(q2) * inv_c + p.eps);
    if (lane == 0 && p.mean) {
      p.mean[row] = mean;
      p.rstd[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const size_t o = (size_t)row * p.C + (lane + 64 * i) * VEC;
      float w[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        w[k] = (v[i][k] - mean) * rstd * gm[i][k] + bt[i][k];
        if (D_POST) w[k] = drop_keep(rk_post, cm[DROP ? i : 0][k], thi_post) ? w[k] * p.post_scale : 0.f;
      }
      *reinterpret_cast<VecT*>(p.out + o) = f_to_chunk<T, VEC>(w);
    }
  }
}

template <typename T>
struct LnBwdParams {
  const T* g1;        // [M,C] grad wrt (post-dropout) LN output
  const T* g2;        // [M,C] second contribution or null
  const T* r;         // [M,C] saved LN input
  const float* mean;
  const float* rstd;
  const float* gamma;
  const T* g_res;     // [M,C] extra gradient added to dr AFTER the LN backward (pre-LN residual stream) or null
  T* dr;              // [M,C] grad wrt r (residual path)
  T* dy;              // [M,C] grad wrt y = dr * premask/(1-p)  (null when no pre-dropout: use dr)
  float* dgamma;      // [C] += (finalize kernel)
  float* dbeta;
  float* partial;     // [gridDim.x][3][C] per-workgroup column partials
  float* dy_colsum;   // [C] += column sums of dy (the bias gradient of the Linear that produced the branch) or null
  const float* beta;  // non-null: the forward applied GELU after the affine (layer-norm conv layers of the XLSR feature
                      // encoder): the incoming gradient is multiplied by gelu'(xhat * gamma + beta) first
  int M, C;
  uint64_t seed;
  uint32_t pre_stream, pre_thresh, post_stream, post_thresh;
  float pre_scale, post_scale;
  int rows_per_wave;
  uint32_t mid_stream, mid_thresh;
  float mid_scale;
};

// VEC = elements per lane chunk: 8 (16-byte accesses) in general; 4 (8-byte accesses) when that divides the row evenly over
// the 64 lanes -- C = 768 is 96 chunks of 8, i.e. 64 + 32 lanes (a quarter of the lane slots and of the registers idle), but
// exactly 3 chunks of 4 per lane.
// PGELU: the forward applied GELU after the affine (p.beta non-null; XLSR feature-encoder conv layers) -- a template flag so
// that the beta registers exist only in that instantiation; DROP: any dropout site active (the column multipliers stay in
// registers for all of a wave's rows).
// SPEC >= 0: a SPECIALISED instantiation for the encoder's hot shape -- the row divides evenly over the lanes (no per-chunk
// bounds branch) and which outputs / dropout sites exist is a compile-time bit set (1 dy, 2 pre-dropout, 4 sum-dropout,
// 8 post-dropout, 16 second gradient operand g2, 32 residual-stream gradient g_res) instead of ~80 scalar branches inside the
// row loop: the generic form (SPEC = -1) spent ~600 instructions per row and lane on 12 elements, this one about half.
template <typename T, int NCH, int VEC, bool PGELU, bool DROP, int SPEC = -1>
__global__ __launch_bounds__(ROW_THREADS) void ln_bwd_kernel(const LnBwdParams<T> p) {
  using VecT = ChunkT<T, VEC>;
  __shared__ float red[3][ROW_THREADS / 64][NCH * VEC][64];  // [dgamma|dbeta|dy sum][wave][slot][lane]
  constexpr bool S = SPEC >= 0;
  const bool has_dy = S ? (SPEC & 1) != 0 : p.dy != nullptr;
  const bool d_pre = S ? (SPEC & 2) != 0 : (DROP && p.pre_thresh != 0);
  const bool d_mid = S ? (SPEC & 4) != 0 : (DROP && p.mid_thresh != 0);
  const bool d_post = S ? (SPEC & 8) != 0 : (DROP && p.post_thresh != 0);
  const bool has_g2 = S ? (SPEC & 16) != 0 : p.g2 != nullptr;
  const bool has_gres = S ? (SPEC & 32) != 0 : p.g_res != nullptr;
  // (the residual-stream gradient is prefetched with the row's other operands unless a second gradient operand already is: all
  // four in flight per row cost the third wave per SIMD)
  constexpr bool PF_GRES = S && (SPEC & 32) != 0 && (SPEC & 16) == 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = p.C / VEC;
  const int wid = blockIdx.x * (ROW_THREADS / 64) + wave;
  float ag[NCH][VEC], ab[NCH][VEC], gm[NCH][VEC], ay[NCH][VEC], bt[PGELU ? NCH : 1][VEC];
  uint32_t cm[DROP ? NCH : 1][VEC];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      ag[i][k] = 0.f;
      ab[i][k] = 0.f;
      ay[i][k] = 0.f;
      gm[i][k] = (S || ch < nch) ? p.gamma[ch * VEC + k] : 0.f;
      if (PGELU) bt[i][k] = (ch < nch) ? p.beta[ch * VEC + k] : 0.f;
    }
    if (DROP && (S || ch < nch)) load_colmul<VEC>(cm[i], ch * VEC);
  }
  const uint32_t thi_pre = p.pre_thresh << 16, thi_mid = p.mid_thresh << 16, thi_post = p.post_thresh << 16;
  // raw operands of the row a wave works on are fetched one row ahead: a wave owns ~8 rows and every row is a dependent
  // chain load -> two wave reductions -> store, so without the prefetch the kernel ran at HBM latency, not bandwidth
  const int rstep = gridDim.x * (ROW_THREADS / 64);
  VecT ra[NCH], rb[NCH], rx[NCH], re[NCH];
  float mean_n = 0.f, rstd_n = 0.f;
  auto fetch = [&](int row) {
    mean_n = p.mean[row];
    rstd_n = p.rstd[row];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = lane + 64 * i;
      if (S || ch < nch) {
        const size_t o = (size_t)row * p.C + ch * VEC;
        ra[i] = *reinterpret_cast<const VecT*>(p.g1 + o);
        if (has_g2) rb[i] = *reinterpret_cast<const VecT*>(p.g2 + o);
        rx[i] = *reinterpret_cast<const VecT*>(p.r + o);
        if (PF_GRES) re[i] = *reinterpret_cast<const VecT*>(p.g_res + o);
      }
    }
  };
  if (wid < p.M) fetch(wid);
  for (int row = wid; row < p.M; row += rstep) {
    const float mean = mean_n, rstd = rstd_n;
    uint32_t rk_pre = 1u, rk_mid = 1u, rk_post = 1u;  // this row's dropout keys (common.h), one hash per active site
    if (DROP) {
      if (d_pre) rk_pre = drop_rowkey(p.seed, p.pre_stream, (uint64_t)row);
      if (d_mid) rk_mid = drop_rowkey(p.seed, p.mid_stream, (uint64_t)row);
      if (d_post) rk_post = drop_rowkey(p.seed, p.post_stream, (uint64_t)row);
    }
    float dyv[NCH][VEC], xh[NCH][VEC];
    float s1 = 0.f, s2 = 0.f;
    VecT ca[NCH], cb[NCH], cx[NCH], ce[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      ca[i] = ra[i];
      cb[i] = rb[i];
      cx[i] = rx[i];
      ce[i] = re[i];
    }
    if (row + rstep < p.M) fetch(row + rstep);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = lane + 64 * i;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        dyv[i][k] = 0.f;
        xh[i][k] = 0.f;
      }
      if (S || ch < nch) {
        float a[VEC], x[VEC];
        chunk_to_f(ca[i], a);
        if (has_g2) {
          float b2[VEC];
          chunk_to_f(cb[i], b2);
#pragma unroll
          for (int k = 0; k < VEC; ++k) a[k] += b2[k];
        }
        chunk_to_f(cx[i], x);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          if (d_post) a[k] = drop_keep(rk_post, cm[DROP ? i : 0][k], thi_post) ? a[k] * p.post_scale : 0.f;
          xh[i][k] = (x[k] - mean) * rstd;
          if (PGELU) a[k] *= gelu_grad_s<T>(fmaf(xh[i][k], gm[i][k], bt[PGELU ? i : 0][k]));
          ag[i][k] += a[k] * xh[i][k];
          ab[i][k] += a[k];
          dyv[i][k] = a[k] * gm[i][k];
          s1 += dyv[i][k];
          s2 += dyv[i][k] * xh[i][k];
        }
      }
    }
    s1 = wave_sum(s1) / (float)p.C;
    s2 = wave_sum(s2) / (float)p.C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = lane + 64 * i;
      if (S || ch < nch) {
        const size_t o = (size_t)row * p.C + ch * VEC;
        float d[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) d[k] = rstd * (dyv[i][k] - s1 - xh[i][k] * s2);
        if (has_gres) {
          float e[VEC];
          if (PF_GRES) chunk_to_f(ce[i], e);
          else chunk_to_f(*reinterpret_cast<const VecT*>(p.g_res + o), e);
#pragma unroll
          for (int k = 0; k < VEC; ++k) d[k] += e[k];
        }
        if (d_mid) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) d[k] = drop_keep(rk_mid, cm[DROP ? i : 0][k], thi_mid) ? d[k] * p.mid_scale : 0.f;
        }
        *reinterpret_cast<VecT*>(p.dr + o) = f_to_chunk<T, VEC>(d);
        if (has_dy) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            d[k] = (!d_pre || drop_keep(rk_pre, cm[DROP ? i : 0][k], thi_pre)) ? d[k] * p.pre_scale : 0.f;
            ay[i][k] += d[k];  // fp32 values, before the rounding of the store
          }
          *reinterpret_cast<VecT*>(p.dy + o) = f_to_chunk<T, VEC>(d);
        }
      }
    }
  }
  // reduce the per-wave column partials across the 4 waves, then one atomic per column per workgroup
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      red[0][wave][i * VEC + k][lane] = ag[i][k];
      red[1][wave][i * VEC + k][lane] = ab[i][k];
      red[2][wave][i * VEC + k][lane] = ay[i][k];
    }
  lds_barrier();  // (not __syncthreads(): the last rows' dr / dy stores are still in flight and nothing here reads them)
  for (int e = threadIdx.x; e < NCH * VEC * 64; e += ROW_THREADS) {
    const int slot = e >> 6, ln = e & 63;
    const int i = slot / VEC, k = slot % VEC;
    const int col = (ln + 64 * i) * VEC + k;
    if (col < p.C) {
      float sg = 0.f, sb = 0.f, sy = 0.f;
#pragma unroll
      for (int w = 0; w < ROW_THREADS / 64; ++w) {
        sg += red[0][w][slot][ln];
        sb += red[1][w][slot][ln];
        sy += red[2][w][slot][ln];
      }
      p.partial[((size_t)blockIdx.x * 3) * p.C + col] = sg;
      p.partial[((size_t)blockIdx.x * 3 + 1) * p.C + col] = sb;
      p.partial[((size_t)blockIdx.x * 3 + 2) * p.C + col] = sy;
    }
  }
}

__global__ __launch_bounds__(1024) void ln_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ dy_colsum) {
  // workgroup = 64 columns x 16 row groups; fixed summation order (deterministic), then one += per column
  __shared__ float red[16][65];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + cx;  // index into [3][C] (the third plane only when dy_colsum is given)
  const int planes = dy_colsum ? 3 : 2;
  float s = 0.f;
  if (e < planes * C) {
    const int which = e / C, col = e % C;
#pragma unroll 8
    for (int b = ry; b < nblk; b += 16) s += partial[((size_t)b * 3 + which) * C + col];
  }
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && e < planes * C) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cx];
    const int which = e / C, col = e % C;
    float* dst = which == 0 ? dgamma : which == 1 ? dbeta : dy_colsum;
    dst[col] += t;
  }
}

// All queued second stages in one launch.  Workgroup = 64 columns x 16 slot groups of one job (fixed summation order).
struct ReduceTable {
  ReduceJob jobs[ReduceSink::CAP];
  int first_block[ReduceSink::CAP + 1];
  int n;
};
__global__ __launch_bounds__(1024) void reduce_jobs_kernel(const ReduceTable t) {
  __shared__ float red[16][65];
  int j = 0;
  for (int k = 1; k < t.n; ++k) j = (int)blockIdx.x >= t.first_block[k] ? k : j;
  const ReduceJob& q = t.jobs[j];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int n = ((int)blockIdx.x - t.first_block[j]) * 64 + cx;
  float s = 0.f;
  if (n < q.ncols) {
    // sixteen independent loads in flight per thread (the 768 partial rows of a LayerNorm backward are 48 per thread: with four in
    // flight the launch was twelve dependent round trips to HBM, ~14 us for 7 MB); the additions stay in slot order
    int k = ry;
    for (; k + 15 * 16 < q.slots; k += 16 * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = q.partial[(long)(k + 16 * u) * q.stride + n];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; k < q.slots; k += 16) s += q.partial[(long)k * q.stride + n];
  }
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && n < q.ncols) {
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) v += red[r][cx];
    q.out[n] += v;
  }
}

// ---------------------------------------------------------------------------------------------- softmax
// Scores arrive in bf16 (what the bf16 QK^T GEMM produces); statistics are fp32.  One wave per row; a lane owns
// 16-byte chunks (8 consecutive keys) so every access is a full-width vector load/store.
template <typename T>
struct SoftmaxParams {
  const T* S;           // [rows, ld] scores (already scaled)
  T* P;                 // [rows, ld] probabilities (pre-dropout); pad columns [cols, ld) are written as 0
  T* Pd;                // [rows, ld] dropped probabilities or null
  const int32_t* klens; // [B] valid keys per utterance or null
  int rows, cols, ld, rows_per_batch;
  uint64_t seed;
  uint32_t stream, thresh;
  float scale;
};

template <typename T, int NCH>
__global__ __launch_bounds__(ROW_THREADS) void softmax_fwd_kernel(const SoftmaxParams<T> p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (ROW_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int kl = p.klens ? min(p.klens[row / p.rows_per_batch], p.cols) : p.cols;
  const int nch = p.ld >> 3;
  float v[NCH][8];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[i][k] = -INFINITY;
    if (ch < nch) {
      float f[8];
      chunk_to_f(ld8<T>(p.S + (size_t)row * p.ld + ch * 8), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[i][k] = (ch * 8 + k < kl) ? f[k] : -INFINITY;
        mx = fmaxf(mx, v[i][k]);
      }
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[i][k] = (v[i][k] == -INFINITY) ? 0.f : __expf(v[i][k] - mx);
      sum += v[i][k];
    }
  const float inv = 1.f / wave_sum(sum);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      const size_t o = (size_t)row * p.ld + ch * 8;
      float pr[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) pr[k] = v[i][k] * inv;
      st8<T>(p.P + o, f_to_chunk8<T>(pr));
      if (p.Pd) {
        // the fused attention kernels' bits (common.h attn_keep_bit): row = (b * nh + h) * F + q, key = column
        const uint32_t rs = drop_rowkey(p.seed, p.stream, (uint64_t)row);
#pragma unroll
        for (int k = 0; k < 8; ++k) pr[k] = (!p.thresh || drop_keep(rs, drop_colmul((uint32_t)(ch * 8 + k)), p.thresh << 16)) ? pr[k] * p.scale : 0.f;
        st8<T>(p.Pd + o, f_to_chunk8<T>(pr));
      }
    }
  }
}

template <typename T>
struct SoftmaxBwdParams {
  const T* dPd;      // [rows, ld] grad wrt dropped probabilities
  const T* P;        // [rows, ld]
  T* dS;             // [rows, ld]; pad columns written as 0
  int rows, cols, ld;
  uint64_t seed;
  uint32_t stream, thresh;
  float scale;
};

template <typename T, int NCH>
__global__ __launch_bounds__(ROW_THREADS) void softmax_bwd_kernel(const SoftmaxBwdParams<T> p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (ROW_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int nch = p.ld >> 3;
  float pr[NCH][8], dp[NCH][8];
  float dot = 0.f;
  const uint32_t rs = drop_rowkey(p.seed, p.stream, (uint64_t)row);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      pr[i][k] = 0.f;
      dp[i][k] = 0.f;
    }
    if (ch < nch) {
      const size_t o = (size_t)row * p.ld + ch * 8;
      float g[8];
      chunk_to_f(ld8<T>(p.P + o), pr[i]);
      chunk_to_f(ld8<T>(p.dPd + o), g);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool in = ch * 8 + k < p.cols;
        pr[i][k] = in ? pr[i][k] : 0.f;
        dp[i][k] = (in && (!p.thresh || drop_keep(rs, drop_colmul((uint32_t)(ch * 8 + k)), p.thresh << 16))) ? g[k] * p.scale : 0.f;
        dot += pr[i][k] * dp[i][k];
      }
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      float d[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = pr[i][k] * (dp[i][k] - dot);
      st8<T>(p.dS + (size_t)row * p.ld + ch * 8, f_to_chunk8<T>(d));
    }
  }
}

// ---------------------------------------------------------------------------------------------- column sums
// out[n] += sum_m X[m, n]  (bias gradients).  Workgroup: 8 threads across 64 columns x 32 row lanes; gridDim.y row groups.
// With a `partial` buffer [gridDim.y][N] every group writes its own row (summed afterwards in a fixed order: deterministic);
// without one the groups add into `out` atomically.  rowmask (optional): only rows with rowmask[m] != 0 that are not
// padding (frame < flens[m / F]) count -- the gradient of the SpecAugment mask embedding.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, long ld, int M, int N, float* __restrict__ out,
                                                     float* __restrict__ partial, const uint8_t* __restrict__ rowmask,
                                                     const int32_t* __restrict__ flens, int F) {
  __shared__ float red[32][65];
  const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cx * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
    for (int m = blockIdx.y * 32 + ry; m < M; m += gridDim.y * 32) {
      if (rowmask && (!rowmask[m] || (flens && m % F >= flens[m / F]))) continue;
      float f[8];
      chunk_to_f(ld8<T>(X + (size_t)m * ld + c0), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += f[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[ry][cx * 8 + k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += red[r][threadIdx.x];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < N) {
      if (partial)
        partial[(size_t)blockIdx.y * N + c] = s;
      else
        atomicAdd(out + c, s);
    }
  }
}
// out[n] += sum over `slots` rows of partial [slots][N], fixed order; workgroup = 64 columns x 16 slot groups
__global__ __launch_bounds__(1024) void colsum_rows_kernel(const float* __restrict__ partial, int slots, int N, float* __restrict__ out) {
  __shared__ float red[16][65];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cx;
  float s = 0.f;
  if (n < N)
    for (int k = ry; k < slots; k += 16) s += partial[(long)k * N + n];
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cx];
    out[n] += t;
  }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ in, bf16* __restrict__ out, long n) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 q = *reinterpret_cast<const float4*>(in + i);
    const bf16x4 t = {(bf16)q.x, (bf16)q.y, (bf16)q.z, (bf16)q.w};
    *reinterpret_cast<bf16x4*>(out + i) = t;
  } else {
    for (long j = i; j < n; ++j) out[j] = (bf16)in[j];
  }
}

// SpecAugment + padding: h[row,:] = embed where mask[row]; = 0 where frame >= flen[b]
template <typename T>
__global__ void specaug_fwd_kernel(T* __restrict__ h, const uint8_t* __restrict__ mask,
                                   const int32_t* __restrict__ flens, const float* __restrict__ embed, int M, int F, int C) {
  const int row = blockIdx.x;
  const int b = row / F, t = row % F;
  const bool pad = flens && t >= flens[b];
  const bool mk = mask && mask[row];
  if (!pad && !mk) return;
  for (int c = threadIdx.x; c < C; c += blockDim.x) h[(size_t)row * C + c] = pad ? (T)0.f : (T)embed[c];
}
// backward: rows that were overwritten pass no gradient; masked rows feed d embed
template <typename T>
__global__ void specaug_bwd_kernel(T* __restrict__ dh, const uint8_t* __restrict__ mask,
                                   const int32_t* __restrict__ flens, float* __restrict__ dembed, int M, int F, int C) {
  const int row = blockIdx.x;
  const int b = row / F, t = row % F;
  const bool pad = flens && t >= flens[b];
  const bool mk = mask && mask[row];
  if (!pad && !mk) return;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    dh[(size_t)row * C + c] = (T)0.f;  // (the masked rows' sum -> d embed is taken before, by the masked column sum)
  }
}

template <typename T>
__global__ void gelu_grad_mul_kernel(const T* __restrict__ dy, const T* __restrict__ pre, T* __restrict__ out, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float a[8], x[8];
    chunk_to_f(ld8<T>(dy + 8 * i), a);
    chunk_to_f(ld8<T>(pre + 8 * i), x);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] *= gelu_grad_s<T>(x[k]);
    st8<T>(out + 8 * i, f_to_chunk8<T>(a));
  }
}
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float x[8], y[8];
    chunk_to_f(ld8<T>(a + 8 * i), x);
    chunk_to_f(ld8<T>(b + 8 * i), y);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] += y[k];
    st8<T>(out + 8 * i, f_to_chunk8<T>(x));
  }
}

// 16-bit dropout threshold and the scale of the probability it realises
uint32_t thresh_of(float p) { return p <= 0.f ? 0u : (uint32_t)fminf(65535.f, roundf(p * 65536.f)); }
float scale_of(float p) { return p <= 0.f ? 1.f : 1.f / (1.f - (float)thresh_of(p) / 65536.f); }
// the sites of one launch share the step's seed; take it from whichever site is active (round 3: it used to be read from the
// pre-dropout site alone, so a launch with only a post- or sum-dropout -- the encoder-input site -- ran on seed 0: the same
// mask every step.  Found by the regularisers-on goldens, tests/test_gpu_dropout.py)
uint64_t seed_of(const DropSpec& pre, const DropSpec& post, const DropSpec& mid) {
  return pre.p > 0.f ? pre.seed : post.p > 0.f ? post.seed : mid.seed;
}

}  // namespace

// ---- internal C++ entry points used by the engine (declared in kernels.h) ------------------------

template <typename T>
int k_layernorm_fwd_t(const T* y, const T* res, const float* gamma, const float* beta, T* r_out, T* out,
                      float* mean, float* rstd, int M, int C, float eps, const DropSpec& pre, const DropSpec& post,
                      hipStream_t st, const DropSpec& mid, bool post_gelu) {
  SSAK_REQUIRE(M > 0 && C > 0 && (C & 7) == 0 && C <= 1536, "layernorm: C=%d must be a multiple of 8 and <= 1536", C);
  LnFwdParams<T> p{y, res, gamma, beta, r_out, out, mean, rstd, M, C, eps, seed_of(pre, post, mid),
                pre.stream, thresh_of(pre.p), post.stream, thresh_of(post.p),
                scale_of(pre.p), scale_of(post.p),
                mid.stream, thresh_of(mid.p), scale_of(mid.p), post_gelu ? 1 : 0};
  const int grid = ssak_cdiv(M, ROW_THREADS / 64);
  const int nch = ssak_cdiv(C / 8, 64);
  ProfScope prof_scope(PROF_LN_FWD, (double)M * C * 2.0 * ((y != nullptr) + (res != nullptr) + (r_out != nullptr) + (out != nullptr)), st);
  // the encoder's hot shapes, bf16 engine, LN output wanted, no post-GELU: specialised multi-row instantiations by (y, res, pre-,
  // sum-, post-dropout, r_out) -- the combinations the engine's forward launches
  if constexpr (std::is_same<T, bf16>::value) {
    if (out && gamma && beta && !post_gelu && (C == 768 || C == 1024)) {
      const int spec = (y ? 1 : 0) | (res ? 2 : 0) | (p.pre_thresh ? 4 : 0) | (p.mid_thresh ? 8 : 0) | (p.post_thresh ? 16 : 0) | (r_out ? 32 : 0);
      const int rgrid = std::min(LN_FWD_BLOCKS, grid);
      bool done = true;
#define LN_FWD_SPEC(SP)                                                                          \
  case SP:                                                                                       \
    if (C == 768) ln_fwd_rows_kernel<T, 3, 4, SP><<<rgrid, ROW_THREADS, 0, st>>>(p);             \
    else ln_fwd_rows_kernel<T, 2, 8, SP><<<rgrid, ROW_THREADS, 0, st>>>(p);                      \
    break;
      switch (spec) {
        LN_FWD_SPEC(1) LN_FWD_SPEC(2)                      // a plain LayerNorm of one operand
        LN_FWD_SPEC(35) LN_FWD_SPEC(39)                    // r = res + [drop] y, LN(r): the encoder layers' two LayerNorms
        LN_FWD_SPEC(17) LN_FWD_SPEC(18)                    // encoder-input dropout behind the LN
        LN_FWD_SPEC(43) LN_FWD_SPEC(47) LN_FWD_SPEC(42)    // stable-LN: dropout on the sum
        default: done = false;
      }
#undef LN_FWD_SPEC
      if (done) {
        SSAK_LAUNCH_CHECK();
        return SSAK_OK;
      }
    }
  }
  if (nch == 1)
    ln_fwd_kernel<T, 1><<<grid, ROW_THREADS, 0, st>>>(p);
  else if (nch == 2)
    ln_fwd_kernel<T, 2><<<grid, ROW_THREADS, 0, st>>>(p);
  else
    ln_fwd_kernel<T, 3><<<grid, ROW_THREADS, 0, st>>>(p);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <typename T>
int k_layernorm_bwd_t(const T* g1, const T* g2, const T* r, const float* mean, const float* rstd,
                      const float* gamma, const T* g_res, T* dr, T* dy, float* dgamma, float* dbeta,
                      float* partial, int M, int C, const DropSpec& pre, const DropSpec& post, hipStream_t st,
                      const DropSpec& mid, float* dy_colsum, const float* post_gelu_beta) {
  SSAK_REQUIRE(M > 0 && C > 0 && (C & 7) == 0 && C <= 1536, "layernorm_bwd: C=%d must be a multiple of 8 and <= 1536", C);
  SSAK_REQUIRE(!dy_colsum || dy, "layernorm_bwd: the dy column sum needs the dy output");
  LnBwdParams<T> p{g1, g2, r, mean, rstd, gamma, g_res, dr, dy, dgamma, dbeta, partial, dy_colsum, post_gelu_beta, M, C, seed_of(pre, post, mid),
                pre.stream, thresh_of(pre.p), post.stream, thresh_of(post.p),
                scale_of(pre.p), scale_of(post.p), 1,
                mid.stream, thresh_of(mid.p), scale_of(mid.p)};
  // LN_BWD_BLOCKS workgroups of 4 waves stride over the rows and keep column partials in registers
  p.rows_per_wave = 0;
  const int grid = std::min(LN_BWD_BLOCKS, ssak_cdiv(M, ROW_THREADS / 64));
  const int nch = ssak_cdiv(C / 8, 64);
  ProfScope prof_scope(PROF_LN_BWD, (double)M * C * 2.0 * (2 + (g2 != nullptr) + (g_res != nullptr) + 1 + (dy != nullptr && dy != dr)), st);
  const bool drop = (p.pre_thresh | p.post_thresh | p.mid_thresh) != 0;
#define LN_BWD_LAUNCH(N_, V_)                                                                              \
  do {                                                                                                     \
    if (post_gelu_beta) ln_bwd_kernel<T, N_, V_, true, true><<<grid, ROW_THREADS, 0, st>>>(p);             \
    else if (drop) ln_bwd_kernel<T, N_, V_, false, true><<<grid, ROW_THREADS, 0, st>>>(p);                 \
    else ln_bwd_kernel<T, N_, V_, false, false><<<grid, ROW_THREADS, 0, st>>>(p);                          \
  } while (0)
  // the encoder's hot shapes (H = 768: three chunks of 4 per lane; H = 1024: two chunks of 8), bf16 engine, no post-GELU: specialised
  // instantiations by (dy, pre-, sum-, post-dropout, g2, g_res) -- every combination the engine's backward launches
  if constexpr (std::is_same<T, bf16>::value) {
    if (!post_gelu_beta && (C == 768 || C == 1024)) {
      const int spec = (dy ? 1 : 0) | (p.pre_thresh ? 2 : 0) | (p.mid_thresh ? 4 : 0) | (p.post_thresh ? 8 : 0) | (g2 ? 16 : 0) | (g_res ? 32 : 0);
      bool done = true;
#define LN_BWD_SPEC(SP)                                                                                              \
  case SP:                                                                                                           \
    if (C == 768) ln_bwd_kernel<T, 3, 4, false, ((SP) & 14) != 0, SP><<<grid, ROW_THREADS, 0, st>>>(p);              \
    else ln_bwd_kernel<T, 2, 8, false, ((SP) & 14) != 0, SP><<<grid, ROW_THREADS, 0, st>>>(p);                       \
    break;
      switch (spec) {
        LN_BWD_SPEC(0) LN_BWD_SPEC(16) LN_BWD_SPEC(32) LN_BWD_SPEC(48)        // plain / g2 / g_res / both (no dy, no dropout)
        LN_BWD_SPEC(1) LN_BWD_SPEC(17) LN_BWD_SPEC(33) LN_BWD_SPEC(49)        // + dy (hidden dropout off)
        LN_BWD_SPEC(3) LN_BWD_SPEC(19) LN_BWD_SPEC(35) LN_BWD_SPEC(51)        // + dy with the hidden dropout
        LN_BWD_SPEC(8) LN_BWD_SPEC(24)                                        // encoder-input dropout behind the LN (post)
        LN_BWD_SPEC(36) LN_BWD_SPEC(52)                                       // encoder-input dropout on the sum (stable-LN)
        default: done = false;
      }
#undef LN_BWD_SPEC
      if (done) {
        SSAK_LAUNCH_CHECK();
        goto launched;
      }
    }
  }
  if (C % 256 == 0 && C / 256 == 3 && (C / 8) % 64 != 0)  // (768: three chunks of 4 per lane, every lane busy)
    LN_BWD_LAUNCH(3, 4);
  else if (nch == 1)
    LN_BWD_LAUNCH(1, 8);
  else if (nch == 2)
    LN_BWD_LAUNCH(2, 8);
  else
    LN_BWD_LAUNCH(3, 8);
#undef LN_BWD_LAUNCH
  SSAK_LAUNCH_CHECK();
launched:
  if (g_reduce_sink && g_reduce_sink->n + 3 <= ReduceSink::CAP) {  // second stage queued: one launch for many (kernels.h)
    g_reduce_sink->push(partial, 3L * C, grid, C, dgamma);
    g_reduce_sink->push(partial + C, 3L * C, grid, C, dbeta);
    if (dy_colsum) g_reduce_sink->push(partial + 2 * C, 3L * C, grid, C, dy_colsum);
    return SSAK_OK;
  }
  ln_bwd_finalize_kernel<<<ssak_cdiv((dy_colsum ? 3 : 2) * C, 64), 1024, 0, st>>>(partial, grid, C, dgamma, dbeta, dy_colsum);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <typename T>
int k_softmax_fwd_t(const T* S, T* P, T* Pd, const int32_t* klens, int rows, int cols, int ld,
                    int rows_per_batch, const DropSpec& drop, hipStream_t st) {
  SSAK_REQUIRE(rows > 0 && cols > 0 && ld >= cols && (ld & 7) == 0 && ld <= 1536, "softmax: cols=%d ld=%d unsupported (ld %% 8 == 0, <= 1536)", cols, ld);
  SoftmaxParams<T> p{S, P, Pd, klens, rows, cols, ld, rows_per_batch, drop.seed, drop.stream, thresh_of(drop.p),
                  scale_of(drop.p)};
  const int grid = ssak_cdiv(rows, ROW_THREADS / 64);
  const int nch = ssak_cdiv(ld / 8, 64);
  ProfScope prof_scope(PROF_SOFTMAX, (double)rows * ld * 2.0 * (2 + (Pd != nullptr)), st);
  if (nch == 1)
    softmax_fwd_kernel<T, 1><<<grid, ROW_THREADS, 0, st>>>(p);
  else if (nch == 2)
    softmax_fwd_kernel<T, 2><<<grid, ROW_THREADS, 0, st>>>(p);
  else
    softmax_fwd_kernel<T, 3><<<grid, ROW_THREADS, 0, st>>>(p);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <typename T>
int k_softmax_bwd_t(const T* dPd, const T* P, T* dS, int rows, int cols, int ld, const DropSpec& drop,
                    hipStream_t st) {
  SSAK_REQUIRE(rows > 0 && cols > 0 && ld >= cols && (ld & 7) == 0 && ld <= 1536, "softmax_bwd: cols=%d ld=%d unsupported", cols, ld);
  SoftmaxBwdParams<T> p{dPd, P, dS, rows, cols, ld, drop.seed, drop.stream, thresh_of(drop.p),
                     scale_of(drop.p)};
  const int grid = ssak_cdiv(rows, ROW_THREADS / 64);
  const int nch = ssak_cdiv(ld / 8, 64);
  ProfScope prof_scope(PROF_SOFTMAX, (double)rows * ld * 2.0 * 3, st);
  if (nch == 1)
    softmax_bwd_kernel<T, 1><<<grid, ROW_THREADS, 0, st>>>(p);
  else if (nch == 2)
    softmax_bwd_kernel<T, 2><<<grid, ROW_THREADS, 0, st>>>(p);
  else
    softmax_bwd_kernel<T, 3><<<grid, ROW_THREADS, 0, st>>>(p);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <typename T>
int k_colsum_t(const T* X, long ld, int M, int N, float* out, hipStream_t st, float* scratch, size_t scratch_floats,
               const uint8_t* rowmask, const int32_t* flens, int F) {
  SSAK_REQUIRE(M > 0 && N > 0 && (N & 7) == 0 && (ld & 7) == 0, "colsum: N=%d ld=%ld must be multiples of 8", N, ld);
  dim3 grid(ssak_cdiv(N, 64), min(64, ssak_cdiv(M, 32)));
  ProfScope prof_scope(PROF_ROWWISE, (double)M * N * sizeof(T), st);
  const bool det = scratch && scratch_floats >= (size_t)grid.y * N;  // deterministic two-stage sum when a scratch is given
  colsum_kernel<T><<<grid, 256, 0, st>>>(X, ld, M, N, out, det ? scratch : nullptr, rowmask, flens, F > 0 ? F : 1);
  SSAK_LAUNCH_CHECK();
  if (det) {
    if (g_reduce_sink && g_reduce_sink->push(scratch, N, (int)grid.y, N, out)) return SSAK_OK;
    colsum_rows_kernel<<<ssak_cdiv(N, 64), 1024, 0, st>>>(scratch, (int)grid.y, N, out);
    SSAK_LAUNCH_CHECK();
  }
  return SSAK_OK;
}

thread_local ReduceSink* g_reduce_sink = nullptr;

int k_colsum_rows(const float* partial, int slots, int N, float* out, hipStream_t st) {
  ProfScope prof_scope(PROF_ROWWISE, (double)slots * N * 4.0, st);
  colsum_rows_kernel<<<ssak_cdiv(N, 64), 1024, 0, st>>>(partial, slots, N, out);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

int k_reduce_flush(ReduceSink& sink, hipStream_t st) {
  int done = 0;
  double bytes = 0.0;
  for (int i = 0; i < sink.n; ++i) bytes += (double)sink.jobs[i].slots * sink.jobs[i].ncols * 4.0;
  ProfScope prof_scope(PROF_ROWWISE, bytes, st);
  while (done < sink.n) {
    // jobs that add into the same vector must not share a launch (plain +=): cut the batch at the first repeat
    ReduceTable t;
    t.n = 0;
    int blocks = 0;
    for (int i = done; i < sink.n; ++i) {
      bool repeat = false;
      for (int k = 0; k < t.n; ++k) repeat |= t.jobs[k].out == sink.jobs[i].out;
      if (repeat) break;
      t.first_block[t.n] = blocks;
      t.jobs[t.n++] = sink.jobs[i];
      blocks += ssak_cdiv(sink.jobs[i].ncols, 64);
    }
    t.first_block[t.n] = blocks;
    reduce_jobs_kernel<<<blocks, 1024, 0, st>>>(t);
    SSAK_LAUNCH_CHECK();
    done += t.n;
  }
  sink.n = 0;
  return SSAK_OK;
}

int k_cast_f32_bf16(const float* in, bf16* out, long n, hipStream_t st) {
  if (n <= 0) return SSAK_OK;
  ProfScope prof_scope(PROF_ROWWISE, (double)n * 6.0, st);
  cast_f32_bf16_kernel<<<ssak_cdiv(ssak_cdiv(n, 4), 256), 256, 0, st>>>(in, out, n);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <typename T>
int k_specaug_fwd_t(T* h, const uint8_t* mask, const int32_t* flens, const float* embed, int B, int F, int C,
                    hipStream_t st) {
  if (!mask && !flens) return SSAK_OK;
  ProfScope prof_scope(PROF_ROWWISE, (double)B * F * 1.0, st);  // the mask is read; ~5 % of the rows are rewritten
  specaug_fwd_kernel<T><<<B * F, 128, 0, st>>>(h, mask, flens, embed, B * F, F, C);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <typename T>
int k_specaug_bwd_t(T* dh, const uint8_t* mask, const int32_t* flens, float* dembed, int B, int F, int C,
                    hipStream_t st, float* scratch, size_t scratch_floats) {
  if (!mask && !flens) return SSAK_OK;
  // d embed = sum of the masked (non-padding) rows of dh, taken before they are zeroed
  if (mask && dembed)
    if (int rc = k_colsum_t<T>(dh, C, B * F, C, dembed, st, scratch, scratch_floats, mask, flens, F)) return rc;
  specaug_bwd_kernel<T><<<B * F, 128, 0, st>>>(dh, mask, flens, dembed, B * F, F, C);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <typename T>
int k_gelu_grad_mul_t(const T* dy, const T* pre, T* out, long n, hipStream_t st) {
  SSAK_REQUIRE((n & 7) == 0, "gelu_grad_mul: n must be a multiple of 8");
  ProfScope prof_scope(PROF_ROWWISE, (double)n * 6.0, st);
  gelu_grad_mul_kernel<T><<<min(4096, ssak_cdiv(n / 8, 256)), 256, 0, st>>>(dy, pre, out, n / 8);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

template <typename T>
int k_add_t(const T* a, const T* b, T* out, long n, hipStream_t st) {
  SSAK_REQUIRE((n & 7) == 0, "add: n must be a multiple of 8");
  ProfScope prof_scope(PROF_ROWWISE, (double)n * 6.0, st);
  add_kernel<T><<<min(4096, ssak_cdiv(n / 8, 256)), 256, 0, st>>>(a, b, out, n / 8);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

// ---- explicit instantiations (bf16: the production engine; float: the fp32-exact verification mode) and the bf16-named
// entry points the other kernel files call
#define SSAK_INSTANTIATE_ROW_KERNELS(T)                                                                                           \
  template int k_layernorm_fwd_t<T>(const T*, const T*, const float*, const float*, T*, T*, float*, float*, int, int, float,      \
                                    const DropSpec&, const DropSpec&, hipStream_t, const DropSpec&, bool);                        \
  template int k_layernorm_bwd_t<T>(const T*, const T*, const T*, const float*, const float*, const float*, const T*, T*, T*,     \
                                    float*, float*, float*, int, int, const DropSpec&, const DropSpec&, hipStream_t,              \
                                    const DropSpec&, float*, const float*);                                                       \
  template int k_softmax_fwd_t<T>(const T*, T*, T*, const int32_t*, int, int, int, int, const DropSpec&, hipStream_t);            \
  template int k_softmax_bwd_t<T>(const T*, const T*, T*, int, int, int, const DropSpec&, hipStream_t);                           \
  template int k_colsum_t<T>(const T*, long, int, int, float*, hipStream_t, float*, size_t, const uint8_t*, const int32_t*, int); \
  template int k_specaug_fwd_t<T>(T*, const uint8_t*, const int32_t*, const float*, int, int, int, hipStream_t);                  \
  template int k_specaug_bwd_t<T>(T*, const uint8_t*, const int32_t*, float*, int, int, int, hipStream_t, float*, size_t);        \
  template int k_gelu_grad_mul_t<T>(const T*, const T*, T*, long, hipStream_t);                                                   \
  template int k_add_t<T>(const T*, const T*, T*, long, hipStream_t);
SSAK_INSTANTIATE_ROW_KERNELS(bf16)
SSAK_INSTANTIATE_ROW_KERNELS(float)

int k_layernorm_fwd(const bf16* y, const bf16* res, const float* gamma, const float* beta, bf16* r_out, bf16* out,
                    float* mean, float* rstd, int M, int C, float eps, const DropSpec& pre, const DropSpec& post,
                    hipStream_t st, const DropSpec& mid, bool post_gelu) {
  return k_layernorm_fwd_t<bf16>(y, res, gamma, beta, r_out, out, mean, rstd, M, C, eps, pre, post, st, mid, post_gelu);
}
int k_layernorm_bwd(const bf16* g1, const bf16* g2, const bf16* r, const float* mean, const float* rstd,
                    const float* gamma, const bf16* g_res, bf16* dr, bf16* dy, float* dgamma, float* dbeta,
                    float* partial, int M, int C, const DropSpec& pre, const DropSpec& post, hipStream_t st,
                    const DropSpec& mid, float* dy_colsum, const float* post_gelu_beta) {
  return k_layernorm_bwd_t<bf16>(g1, g2, r, mean, rstd, gamma, g_res, dr, dy, dgamma, dbeta, partial, M, C, pre, post, st, mid,
                                 dy_colsum, post_gelu_beta);
}
int k_colsum(const bf16* X, long ld, int M, int N, float* out, hipStream_t st, float* scratch, size_t scratch_floats,
             const uint8_t* rowmask, const int32_t* flens, int F) {
  return k_colsum_t<bf16>(X, ld, M, N, out, st, scratch, scratch_floats, rowmask, flens, F);
}

// ---- debug: the dropout bits of one site, written out (tests/test_gpu_dropout.py pins oracle/dropout_hash.py against them).
// The product kernels inline the same device functions (common.h drop_rowkey / drop_colmul / drop_keep); nothing on the hot path
// calls these.  Element-wise sites: (row, col) of their [rows, cols] tensor; attention: row = (b * nh + h) * F + q, col = key --
// one definition since round 5, so the attention entry is the element-wise one on [B * nh * F, F].
__global__ void debug_dropout_mask_kernel(uint64_t seed, uint32_t stream, uint32_t thresh, long rows, int cols, uint8_t* keep) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  keep[i] = (!thresh || keep_bit(seed, stream, (uint64_t)(i / cols), (uint32_t)(i % cols), thresh)) ? 1 : 0;
}
extern "C" int ssak_debug_dropout_mask(uint64_t seed, uint32_t site, float p, long rows, int cols, uint8_t* keep, float* scale_out /*host*/,
                                       void* stream) {
  SSAK_REQUIRE(keep && rows > 0 && cols > 0, "debug_dropout_mask: bad arguments");
  const long n = rows * cols;
  debug_dropout_mask_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(seed, site, thresh_of(p), rows, cols, keep);
  SSAK_LAUNCH_CHECK();
  if (scale_out) *scale_out = scale_of(p);
  return SSAK_OK;
}
extern "C" int ssak_debug_attention_dropout_mask(uint64_t seed, uint32_t site, float p, int B, int nh, int F, uint8_t* keep,
                                                 void* stream) {
  SSAK_REQUIRE(keep && B > 0 && nh > 0 && F > 0, "debug_attention_dropout_mask: bad arguments");
  return ssak_debug_dropout_mask(seed, site, p, (long)B * nh * F, F, keep, nullptr, stream);
}
