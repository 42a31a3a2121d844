// a14: data-movement kernels of the Whisper encoder front end (conv1 k3 p1 + GELU, conv2 k3 s2 p1 + GELU,
// + sinusoidal positions; transformers modeling_whisper.py:618-624) and of its backward.  gfx950.
//
// The two convolutions themselves are GEMMs on overlapping rows of zero-padded channels-last buffers (the same
// Toeplitz operand as the wav2vec2 conv stack): buffers hold one utterance per RS1 = 2*RS2 rows so that the
// stride-2 window of output row kk = b*RS2 + t starts at input row 2*kk for EVERY utterance, which lets the weight
// gradient run as one long-K GEMM over the whole batch.  What is left for this file is HBM-bound reshaping:
// transposing the mel features into channels-last, adding the position table, scattering the input gradient of
// the strided conv back (col2im, fused with GELU') and undoing the [Co][k][Ci] weight layout on the gradient.
#include "common.h"
#include "kernels.h"

namespace {

// mel [B, C, T] fp32 -> cl [B*RS + ...][C] bf16 at row b*RS + lead + t  (32x32 LDS transpose tiles)
template <typename T_>
__global__ __launch_bounds__(256) void mel_to_cl_kernel(const float* __restrict__ mel, T_* __restrict__ cl, int C, int T,
                                                        int RS, int lead) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, t = t0 + tx;
    tile[r][tx] = (c < C && t < T) ? mel[((long)b * C + c) * T + t] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, c = c0 + tx;
    if (t < T && c < C) cl[((long)b * RS + lead + t) * C + c] = (T_)tile[tx][r];
  }
}

template <typename T_>
__global__ void add_rowvec_kernel(const T_* __restrict__ x, const T_* __restrict__ pos, T_* __restrict__ out, int F,
                                  int H, long n8) {
  const int hc = H >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long row = i / hc;
    const int c = (int)(i % hc);
    const int t = (int)(row % F);
    float a[8], p[8];
    chunk_to_f(ld8<T_>(x + 8 * i), a);
    chunk_to_f(ld8<T_>(pos + ((long)t * hc + c) * 8), p);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += p[k];
    st8<T_>(out + 8 * i, f_to_chunk8<T_>(a));
  }
}

// src [B, F, H] dense -> dst [B, RS, H] (rows >= F zero)
template <typename T_>
__global__ void copy_rows_padded_kernel(const T_* __restrict__ src, T_* __restrict__ dst, int F, int RS, int H, long n8) {
  const int hc = H >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long row = i / hc;
    const int c = (int)(i % hc);
    const long b = row / RS;
    const int t = (int)(row % RS);
    float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    Chunk8<T_> v = f_to_chunk8<T_>(z);
    if (t < F) v = ld8<T_>(src + (((long)b * F + t) * hc + c) * 8);
    st8<T_>(dst + 8 * i, v);
  }
}

// input gradient of Conv1d(k=3, stride=2, pad=1) from the column form dxcol [B, F, 3, H] (tap-major), times GELU'(pre):
//   dx[u] = dxcol[u/2][1]                                  (u even)
//         = dxcol[(u+1)/2][0] (if (u+1)/2 < F) + dxcol[(u-1)/2][2]   (u odd)
// written at row b*RS1 + u of out (rows >= Tin zero); pre is stored with a one-row lead (row b*RS1 + 1 + u).
template <typename T_>
__global__ void col2im_k3s2_kernel(const T_* __restrict__ dxcol, const T_* __restrict__ pre, T_* __restrict__ out, int F,
                                   int Tin, int RS1, int H, long n8) {
  const int hc = H >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long row = i / hc;
    const int c = (int)(i % hc);
    const long b = row / RS1;
    const int u = (int)(row % RS1);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (u < Tin) {
      auto add = [&](int t, int tap) {
        float q[8];
        chunk_to_f(ld8<T_>(dxcol + ((((long)b * F + t) * 3 + tap) * hc + c) * 8), q);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += q[k];
      };
      if ((u & 1) == 0) {
        add(u >> 1, 1);
      } else {
        if (((u + 1) >> 1) < F) add((u + 1) >> 1, 0);
        add((u - 1) >> 1, 2);
      }
      float pq[8];
      chunk_to_f(ld8<T_>(pre + (((long)b * RS1 + 1 + u) * hc + c) * 8), pq);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] *= gelu_grad_s<T_>(pq[k]);
    }
    st8<T_>(out + 8 * i, f_to_chunk8<T_>(acc));
  }
}

// grads[co][ci][k] += dwr[co][k][ci]
__global__ void conv_wgrad_unrearrange_kernel(const float* __restrict__ dwr, float* __restrict__ g, int Co, int Ci, int k) {
  const long n = (long)Co * Ci * k;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int kk = (int)(e % k);
    const int ci = (int)((e / k) % Ci);
    const int co = (int)(e / ((long)k * Ci));
    g[e] += dwr[((long)co * k + kk) * Ci + ci];
  }
}

}  // namespace

template <typename T_>
int k_mel_to_cl_t(const float* mel, T_* cl, int B, int C, int T, int RS, int lead, hipStream_t st) {
  mel_to_cl_kernel<T_><<<dim3(ssak_cdiv(T, 32), ssak_cdiv(C, 32), B), 256, 0, st>>>(mel, cl, C, T, RS, lead);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
template <typename T_>
int k_add_rowvec_t(const T_* x, const T_* pos, T_* out, int B, int F, int H, hipStream_t st) {
  const long n8 = (long)B * F * H / 8;
  add_rowvec_kernel<T_><<<min(4096, ssak_cdiv(n8, 256)), 256, 0, st>>>(x, pos, out, F, H, n8);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
template <typename T_>
int k_copy_rows_padded_t(const T_* src, T_* dst, int B, int F, int RS, int H, hipStream_t st) {
  const long n8 = (long)B * RS * H / 8;
  copy_rows_padded_kernel<T_><<<min(4096, ssak_cdiv(n8, 256)), 256, 0, st>>>(src, dst, F, RS, H, n8);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
template <typename T_>
int k_col2im_k3s2_t(const T_* dxcol, const T_* pre, T_* out, int B, int F, int Tin, int RS1, int H, hipStream_t st) {
  const long n8 = (long)B * RS1 * H / 8;
  col2im_k3s2_kernel<T_><<<min(4096, ssak_cdiv(n8, 256)), 256, 0, st>>>(dxcol, pre, out, F, Tin, RS1, H, n8);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
#define SSAK_INST_WHISPER(T_)                                                                                      \
  template int k_mel_to_cl_t<T_>(const float*, T_*, int, int, int, int, int, hipStream_t);                        \
  template int k_add_rowvec_t<T_>(const T_*, const T_*, T_*, int, int, int, hipStream_t);                         \
  template int k_copy_rows_padded_t<T_>(const T_*, T_*, int, int, int, int, hipStream_t);                         \
  template int k_col2im_k3s2_t<T_>(const T_*, const T_*, T_*, int, int, int, int, int, hipStream_t);
SSAK_INST_WHISPER(bf16)
SSAK_INST_WHISPER(float)
int k_conv_wgrad_unrearrange(const float* dwr, float* g, int Co, int Ci, int k, hipStream_t st) {
  conv_wgrad_unrearrange_kernel<<<min(2048, ssak_cdiv((long)Co * Ci * k, 256)), 256, 0, st>>>(dwr, g, Co, Ci, k);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
