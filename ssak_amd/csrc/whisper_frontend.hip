// a14: data-movement kernels of the Whisper encoder front end (conv1 k3 p1 + GELU, conv2 k3 s2 p1 + GELU,
// + sinusoidal positions; transformers modeling_whisper.py:618-624) and of its backward.  gfx950.
//
// The two convolutions themselves are GEMMs on overlapping rows of zero-padded channels-last buffers (the same
// Toeplitz operand as the wav2vec2 conv stack): buffers hold one utterance per RS1 = 2*RS2 rows so that the
// stride-2 window of output row kk = b*RS2 + t starts at input row 2*kk for EVERY utterance, which lets the weight
// gradient run as one long-K GEMM over the whole batch.  What is left for this file is HBM-bound reshaping:
// transposing the mel features into channels-last, adding the position table, scattering the input gradient of
// the strided conv back (col2im, fused with GELU') and undoing the [Co][k][Ci] weight layout on the gradient.
#include "kernels.h"

namespace {

// mel [B, C, T] fp32 -> cl [B*RS + ...][C] bf16 at row b*RS + lead + t  (32x32 LDS transpose tiles)
__global__ __launch_bounds__(256) void mel_to_cl_kernel(const float* __restrict__ mel, bf16* __restrict__ cl, int C, int T,
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
    if (t < T && c < C) cl[((long)b * RS + lead + t) * C + c] = (bf16)tile[tx][r];
  }
}

__global__ void add_rowvec_kernel(const bf16* __restrict__ x, const bf16* __restrict__ pos, bf16* __restrict__ out, int F,
                                  int H, long n8) {
  const int hc = H >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long row = i / hc;
    const int c = (int)(i % hc);
    const int t = (int)(row % F);
    const uint4 a = reinterpret_cast<const uint4*>(x)[i];
    const uint4 p = reinterpret_cast<const uint4*>(pos)[(long)t * hc + c];
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, pw[4] = {p.x, p.y, p.z, p.w};
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float lo = __uint_as_float(aw[k] << 16) + __uint_as_float(pw[k] << 16);
      const float hi = __uint_as_float(aw[k] & 0xffff0000u) + __uint_as_float(pw[k] & 0xffff0000u);
      const bf16x2 t2 = {(bf16)lo, (bf16)hi};
      o[k] = __builtin_bit_cast(uint32_t, t2);
    }
    reinterpret_cast<uint4*>(out)[i] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// src [B, F, H] dense -> dst [B, RS, H] (rows >= F zero)
__global__ void copy_rows_padded_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int F, int RS, int H, long n8) {
  const int hc = H >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long row = i / hc;
    const int c = (int)(i % hc);
    const long b = row / RS;
    const int t = (int)(row % RS);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (t < F) v = reinterpret_cast<const uint4*>(src)[((long)b * F + t) * hc + c];
    reinterpret_cast<uint4*>(dst)[i] = v;
  }
}

// input gradient of Conv1d(k=3, stride=2, pad=1) from the column form dxcol [B, F, 3, H] (tap-major), times GELU'(pre):
//   dx[u] = dxcol[u/2][1]                                  (u even)
//         = dxcol[(u+1)/2][0] (if (u+1)/2 < F) + dxcol[(u-1)/2][2]   (u odd)
// written at row b*RS1 + u of out (rows >= Tin zero); pre is stored with a one-row lead (row b*RS1 + 1 + u).
__global__ void col2im_k3s2_kernel(const bf16* __restrict__ dxcol, const bf16* __restrict__ pre, bf16* __restrict__ out, int F,
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
        const uint4 q = reinterpret_cast<const uint4*>(dxcol)[(((long)b * F + t) * 3 + tap) * hc + c];
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[2 * k] += __uint_as_float(w[k] << 16);
          acc[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u);
        }
      };
      if ((u & 1) == 0) {
        add(u >> 1, 1);
      } else {
        if (((u + 1) >> 1) < F) add((u + 1) >> 1, 0);
        add((u - 1) >> 1, 2);
      }
      const uint4 pq = reinterpret_cast<const uint4*>(pre)[((long)b * RS1 + 1 + u) * hc + c];
      const uint32_t pw[4] = {pq.x, pq.y, pq.z, pq.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[2 * k] *= gelu_grad_f(__uint_as_float(pw[k] << 16));
        acc[2 * k + 1] *= gelu_grad_f(__uint_as_float(pw[k] & 0xffff0000u));
      }
    }
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bf16x2 t2 = {(bf16)acc[2 * k], (bf16)acc[2 * k + 1]};
      o[k] = __builtin_bit_cast(uint32_t, t2);
    }
    reinterpret_cast<uint4*>(out)[i] = make_uint4(o[0], o[1], o[2], o[3]);
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

int k_mel_to_cl(const float* mel, bf16* cl, int B, int C, int T, int RS, int lead, hipStream_t st) {
  mel_to_cl_kernel<<<dim3(ssak_cdiv(T, 32), ssak_cdiv(C, 32), B), 256, 0, st>>>(mel, cl, C, T, RS, lead);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
int k_add_rowvec(const bf16* x, const bf16* pos, bf16* out, int B, int F, int H, hipStream_t st) {
  const long n8 = (long)B * F * H / 8;
  add_rowvec_kernel<<<min(4096, ssak_cdiv(n8, 256)), 256, 0, st>>>(x, pos, out, F, H, n8);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
int k_copy_rows_padded(const bf16* src, bf16* dst, int B, int F, int RS, int H, hipStream_t st) {
  const long n8 = (long)B * RS * H / 8;
  copy_rows_padded_kernel<<<min(4096, ssak_cdiv(n8, 256)), 256, 0, st>>>(src, dst, F, RS, H, n8);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
int k_col2im_k3s2(const bf16* dxcol, const bf16* pre, bf16* out, int B, int F, int Tin, int RS1, int H, hipStream_t st) {
  const long n8 = (long)B * RS1 * H / 8;
  col2im_k3s2_kernel<<<min(4096, ssak_cdiv(n8, 256)), 256, 0, st>>>(dxcol, pre, out, F, Tin, RS1, H, n8);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
int k_conv_wgrad_unrearrange(const float* dwr, float* g, int Co, int Ci, int k, hipStream_t st) {
  conv_wgrad_unrearrange_kernel<<<min(2048, ssak_cdiv((long)Co * Ci * k, 256)), 256, 0, st>>>(dwr, g, Co, Ci, k);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
