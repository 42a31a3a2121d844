// a13: Whisper log-mel front end on the device.  gfx950.
//
// Stands behind WhisperFeatureExtractor (transformers feature_extraction_whisper.py:95-168), reached from
// ssak/utils/dataset.py:632-637 with a Whisper processor (set up by ssak/train/transformers/whisper_train.py:356-367):
// pad/trim to 30 s, centred reflect-padded STFT (n_fft 400, hop 160, periodic Hann), |.|^2, drop the last frame,
// 80 Slaney mel filters, log10(max(., 1e-10)), max(x, max(x) - 8), (x + 4) / 4.
//
// Everything is fp32 (the reference is fp32/fp64; bf16 would not hold the 1e-4 tolerance on the log scale).
// The STFT is a dense DFT written as a GEMM on OVERLAPPING rows of the reflect-padded waveform (frame f = the 400
// samples at offset 160 f: lda = 160 < K = 400, the same Toeplitz trick as the conv layers) against a
// [400 x 402] matrix of Hann-weighted cos / -sin columns; its epilogue squares and adds the (re, im) pairs, so the
// complex spectrum never reaches HBM.  A second small GEMM applies the mel filters with log10 in the epilogue and a
// per-utterance running maximum; a last pass clamps, scales and writes [B, 80, 3000] (and, for the Whisper encoder,
// a zero-padded channels-last bf16 copy).  Algorithmic bytes 2.88 MB per 30 s window; the DFT makes it
// FP32-VALU-bound rather than HBM-bound (0.96 GFLOP per window).
#include <math.h>

#include <vector>

#include "kernels.h"

namespace {

constexpr int N_FFT = 400, HOP = 160, N_BINS = 201, N_MELS = 80;
constexpr int DFT_COLS = 2 * N_BINS;   // 402: (re, im) interleaved
constexpr int DFT_LD = 408;            // padded leading dimension of the tables
constexpr int PW_LD = 208;             // power spectrum leading dimension (201 -> 208)

// C[m][n] = sum_k A[m*lda + k] * Bm[k*ldb + n]; 64x64 tile, 256 threads, 4x4 per thread, K chunks of 16 through LDS
constexpr int TS = 64, TK = 16;
enum { EPI_POWER = 0, EPI_LOGMEL = 1 };

template <int EPI>
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, long lda, long sa, const float* __restrict__ Bm,
                                                    int ldb, float* __restrict__ C, int ldc, long sc, int M, int N, int K,
                                                    unsigned int* __restrict__ gmax) {
  __shared__ float As[TK][TS + 4];
  __shared__ float Bs[TK][TS + 4];
  const int b = blockIdx.z;
  const float* Ab = A + (long)b * sa;
  float* Cb = C + (long)b * sc;
  const int m0 = blockIdx.y * TS, n0 = blockIdx.x * TS;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += TK) {
    for (int e = threadIdx.x; e < TS * TK; e += 256) {
      const int kk = e & (TK - 1), r = e >> 4;  // A: consecutive threads walk k (contiguous in memory)
      const int m = m0 + r, k = k0 + kk;
      As[kk][r] = (m < M && k < K) ? Ab[(long)m * lda + k] : 0.f;
    }
    for (int e = threadIdx.x; e < TS * TK; e += 256) {
      const int c = e & (TS - 1), kk = e >> 6;  // B: consecutive threads walk n
      const int n = n0 + c, k = k0 + kk;
      Bs[kk][c] = (n < N && k < K) ? Bm[(long)k * ldb + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      float a[4], bb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) bb[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], bb[j], acc[i][j]);
    }
    __syncthreads();
  }
  float lmax = -INFINITY;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
    if (EPI == EPI_POWER) {
      // columns come in (re, im) pairs: this thread's 4 columns are bins n/2 and n/2+1
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const int n = n0 + tx * 4 + j;
        if (n < N) Cb[(long)m * ldc + (n >> 1)] = acc[i][j] * acc[i][j] + acc[i][j + 1] * acc[i][j + 1];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + tx * 4 + j;
        if (n < N) {
          const float v = log10f(fmaxf(acc[i][j], 1e-10f));
          Cb[(long)m * ldc + n] = v;
          lmax = fmaxf(lmax, v);
        }
      }
    }
  }
  if (EPI == EPI_LOGMEL) {
    lmax = wave_max(lmax);
    if ((threadIdx.x & 63) == 0 && lmax > -INFINITY) {
      // order-preserving float -> uint map so that atomicMax works for negative values too
      unsigned int u = __float_as_uint(lmax);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      atomicMax(gmax + b, u);
    }
  }
}

// reflect padding (n_fft/2 on both sides) of the zero-padded / trimmed 30 s window
__global__ void reflect_pad_kernel(const float* __restrict__ wav, const int32_t* __restrict__ lens, int T, int n_samples,
                                   float* __restrict__ out, int out_len) {
  const int b = blockIdx.y;
  const int len = min(lens ? lens[b] : T, min(T, n_samples));
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < out_len; i += gridDim.x * blockDim.x) {
    int s = i - N_FFT / 2;
    if (s < 0) s = -s;
    if (s >= n_samples) s = 2 * (n_samples - 1) - s;
    out[(long)b * out_len + i] = (s < len) ? wav[(long)b * T + s] : 0.f;
  }
}

__global__ void logmel_finalize_kernel(const float* __restrict__ lm /*[B][F][80]*/, const unsigned int* __restrict__ gmax,
                                       int F, float* __restrict__ mel /*[B][80][F]*/, bf16* __restrict__ cl /*[B][rs][80]*/,
                                       int cl_rows, int cl_lead) {
  __shared__ float tile[32][N_MELS + 1];
  const int b = blockIdx.y, f0 = blockIdx.x * 32;
  unsigned int u = gmax[b];
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  const float floor_v = __uint_as_float(u) - 8.f;
  for (int e = threadIdx.x; e < 32 * N_MELS; e += blockDim.x) {
    const int fr = e / N_MELS, m = e % N_MELS;
    const int f = f0 + fr;
    float v = 0.f;
    if (f < F) v = (fmaxf(lm[((long)b * F + f) * N_MELS + m], floor_v) + 4.f) * 0.25f;
    tile[fr][m] = v;
    if (cl && f < F) cl[((long)b * cl_rows + cl_lead + f) * N_MELS + m] = (bf16)v;
  }
  __syncthreads();
  if (mel) {
    for (int e = threadIdx.x; e < 32 * N_MELS; e += blockDim.x) {
      const int m = e >> 5, fr = e & 31;
      if (f0 + fr < F) mel[((long)b * N_MELS + m) * F + f0 + fr] = tile[fr][m];
    }
  }
}

double hz_to_mel(double f) { return f >= 1000.0 ? 15.0 + log(f / 1000.0) * (27.0 / log(6.4)) : 3.0 * f / 200.0; }
double mel_to_hz(double m) { return m >= 15.0 ? 1000.0 * exp(log(6.4) / 27.0 * (m - 15.0)) : 200.0 * m / 3.0; }

}  // namespace

extern "C" size_t ssak_logmel_table_floats(void) { return (size_t)N_FFT * DFT_LD + (size_t)PW_LD * N_MELS; }

// tables (host computes in double, like the reference's numpy path): DFT [400][408] then mel filters [208][80]
extern "C" int ssak_logmel_init_tables(float* tables_dev) {
  SSAK_REQUIRE(tables_dev, "logmel_init_tables: null pointer");
  std::vector<float> h(ssak_logmel_table_floats(), 0.f);
  const double PI = 3.14159265358979323846;
  for (int n = 0; n < N_FFT; ++n) {
    const double w = 0.5 - 0.5 * cos(2.0 * PI * n / N_FFT);  // periodic Hann
    for (int k = 0; k < N_BINS; ++k) {
      const double ang = 2.0 * PI * (double)((long)k * n % N_FFT) / N_FFT;
      h[(size_t)n * DFT_LD + 2 * k] = (float)(w * cos(ang));
      h[(size_t)n * DFT_LD + 2 * k + 1] = (float)(-w * sin(ang));
    }
  }
  float* mf = h.data() + (size_t)N_FFT * DFT_LD;
  std::vector<double> fpts(N_MELS + 2);
  const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(8000.0);
  for (int i = 0; i < N_MELS + 2; ++i) fpts[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (N_MELS + 1));
  for (int k = 0; k < N_BINS; ++k) {
    const double fk = 8000.0 * k / (N_BINS - 1);
    for (int m = 0; m < N_MELS; ++m) {
      const double down = (fk - fpts[m]) / (fpts[m + 1] - fpts[m]);
      const double up = (fpts[m + 2] - fk) / (fpts[m + 2] - fpts[m + 1]);
      const double v = fmax(0.0, fmin(down, up)) * (2.0 / (fpts[m + 2] - fpts[m]));
      mf[(size_t)k * N_MELS + m] = (float)v;
    }
  }
  SSAK_HIP(hipMemcpy(tables_dev, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return SSAK_OK;
}

extern "C" size_t ssak_logmel_workspace_bytes(int B, int n_samples) {
  const size_t F = (size_t)n_samples / HOP;
  const size_t padded = (size_t)n_samples + N_FFT;
  return ((size_t)B * padded + (size_t)B * (F + 1) * PW_LD + (size_t)B * F * N_MELS + 64 + (size_t)B) * sizeof(float);
}

extern "C" int ssak_logmel_whisper(const float* wav, const int32_t* lens, int B, int T, int n_samples, const float* tables,
                                   float* mel, void* mel_cl_bf16, int cl_rows, int cl_lead, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(wav && tables && workspace && (mel || mel_cl_bf16), "logmel: null pointer");
  SSAK_REQUIRE(B > 0 && T > 0 && n_samples >= N_FFT && n_samples % HOP == 0, "logmel: n_samples must be a multiple of 160 (>= 400)");
  SSAK_REQUIRE(workspace_bytes >= ssak_logmel_workspace_bytes(B, n_samples), "logmel: workspace too small");
  const int F = n_samples / HOP;          // 3000 frames kept (the STFT's last frame is dropped)
  const int padded = n_samples + N_FFT;
  SSAK_REQUIRE(!mel_cl_bf16 || cl_rows >= cl_lead + F, "logmel: channels-last copy too small");
  hipStream_t st = (hipStream_t)stream;
  float* xp = (float*)workspace;
  float* pw = xp + (size_t)B * padded;
  float* lm = pw + (size_t)B * (F + 1) * PW_LD;
  unsigned int* gmax = (unsigned int*)(lm + (size_t)B * F * N_MELS + 32);
  SSAK_HIP(hipMemsetAsync(gmax, 0, (size_t)B * sizeof(unsigned int), st));
  reflect_pad_kernel<<<dim3(ssak_cdiv(padded, 256 * 8), B), 256, 0, st>>>(wav, lens, T, n_samples, xp, padded);
  SSAK_LAUNCH_CHECK();
  // frames x DFT: M = F (last frame dropped), N = 402, K = 400, A rows overlap (lda = hop)
  sgemm_kernel<EPI_POWER><<<dim3(ssak_cdiv(DFT_COLS, TS), ssak_cdiv(F, TS), B), 256, 0, st>>>(
      xp, HOP, padded, tables, DFT_LD, pw, PW_LD, (long)(F + 1) * PW_LD, F, DFT_COLS, N_FFT, nullptr);
  SSAK_LAUNCH_CHECK();
  const float* melf = tables + (size_t)N_FFT * DFT_LD;
  sgemm_kernel<EPI_LOGMEL><<<dim3(ssak_cdiv(N_MELS, TS), ssak_cdiv(F, TS), B), 256, 0, st>>>(
      pw, PW_LD, (long)(F + 1) * PW_LD, melf, N_MELS, lm, N_MELS, (long)F * N_MELS, F, N_MELS, N_BINS, gmax);
  SSAK_LAUNCH_CHECK();
  logmel_finalize_kernel<<<dim3(ssak_cdiv(F, 32), B), 256, 0, st>>>(lm, gmax, F, mel, (bf16*)mel_cl_bf16, cl_rows, cl_lead);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
