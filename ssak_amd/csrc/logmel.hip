// a13: Whisper log-mel front end on the device.  gfx950.
//
// Stands behind WhisperFeatureExtractor (transformers feature_extraction_whisper.py:95-168), reached from
// ssak/utils/dataset.py:632-637 with a Whisper processor (set up by ssak/train/transformers/whisper_train.py:356-367):
// pad/trim to 30 s, centred reflect-padded STFT (n_fft 400, hop 160, periodic Hann), |.|^2, drop the last frame,
// 80 Slaney mel filters, log10(max(., 1e-10)), max(x, max(x) - 8), (x + 4) / 4.
//
// Everything is fp32 (the reference is fp32/fp64; bf16 -- also as a 3-term split -- would not hold the 1e-4 tolerance on the
// log scale in weak bins), on the fp32 MATRIX pipe since round 4 (v_mfma_f32_32x32x2_f32: exact fp32 products and sums).
//
// One kernel does reflect padding, the STFT, |.|^2, the mel filters and log10 (stft_mel_kernel below); what reaches HBM in
// between is nothing.  The STFT is a dense DFT with the real-input symmetry folded in: with the periodic Hann window
// w[n] = w[400 - n] and cos / sin even / odd about n = 200,
//     re[k] = sum_{n=0..200} (x[n] + x[400-n]) * cw[n][k],     im[k] = sum_{n=1..199} (x[n] - x[400-n]) * sw[n][k]
// (cw = w cos, halved in row 200 where the pair is one sample; row 0 is zero because w[0] = 0), i.e. half the multiply-adds
// of the plain [400 x 402] DFT matrix of rounds 1-3.  One WAVE owns 32 frames x all 201 bins: its 5 361 samples sit in LDS
// (one pad word per 32, so the 32 frames' equal-n reads hit 32 banks), every K = 2 step is two LDS reads, an add, a subtract
// and 14 MFMAs (7 column tiles of re, 7 of im; the table rows come straight from L2, one load per MFMA gap, two steps ahead); the power
// spectrum goes through the same LDS bytes (transposed) into a second, short MFMA loop against the 80 Slaney filters.  No
// shared operand: the four waves of a workgroup are independent (one per SIMD), 94 waves per 30 s window.  A last pass clamps, scales and writes
// [B, 80, 3000] (and, for the Whisper encoder, a zero-padded channels-last bf16 copy).  Algorithmic bytes 2.88 MB per window;
// 0.53 GFLOP per window on the matrix pipe (155 TFLOP/s fp32) is what bounds it.  Before / after: profiles/r04_logmel_before_after.log.
#include <math.h>

#include <vector>

#include "kernels.h"

namespace {

constexpr int N_FFT = 400, HOP = 160, N_BINS = 201, N_MELS = 80;
constexpr int FOLD_K = 204;             // n = 0..200 and three zero rows (102 K steps = 17 x 6)
constexpr int FOLD_ROWS = FOLD_K + 4;   // + the rows the last trip's reloads touch (zeros): the folded DFT's contraction length (K = 2 per MFMA)
constexpr int BIN_LD = 224;             // 201 bins -> 7 MFMA column tiles of 32
constexpr int MEL_LD = 96;              // 80 filters -> 3 column tiles
constexpr int MEL_K = 216;              // filter rows (bins), zero from 201 on (the last trip's reloads included)
constexpr int XS_SPAN = 31 * HOP + N_FFT + 1;  // samples behind one wave's 32 frames (x[400] of the last frame included)
constexpr int XS_WORDS = XS_SPAN + (XS_SPAN >> 5) + 1;
constexpr int PS_PITCH = 33;            // power spectrum in LDS: [bin][frame], odd pitch
constexpr int LDS_WORDS = BIN_LD * PS_PITCH > XS_WORDS ? BIN_LD * PS_PITCH : XS_WORDS;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ int xs_addr(int i) { return i + (i >> 5); }

// v_mfma_f32_32x32x2_f32: lane l supplies A[row l % 32][k = l / 32] and B[k = l / 32][col l % 32];
// acc[r] = C[row = 8 * (r / 4) + 4 * (l / 32) + r % 4][col = l % 32]
__global__ __launch_bounds__(256) void stft_mel_kernel(const float* __restrict__ wav, const int32_t* __restrict__ lens, int T,
                                                      int n_samples, const float* __restrict__ fold /*[208][2][224]*/,
                                                      const float* __restrict__ melw /*[216][96]*/, float* __restrict__ lm /*[B][F][80]*/,
                                                      int F, unsigned int* __restrict__ gmax) {
  // four independent waves per workgroup (one per SIMD, a workgroup fills a CU's LDS share): single-wave workgroups were placed
  // two to a SIMD here and there, and the launch lasted two wave lifetimes
  __shared__ float lds_all[4][LDS_WORDS];
  float* lds = lds_all[threadIdx.x >> 6];
  const int b = blockIdx.y, f0 = blockIdx.x * 128 + (threadIdx.x >> 6) * 32, lane = threadIdx.x & 63;
  const int r32 = lane & 31, kh = lane >> 5;
  {
    // the zero-padded / trimmed 30 s window, reflect-padded by n_fft / 2 on both sides, read where it lies
    const int len = min(lens ? lens[b] : T, min(T, n_samples));
    const float* wb = wav + (long)b * T;
    const int s0 = f0 * HOP - N_FFT / 2;
    // all 84 loads of a lane in flight at once (the accumulators are not live yet): one round trip to HBM instead of 84
    constexpr int XB = (XS_SPAN + 63) / 64;
    float v[XB];
#pragma unroll
    for (int q = 0; q < XB; ++q) {
      int sidx = s0 + lane + 64 * q;
      if (sidx < 0) sidx = -sidx;
      if (sidx >= n_samples) sidx = 2 * (n_samples - 1) - sidx;
      v[q] = (sidx >= 0 && sidx < len && lane + 64 * q < XS_SPAN) ? wb[sidx] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < XB; ++q)
      if (lane + 64 * q < XS_SPAN) lds[xs_addr(lane + 64 * q)] = v[q];
  }
  __syncthreads();
  f32x16 re[7], im[7];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) re[t][r] = 0.f, im[t][r] = 0.f;
  // Two K steps per trip, each with its own 14 table registers: a register is reloaded (row + 2 steps) right behind the MFMA
  // that read it, so every load has 28 MFMAs (1 800 cycles) to come back and exactly one memory instruction sits in each MFMA
  // gap -- issued in one burst ahead of the MFMAs, 14 loads hold the wave's issue port while the matrix pipe idles (101 us per
  // launch of 8 windows; this form: profiles/r04_logmel_before_after.log).  The next trip's four samples are read from LDS in
  // the middle of this one.  Kept a LOOP on purpose: fully unrolled, the 1 400 MFMAs are 56 KB of straight-line code and the wave
  // waits on the instruction cache instead (162 us).  The table is padded with zero rows for the reloads of the last trip.
  const float* tp = fold + (size_t)kh * (2 * BIN_LD) + r32;
  float cA[7], sA[7], cB[7], sB[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    cA[t] = tp[32 * t], sA[t] = tp[BIN_LD + 32 * t];
    cB[t] = tp[4 * BIN_LD + 32 * t], sB[t] = tp[5 * BIN_LD + 32 * t];
  }
  const int xrow = r32 * HOP;
  float xa0 = lds[xs_addr(xrow + kh)], xa1 = lds[xs_addr(xrow + N_FFT - kh)];
  float xb0 = lds[xs_addr(xrow + 2 + kh)], xb1 = lds[xs_addr(xrow + N_FFT - 2 - kh)];
#pragma clang loop unroll(disable)
  for (int s = 0; s < FOLD_K / 2; s += 2) {
    const float* tn = tp + (size_t)(s + 2) * (4 * BIN_LD);
    const float apA = xa0 + xa1, amA = xa0 - xa1, apB = xb0 + xb1, amB = xb0 - xb1;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      re[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(apA, cA[t], re[t], 0, 0, 0);
      cA[t] = tn[32 * t];
      __builtin_amdgcn_sched_barrier(0);
      im[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(amA, sA[t], im[t], 0, 0, 0);
      sA[t] = tn[BIN_LD + 32 * t];
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      const int n = 2 * (s + 2) + kh;  // (reads past n = 203 on the last trip stay inside the wave's samples and are not used)
      xa0 = lds[xs_addr(xrow + n)], xa1 = lds[xs_addr(xrow + N_FFT - n)];
      xb0 = lds[xs_addr(xrow + n + 2)], xb1 = lds[xs_addr(xrow + N_FFT - n - 2)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      re[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(apB, cB[t], re[t], 0, 0, 0);
      cB[t] = tn[4 * BIN_LD + 32 * t];
      __builtin_amdgcn_sched_barrier(0);
      im[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(amB, sB[t], im[t], 0, 0, 0);
      sB[t] = tn[5 * BIN_LD + 32 * t];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();  // the samples are dead: their LDS bytes take the power spectrum, [bin][frame]
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      lds[(32 * t + r32) * PS_PITCH + 8 * (r >> 2) + 4 * kh + (r & 3)] = re[t][r] * re[t][r] + im[t][r] * im[t][r];
  __syncthreads();
  f32x16 ml[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) ml[j][r] = 0.f;
  // the same pattern on the filters: six K steps (18 MFMAs) per trip, a filter register reloaded six steps ahead right behind its MFMA
  const float* mp = melw + (size_t)kh * MEL_LD + r32;
  constexpr int MCH = 6;
  static_assert((FOLD_K / 2) % MCH == 0 && FOLD_K + 2 * MCH <= MEL_K && FOLD_K % 4 == 0, "whole trips inside the padded tables");
  float mc[MCH][3], pa[MCH];
#pragma unroll
  for (int q = 0; q < MCH; ++q) {
    pa[q] = lds[(2 * q + kh) * PS_PITCH + r32];
#pragma unroll
    for (int j = 0; j < 3; ++j) mc[q][j] = mp[(size_t)q * (2 * MEL_LD) + 32 * j];
  }
#pragma clang loop unroll(disable)
  for (int s = 0; s < FOLD_K / 2; s += MCH) {
    float a[MCH];
#pragma unroll
    for (int q = 0; q < MCH; ++q) a[q] = pa[q];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < MCH; ++q) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        ml[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], mc[q][j], ml[j], 0, 0, 0);
        mc[q][j] = mp[(size_t)(s + MCH + q) * (2 * MEL_LD) + 32 * j];
        __builtin_amdgcn_sched_barrier(0);
      }
      // the next trip's power-spectrum column (bins past 223 on the last trip: clamped, multiplied by zero filter rows)
      const int kb = 2 * (s + MCH + q) + kh;
      pa[q] = lds[(kb < BIN_LD ? kb : BIN_LD - 1) * PS_PITCH + r32];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float lmax = -INFINITY;
  float* lb = lm + (long)b * F * N_MELS;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int m = 32 * j + r32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = f0 + 8 * (r >> 2) + 4 * kh + (r & 3);
      if (m < N_MELS && f < F) {
        const float lv = log10f(fmaxf(ml[j][r], 1e-10f));
        lb[(long)f * N_MELS + m] = lv;
        lmax = fmaxf(lmax, lv);
      }
    }
  }
  lmax = wave_max(lmax);
  if (lane == 0 && lmax > -INFINITY) {
    // order-preserving float -> uint map so that atomicMax works for negative values too
    unsigned int u = __float_as_uint(lmax);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    atomicMax(gmax + b, u);
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

extern "C" size_t ssak_logmel_table_floats(void) { return (size_t)FOLD_ROWS * 2 * BIN_LD + (size_t)MEL_K * MEL_LD; }

// tables (host computes in double, like the reference's numpy path): folded DFT [208][2][224] (Hann-weighted cos rows, then sin
// rows, of n = 0..200; row 200 of the cos half is halved -- see the header), then the mel filters [216][96]
extern "C" int ssak_logmel_init_tables(float* tables_dev) {
  SSAK_REQUIRE(tables_dev, "logmel_init_tables: null pointer");
  std::vector<float> h(ssak_logmel_table_floats(), 0.f);
  const double PI = 3.14159265358979323846;
  for (int n = 0; n <= N_FFT / 2; ++n) {
    const double w = (0.5 - 0.5 * cos(2.0 * PI * n / N_FFT)) * (n == N_FFT / 2 ? 0.5 : 1.0);  // periodic Hann
    for (int k = 0; k < N_BINS; ++k) {
      const double ang = 2.0 * PI * (double)((long)k * n % N_FFT) / N_FFT;
      h[((size_t)n * 2 + 0) * BIN_LD + k] = (float)(w * cos(ang));
      h[((size_t)n * 2 + 1) * BIN_LD + k] = (float)(w * sin(ang));
    }
  }
  float* mf = h.data() + (size_t)FOLD_ROWS * 2 * BIN_LD;
  std::vector<double> fpts(N_MELS + 2);
  const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(8000.0);
  for (int i = 0; i < N_MELS + 2; ++i) fpts[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (N_MELS + 1));
  for (int k = 0; k < N_BINS; ++k) {
    const double fk = 8000.0 * k / (N_BINS - 1);
    for (int m = 0; m < N_MELS; ++m) {
      const double down = (fk - fpts[m]) / (fpts[m + 1] - fpts[m]);
      const double up = (fpts[m + 2] - fk) / (fpts[m + 2] - fpts[m + 1]);
      const double v = fmax(0.0, fmin(down, up)) * (2.0 / (fpts[m + 2] - fpts[m]));
      mf[(size_t)k * MEL_LD + m] = (float)v;
    }
  }
  SSAK_HIP(hipMemcpy(tables_dev, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return SSAK_OK;
}

extern "C" size_t ssak_logmel_workspace_bytes(int B, int n_samples) {
  const size_t F = (size_t)n_samples / HOP;
  return ((size_t)B * F * N_MELS + 64 + (size_t)B) * sizeof(float);  // log10 mel energies before the clamp + one running maximum per window
}

extern "C" int ssak_logmel_whisper(const float* wav, const int32_t* lens, int B, int T, int n_samples, const float* tables,
                                   float* mel, void* mel_cl_bf16, int cl_rows, int cl_lead, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(wav && tables && workspace && (mel || mel_cl_bf16), "logmel: null pointer");
  SSAK_REQUIRE(B > 0 && T > 0 && n_samples >= N_FFT && n_samples % HOP == 0, "logmel: n_samples must be a multiple of 160 (>= 400)");
  SSAK_REQUIRE(workspace_bytes >= ssak_logmel_workspace_bytes(B, n_samples), "logmel: workspace too small");
  const int F = n_samples / HOP;          // 3000 frames kept (the STFT's last frame is dropped)
  SSAK_REQUIRE(!mel_cl_bf16 || cl_rows >= cl_lead + F, "logmel: channels-last copy too small");
  hipStream_t st = (hipStream_t)stream;
  // algorithmic bytes: the waveform read once, the features written once in each form asked for (2.88 MB / window)
  ProfScope prof_scope(PROF_LOGMEL, (double)B * ((double)n_samples * 4.0 + (double)F * N_MELS * ((mel ? 4.0 : 0.0) + (mel_cl_bf16 ? 2.0 : 0.0))), st);
  float* lm = (float*)workspace;
  unsigned int* gmax = (unsigned int*)(lm + (size_t)B * F * N_MELS + 32);
  SSAK_HIP(hipMemsetAsync(gmax, 0, (size_t)B * sizeof(unsigned int), st));
  const float* melf = tables + (size_t)FOLD_ROWS * 2 * BIN_LD;
  stft_mel_kernel<<<dim3(ssak_cdiv(F, 128), B), 256, 0, st>>>(wav, lens, T, n_samples, tables, melf, lm, F, gmax);
  SSAK_LAUNCH_CHECK();
  logmel_finalize_kernel<<<dim3(ssak_cdiv(F, 32), B), 256, 0, st>>>(lm, gmax, F, mel, (bf16*)mel_cl_bf16, cl_rows, cl_lead);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
