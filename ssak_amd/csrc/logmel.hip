// a13: Whisper log-mel front end on the device.  gfx950.
//
// Stands behind WhisperFeatureExtractor (transformers feature_extraction_whisper.py:95-168), reached from
// ssak/utils/dataset.py:632-637 with a Whisper processor (set up by ssak/train/transformers/whisper_train.py:356-367):
// pad/trim to 30 s, centred reflect-padded STFT (n_fft 400, hop 160, periodic Hann), |.|^2, drop the last frame,
// 80 Slaney mel filters, log10(max(., 1e-10)), max(x, max(x) - 8), (x + 4) / 4.
//
// Everything is fp32 (the reference is fp32 / fp64; bf16 would not hold the 2e-4 tolerance on the log scale in weak bins).
//
// Round 5: the STFT is an FFT, not a DFT matrix (rounds 1-4: [400 x 402] fp32 GEMM, then the folded [201 x 402] product on the
// fp32 matrix pipe, 0.53 GFLOP per window and 84 us per launch of 8 windows).  A frame's 400 windowed samples are packed as
// 200 complex points z[m] = y[2m] + i y[2m+1]; Z = FFT_200(z) by a mixed-radix Stockham transform (5 x 5 x 8: in this order
// every pass reads and writes LDS at strides of 1 or 5 slots over the lanes -- no padding, no swizzle, and a butterfly's
// addresses are one base plus immediates) and
//     E = (Z[k] + conj Z[200-k]) / 2,  T = e^{-2 pi i k / 400} * (Z[k] - conj Z[200-k]) / (2 i):   |X[k]|^2 = |E + T|^2,  |X[200-k]|^2 = |E - T|^2
// gives the 201 bins: ~10 k flops per frame instead of 322 k.  One WAVE owns 8 frames.  Its LDS slice (12.8 KB) holds, in
// turn, the wave's 1 520 samples, the transform in place (a pass reads ALL its butterflies' inputs into registers before it
// writes any output: a wave executes in program order, so no second buffer and no barrier), and the power spectrum
// [bin][frame] that feeds the mel product -- 55 v_mfma_f32_16x16x4_f32 per wave (exact fp32 products), the filter bank's
// band structure known at compile time (a 16-filter row tile touches 4-24 K steps of 4 bins, not 51; eight of the tile's 16
// frame columns are idle).
// Twelve waves per CU (two 6-wave workgroups, 161.5 KB of LDS): 504 workgroups for 8 windows = one round.  Per-wave maxima go to slots of their own
// (no atomics, no memset launch); the last pass clamps, scales and writes [B, 80, 3000] (and, for the Whisper encoder, a
// zero-padded channels-last bf16 copy).  Algorithmic bytes 2.88 MB per window.  Before / after: profiles/r05_logmel_fft.log.
#include <math.h>

#include <vector>

#include "kernels.h"

namespace {

constexpr int N_FFT = 400, HOP = 160, N_BINS = 201, N_MELS = 80;
constexpr int NZ = N_FFT / 2;             // complex points per frame
constexpr int FPW = 8;                    // frames per wave
constexpr int WPB = 6;                    // waves per workgroup: 2 workgroups = 12 waves per CU, three to a SIMD
constexpr int ZP = NZ;                    // complex slots per frame in LDS
constexpr int WAVE_LDS = FPW * ZP * 8;    // 12 800 bytes
constexpr int XS_SPAN = (FPW - 1) * HOP + N_FFT;  // samples behind one wave's frames
constexpr int PS_PITCH = FPW + 1;         // power spectrum in LDS: [bin][frame], odd pitch
static_assert(XS_SPAN * 4 <= WAVE_LDS && N_BINS * PS_PITCH * 4 <= WAVE_LDS, "samples and power spectrum share the transform's bytes");
// tables (floats): W_25^{qk} [5][4] complex, W_200^{qk} [25][7] complex, e^{-2 pi i k / 400} [104] complex, Hann [400] (+ 2 to a
// multiple of 4); then the filters
constexpr int TAB_TW2 = 0, TAB_TW3 = TAB_TW2 + 5 * 4 * 2, TAB_TWP = TAB_TW3 + 25 * 7 * 2, TAB_HANN = TAB_TWP + 104 * 2;
constexpr int FFT_TAB = TAB_HANN + N_FFT + 2;  // 1 000
static_assert(FFT_TAB % 4 == 0 && FFT_TAB <= 1024 && FFT_TAB / 4 <= 64 * WPB, "one float4 per thread copies the tables");
constexpr int MEL_OFF = 1024;             // filters [bin][MEL_LD] behind the transform's tables
constexpr int MEL_LD = 96, MEL_K = 204;
constexpr int LDS_BYTES = WPB * WAVE_LDS + FFT_TAB * 4;
// bins [MEL_K0[t], MEL_K1[t]) hold every nonzero weight of filters 16 t .. 16 t + 15 (checked against the table by
// ssak_logmel_init_tables)
constexpr int MEL_K0[5] = {0, 12, 28, 56, 104};
constexpr int MEL_K1[5] = {16, 32, 60, 112, 200};

struct cf {
  float x, y;
};
__device__ __forceinline__ cf operator+(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf operator-(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf operator*(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cf mul_mi(cf a) { return {a.y, -a.x}; }  // a * (-i)
// u[p] = sum_q v[q] e^{-2 pi i p q / 4}
__device__ __forceinline__ void dft4(cf a0, cf a1, cf a2, cf a3, cf& u0, cf& u1, cf& u2, cf& u3) {
  const cf t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mul_mi(a1 - a3);
  u0 = t0 + t2, u1 = t1 + t3, u2 = t0 - t2, u3 = t1 - t3;
}
__device__ __forceinline__ void dft8(cf (&v)[8]) {
  cf e[4], o[4];
  dft4(v[0], v[2], v[4], v[6], e[0], e[1], e[2], e[3]);
  dft4(v[1], v[3], v[5], v[7], o[0], o[1], o[2], o[3]);
  const float h = 0.70710678118654752f;
  o[1] = (cf){(o[1].x + o[1].y) * h, (o[1].y - o[1].x) * h};   // * (1 - i) / sqrt 2
  o[2] = mul_mi(o[2]);
  o[3] = (cf){(o[3].y - o[3].x) * h, -(o[3].x + o[3].y) * h};  // * (-1 - i) / sqrt 2
#pragma unroll
  for (int p = 0; p < 4; ++p) v[p] = e[p] + o[p], v[p + 4] = e[p] - o[p];
}
__device__ __forceinline__ void dft5(cf (&v)[5]) {
  const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f, s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
  const cf p1 = v[1] + v[4], p2 = v[2] + v[3], d1 = v[1] - v[4], d2 = v[2] - v[3];
  const cf a1 = {v[0].x + c1 * p1.x + c2 * p2.x, v[0].y + c1 * p1.y + c2 * p2.y};
  const cf a2 = {v[0].x + c2 * p1.x + c1 * p2.x, v[0].y + c2 * p1.y + c1 * p2.y};
  const cf b1 = mul_mi((cf){s1 * d1.x + s2 * d2.x, s1 * d1.y + s2 * d2.y});
  const cf b2 = mul_mi((cf){s2 * d1.x - s1 * d2.x, s2 * d1.y - s1 * d2.y});
  v[0] = v[0] + p1 + p2;
  v[1] = a1 + b1, v[4] = a1 - b1, v[2] = a2 + b2, v[3] = a2 - b2;
}

typedef __attribute__((ext_vector_type(4))) float f32x4_;

__global__ __launch_bounds__(64 * WPB) void stft_fft_mel_kernel(const float* __restrict__ wav, const int32_t* __restrict__ lens, int T,
                                                               int n_samples, const float* __restrict__ tables,
                                                               float* __restrict__ lm /*[B][F][80]*/, int F,
                                                               float* __restrict__ wmax /*[B][gridDim.x * WPB]*/) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  char* const wl = smem + wave * WAVE_LDS;
  float* const wf = reinterpret_cast<float*>(wl);
  cf* const wz = reinterpret_cast<cf*>(wl);
  const float* const tab = reinterpret_cast<const float*>(smem + WPB * WAVE_LDS);
  const int b = blockIdx.y, f0 = blockIdx.x * (FPW * WPB) + wave * FPW;
  if (threadIdx.x < FFT_TAB / 4)
    reinterpret_cast<f32x4_*>(smem + WPB * WAVE_LDS)[threadIdx.x] = reinterpret_cast<const f32x4_*>(tables)[threadIdx.x];
  {
    // the zero-padded / trimmed 30 s window, reflect-padded by n_fft / 2 on both sides, read where it lies; all loads of a
    // lane in flight at once
    const int len = min(lens ? lens[b] : T, min(T, n_samples));
    const float* wb = wav + (long)b * T;
    const int s0 = f0 * HOP - N_FFT / 2;
    constexpr int XB = (XS_SPAN + 63) / 64;
    float v[XB];
    if (f0 < F && s0 >= 0 && s0 + XB * 64 <= len) {
      // interior wave (all but the first and last of a window, and those past a short utterance's end): plain loads
#pragma unroll
      for (int q = 0; q < XB; ++q) v[q] = wb[s0 + lane + 64 * q];
    } else {
#pragma unroll
      for (int q = 0; q < XB; ++q) {
        int sidx = s0 + lane + 64 * q;
        if (sidx < 0) sidx = -sidx;
        if (sidx >= n_samples) sidx = 2 * (n_samples - 1) - sidx;
        v[q] = (f0 < F && sidx >= 0 && sidx < len && lane + 64 * q < XS_SPAN) ? wb[sidx] : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < XB; ++q)
      if (lane + 64 * q < XS_SPAN) wf[lane + 64 * q] = v[q];
  }
  __syncthreads();  // the tables (shared) and this wave's samples; from here on the waves are independent
  if (f0 >= F) return;
  // ---- pass 1: radix 5, Ns = 1 (no twiddles).  Butterfly (frame, j < 40) reads y[2 (j + 40 q)], y[.. + 1], writes points 5 j + p
  // (all addresses of a butterfly = one base + immediates; a stride of 5 slots over the lanes touches every bank once)
  constexpr int NIT5 = FPW * 40 / 64;
  static_assert(FPW * 40 % 64 == 0, "whole iterations");
  {
    cf v[NIT5][5];
#pragma unroll
    for (int it = 0; it < NIT5; ++it) {
      const int bf = it * 64 + lane, fr = bf / 40, j = bf - 40 * fr;
      const float* ys = wf + fr * HOP + 2 * j;
      const float* ws = tab + TAB_HANN + 2 * j;
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const float2 y = *reinterpret_cast<const float2*>(ys + 80 * q);
        const float2 w = *reinterpret_cast<const float2*>(ws + 80 * q);
        v[it][q] = (cf){y.x * w.x, y.y * w.y};
      }
    }
#pragma unroll
    for (int it = 0; it < NIT5; ++it) {
      const int bf = it * 64 + lane, fr = bf / 40, j = bf - 40 * fr;
      dft5(v[it]);
      cf* o = wz + fr * ZP + 5 * j;
#pragma unroll
      for (int p = 0; p < 5; ++p) o[p] = v[it][p];
    }
  }
  // ---- pass 2: radix 5, Ns = 5.  Reads z[j + 40 q] * W_25^{q k}, k = j mod 5, writes points 25 (j / 5) + k + 5 p
  {
    cf v[NIT5][5];
#pragma unroll
    for (int it = 0; it < NIT5; ++it) {
      const int bf = it * 64 + lane, fr = bf / 40, j = bf - 40 * fr;
      const cf* in = wz + fr * ZP + j;
#pragma unroll
      for (int q = 0; q < 5; ++q) v[it][q] = in[40 * q];
    }
#pragma unroll
    for (int it = 0; it < NIT5; ++it) {
      const int bf = it * 64 + lane, fr = bf / 40, j = bf - 40 * fr, j5 = j / 5, k = j - 5 * j5;
      const cf* tw = reinterpret_cast<const cf*>(tab + TAB_TW2) + 4 * k;
#pragma unroll
      for (int q = 1; q < 5; ++q) v[it][q] = v[it][q] * tw[q - 1];
      dft5(v[it]);
      cf* o = wz + fr * ZP + 25 * j5 + k;
#pragma unroll
      for (int p = 0; p < 5; ++p) o[5 * p] = v[it][p];
    }
  }
  // ---- pass 3: radix 8, Ns = 25.  Butterfly (frame, j < 25) reads z[j + 25 q] * W_200^{q j} and writes the same slots
  {
    constexpr int NIT = (FPW * 25 + 63) / 64;
    cf v[NIT][8];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int bf = min(it * 64 + lane, FPW * 25 - 1), fr = bf / 25, j = bf - 25 * fr;
      const cf* in = wz + fr * ZP + j;
#pragma unroll
      for (int q = 0; q < 8; ++q) v[it][q] = in[25 * q];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int bf0 = it * 64 + lane, bf = min(bf0, FPW * 25 - 1), fr = bf / 25, j = bf - 25 * fr;
      const cf* tw = reinterpret_cast<const cf*>(tab + TAB_TW3) + 7 * j;
#pragma unroll
      for (int q = 1; q < 8; ++q) v[it][q] = v[it][q] * tw[q - 1];
      dft8(v[it]);
      if (bf0 < FPW * 25) {
        cf* o = wz + fr * ZP + j;
#pragma unroll
        for (int p = 0; p < 8; ++p) o[25 * p] = v[it][p];
      }
    }
  }
  // ---- bins k and 200 - k from Z[k] and Z[200 - k]; then the power spectrum takes the transform's place, [bin][frame]
  {
    constexpr int NP = N_FFT / 4 + 1;  // 101 pairs
    constexpr int NIT = (FPW * NP + 63) / 64;
    float pl[NIT], ph[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int item = min(it * 64 + lane, FPW * NP - 1), fr = item / NP, k = item % NP;
      const cf zk = wz[fr * ZP + k], zm = wz[fr * ZP + (k == 0 ? 0 : NZ - k)];
      const cf w = reinterpret_cast<const cf*>(tab + TAB_TWP)[k];
      const cf e = {0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)};   // (Z[k] + conj Z[200-k]) / 2
      const cf o = {0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x)};  // (Z[k] - conj Z[200-k]) / (2 i)
      const cf t = w * o;
      const cf xp = e + t, xm = e - t;
      pl[it] = xp.x * xp.x + xp.y * xp.y;
      ph[it] = xm.x * xm.x + xm.y * xm.y;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int item = it * 64 + lane, fr = min(item, FPW * NP - 1) / NP, k = min(item, FPW * NP - 1) % NP;
      if (item < FPW * NP) {
        wf[k * PS_PITCH + fr] = pl[it];
        wf[(NZ - k) * PS_PITCH + fr] = ph[it];  // bin 200 - k (k = 100: the same bin, the same value)
      }
    }
  }
  // ---- mel filters on the fp32 matrix pipe: D[filter][frame] = sum_k W[k][filter] P[k][frame], 16 x 16 x 4 per instruction
  // (lane l supplies A[row l % 16][k = l / 16] and B[k = l / 16][col l % 16]; acc[r] = D[row 4 (l / 16) + r][col l % 16])
  const float* melw = tables + MEL_OFF;
  const int l16 = lane & 15, lq = lane >> 4;
  float lmax = -INFINITY;
  float* lb = lm + (long)b * F * N_MELS;
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int k0 = MEL_K0[t], nst = (MEL_K1[t] - MEL_K0[t]) / 4;
    float a[24];
#pragma unroll
    for (int s2 = 0; s2 < 24; ++s2)
      if (s2 < nst) a[s2] = melw[(k0 + 4 * s2 + lq) * MEL_LD + 16 * t + l16];
    f32x4_ acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 24; ++s2)
      if (s2 < nst) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s2], wf[(k0 + 4 * s2 + lq) * PS_PITCH + l16], acc, 0, 0, 0);
    const int f = f0 + l16;
    if (l16 < FPW && f < F) {
      f32x4_ lv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        lv[r] = __builtin_amdgcn_logf(fmaxf(acc[r], 1e-10f)) * 0.30102999566398120f;  // v_log_f32 (log2, 1 ulp; the argument is a normal number)
        lmax = fmaxf(lmax, lv[r]);
      }
      *reinterpret_cast<f32x4_*>(lb + (long)f * N_MELS + 16 * t + 4 * lq) = lv;
    }
  }
  lmax = wave_max(lmax);
  if (lane == 0) wmax[(long)b * gridDim.x * WPB + blockIdx.x * WPB + wave] = lmax;
}

__global__ void logmel_finalize_kernel(const float* __restrict__ lm /*[B][F][80]*/, const float* __restrict__ wmax, int nslots,
                                       int F, float* __restrict__ mel /*[B][80][F]*/, bf16* __restrict__ cl /*[B][rs][80]*/,
                                       int cl_rows, int cl_lead) {
  __shared__ float tile[32][N_MELS + 1];
  __shared__ float red[16];
  const int b = blockIdx.y, f0 = blockIdx.x * 32;
  // the window's maximum: the per-wave maxima of the first kernel (waves without a frame wrote nothing and are not counted)
  float m = -INFINITY;
  const int used = (F + FPW - 1) / FPW;
  for (int i = threadIdx.x; i < nslots; i += blockDim.x)
    if (i < used) m = fmaxf(m, wmax[(long)b * nslots + i]);
  // the block's 32 x 80 values: ten loads per thread, all in flight before the maximum is known
  constexpr int NE = 32 * N_MELS / 256;
  float x[NE];
  const long row0 = (long)b * F + f0;
  const int nvalid = min(32, F - f0) * N_MELS;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = threadIdx.x + 256 * i;
    x[i] = e < nvalid ? lm[row0 * N_MELS + e] : 0.f;
  }
  const float floor_v = block_max(m, red) - 8.f;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = threadIdx.x + 256 * i;
    const int fr = e / N_MELS, mi = e - fr * N_MELS;
    const float v = e < nvalid ? (fmaxf(x[i], floor_v) + 4.f) * 0.25f : 0.f;
    tile[fr][mi] = v;
    if (cl && e < nvalid) cl[((long)b * cl_rows + cl_lead + f0) * N_MELS + e] = (bf16)v;
  }
  __syncthreads();
  if (mel) {
    for (int e = threadIdx.x; e < 32 * N_MELS; e += blockDim.x) {
      const int mi = e >> 5, fr = e & 31;
      if (f0 + fr < F) mel[((long)b * N_MELS + mi) * F + f0 + fr] = tile[fr][mi];
    }
  }
}

double hz_to_mel(double f) { return f >= 1000.0 ? 15.0 + log(f / 1000.0) * (27.0 / log(6.4)) : 3.0 * f / 200.0; }
double mel_to_hz(double m) { return m >= 15.0 ? 1000.0 * exp(log(6.4) / 27.0 * (m - 15.0)) : 200.0 * m / 3.0; }

}  // namespace

extern "C" size_t ssak_logmel_table_floats(void) { return (size_t)MEL_OFF + (size_t)MEL_K * MEL_LD; }

// tables (host computes in double, like the reference's numpy path): the transform's twiddles and the periodic Hann window,
// then the mel filters [204][96]
extern "C" int ssak_logmel_init_tables(float* tables_dev) {
  SSAK_REQUIRE(tables_dev, "logmel_init_tables: null pointer");
  std::vector<float> h(ssak_logmel_table_floats(), 0.f);
  const double PI = 3.14159265358979323846;
  auto put = [&](int off, int idx, double ang) {
    h[off + 2 * idx] = (float)cos(ang);
    h[off + 2 * idx + 1] = (float)-sin(ang);  // e^{-i ang}
  };
  for (int k = 0; k < 5; ++k)
    for (int q = 1; q < 5; ++q) put(TAB_TW2, 4 * k + q - 1, 2.0 * PI * (q * k % 25) / 25.0);
  for (int k = 0; k < 25; ++k)
    for (int q = 1; q < 8; ++q) put(TAB_TW3, 7 * k + q - 1, 2.0 * PI * (q * k % 200) / 200.0);
  for (int k = 0; k <= N_FFT / 4; ++k) put(TAB_TWP, k, 2.0 * PI * k / N_FFT);
  for (int n = 0; n < N_FFT; ++n) h[TAB_HANN + n] = (float)(0.5 - 0.5 * cos(2.0 * PI * n / N_FFT));  // periodic Hann
  float* mf = h.data() + MEL_OFF;
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
      // the kernel multiplies a row tile of 16 filters with the bins of its band only
      SSAK_REQUIRE(v == 0.0 || (k >= MEL_K0[m / 16] && k < MEL_K1[m / 16]), "logmel_init_tables: a filter weight lies outside its compiled band");
    }
  }
  SSAK_HIP(hipMemcpy(tables_dev, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return SSAK_OK;
}

static inline int logmel_slots(int F) { return ssak_cdiv(F, FPW * WPB) * WPB; }

extern "C" size_t ssak_logmel_workspace_bytes(int B, int n_samples) {
  const size_t F = (size_t)n_samples / HOP;
  return ((size_t)B * F * N_MELS + 64 + (size_t)B * logmel_slots((int)F)) * sizeof(float);  // log10 mel energies before the clamp + the waves' maxima
}

extern "C" int ssak_logmel_whisper(const float* wav, const int32_t* lens, int B, int T, int n_samples, const float* tables,
                                   float* mel, void* mel_cl_bf16, int cl_rows, int cl_lead, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(wav && tables && workspace && (mel || mel_cl_bf16), "logmel: null pointer");
  SSAK_REQUIRE(B > 0 && T > 0 && n_samples >= N_FFT && n_samples % HOP == 0, "logmel: n_samples must be a multiple of 160 (>= 400)");
  SSAK_REQUIRE(workspace_bytes >= ssak_logmel_workspace_bytes(B, n_samples), "logmel: workspace too small");
  SSAK_REQUIRE(((uintptr_t)tables & 15) == 0, "logmel: tables must be 16-byte aligned");
  const int F = n_samples / HOP;          // 3000 frames kept (the STFT's last frame is dropped)
  SSAK_REQUIRE(!mel_cl_bf16 || cl_rows >= cl_lead + F, "logmel: channels-last copy too small");
  hipStream_t st = (hipStream_t)stream;
  // algorithmic bytes: the waveform read once, the features written once in each form asked for (2.88 MB / window)
  ProfScope prof_scope(PROF_LOGMEL, (double)B * ((double)n_samples * 4.0 + (double)F * N_MELS * ((mel ? 4.0 : 0.0) + (mel_cl_bf16 ? 2.0 : 0.0))), st);
  float* lm = (float*)workspace;
  float* wmax = lm + (size_t)B * F * N_MELS + 32;
  static const bool attr = [] {
    return hipFuncSetAttribute((const void*)stft_fft_mel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess;
  }();
  SSAK_REQUIRE(attr, "logmel: could not reserve the kernel's LDS");
  const int nslots = logmel_slots(F);
  stft_fft_mel_kernel<<<dim3(nslots / WPB, B), 64 * WPB, LDS_BYTES, st>>>(wav, lens, T, n_samples, tables, lm, F, wmax);
  SSAK_LAUNCH_CHECK();
  logmel_finalize_kernel<<<dim3(ssak_cdiv(F, 32), B), 256, 0, st>>>(lm, wmax, nslots, F, mel, (bf16*)mel_cl_bf16, cl_rows, cl_lead);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
