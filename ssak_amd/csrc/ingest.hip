// Audio ingest on the device: raw PCM frames -> mono fp32 -> model sample rate.
//
// Stands behind load_audio / conform_audio of ssak/utils/audio.py:24-154 for PCM input: the segment cut is a byte range
// chosen on the host (offset = int(start * sr), :85-92), the channel average is librosa.to_mono (:118), and the rate change
// is torchaudio.transforms.Resample(sr, 16000) with its defaults (:134) -- windowed-sinc interpolation
// ("sinc_interp_hann", lowpass_filter_width 6, rolloff 0.99) applied as a strided convolution with `new` polyphase
// filters of 2*width + orig taps (orig, new = the rates divided by their gcd).  The filter table is computed on the host
// in double and rounded to fp32 exactly as torchaudio builds it; the kernel is the polyphase FIR:
//     out[n * new + j] = sum_k h[j][k] * x[n * orig + k - width]      (x = 0 outside the utterance)
// HBM-bound (2 B in, 4 B out per sample; 15-475 MACs per output sample stay in LDS / L2).
#include <cmath>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int RS_BLOCK = 256;  // outputs per workgroup

// raw interleaved little-endian PCM (8-bit unsigned, 16-bit or 32-bit signed) -> mono fp32 in [-1, 1)
__global__ __launch_bounds__(256) void pcm_to_mono_kernel(const uint8_t* __restrict__ raw, const int64_t* __restrict__ byte_off,
                                                          const int32_t* __restrict__ nframes, int channels, int width, int Tmax,
                                                          float* __restrict__ out) {
  const int b = blockIdx.y;
  const int n = min(max(nframes[b], 0), Tmax);
  const uint8_t* src = raw + byte_off[b];
  float* dst = out + (size_t)b * Tmax;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Tmax; t += gridDim.x * blockDim.x) {
    float acc = 0.f;
    if (t < n) {
      for (int c = 0; c < channels; ++c) {
        const uint8_t* s = src + ((size_t)t * channels + c) * width;
        float v;
        if (width == 2)
          v = (float)(int16_t)((uint16_t)s[0] | ((uint16_t)s[1] << 8)) * (1.f / 32768.f);
        else if (width == 4)
          v = (float)(int32_t)((uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24)) * (1.f / 2147483648.f);
        else
          v = ((float)s[0] - 128.f) * (1.f / 128.f);
        acc += v;
      }
      if (channels > 1) acc /= (float)channels;  // np.mean over the channels, as librosa.to_mono
    }
    dst[t] = acc;
  }
}

// one workgroup = RS_BLOCK consecutive outputs of one utterance; the input span they touch is staged in LDS
__global__ __launch_bounds__(RS_BLOCK) void resample_kernel(const float* __restrict__ in, const int32_t* __restrict__ in_lens, int Tin,
                                                            int orig, int neu, int width, int taps, const float* __restrict__ table,
                                                            float* __restrict__ out, int Tout, int32_t* __restrict__ out_lens) {
  extern __shared__ float span[];
  const int b = blockIdx.y;
  const int len = in_lens ? min(max(in_lens[b], 0), Tin) : Tin;
  const long olen = ((long)neu * len + orig - 1) / orig;  // ceil(new * length / orig)
  if (blockIdx.x == 0 && threadIdx.x == 0 && out_lens) out_lens[b] = (int32_t)min(olen, (long)Tout);
  const long m0 = (long)blockIdx.x * RS_BLOCK;
  if (m0 >= Tout) return;
  const long n0 = m0 / neu, n1 = min(m0 + RS_BLOCK - 1, (long)Tout - 1) / neu;
  const long x0 = n0 * orig - width;                 // first input index of the span
  const int nspan = (int)((n1 - n0) * orig + taps);  // inputs the block needs
  const float* src = in + (size_t)b * Tin;
  for (int i = threadIdx.x; i < nspan; i += RS_BLOCK) {
    const long xi = x0 + i;
    span[i] = (xi >= 0 && xi < len) ? src[xi] : 0.f;
  }
  __syncthreads();
  const long m = m0 + threadIdx.x;
  if (m >= Tout) return;
  float acc = 0.f;
  if (m < olen) {
    const int j = (int)(m % neu);
    const float* h = table + (size_t)j * taps;
    const float* x = span + (m / neu - n0) * orig;
    for (int k = 0; k < taps; ++k) acc = fmaf(h[k], x[k], acc);
  }
  out[(size_t)b * Tout + m] = acc;
}

long gcd_l(long a, long b) { return b ? gcd_l(b, a % b) : a; }

}  // namespace

// the polyphase filters of torchaudio.functional.resample (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99)
extern "C" int ssak_resample_plan(int orig_sr, int new_sr, int* orig_r, int* new_r, int* width, int* taps) {
  SSAK_REQUIRE(orig_sr > 0 && new_sr > 0 && orig_r && new_r && width && taps, "resample_plan: bad arguments");
  const long g = gcd_l(orig_sr, new_sr);
  const int o = (int)(orig_sr / g), n = (int)(new_sr / g);
  const double base = std::min(o, n) * 0.99;
  *orig_r = o;
  *new_r = n;
  *width = (int)std::ceil(6.0 * o / base);
  *taps = 2 * *width + o;
  return SSAK_OK;
}

extern "C" int ssak_resample_table(int orig_sr, int new_sr, float* table /*host, [new_r][taps]*/) {
  int o, n, w, taps;
  if (int rc = ssak_resample_plan(orig_sr, new_sr, &o, &n, &w, &taps)) return rc;
  SSAK_REQUIRE(table, "resample_table: null table");
  const double lpw = 6.0, base = std::min(o, n) * 0.99, scale = base / o, pi = 3.14159265358979323846;
  for (int j = 0; j < n; ++j)
    for (int k = 0; k < taps; ++k) {
      // (-j / new is a float32 division in torchaudio -- an int64 arange divided by an int -- before it meets the float64 index)
      double t = ((double)((float)(-j) / (float)n) + (double)(k - w) / o) * base;
      t = std::max(-lpw, std::min(lpw, t));
      const double c = std::cos(t * pi / lpw / 2.0);
      const double window = c * c;
      t *= pi;
      const double s = (t == 0.0) ? 1.0 : std::sin(t) / t;
      table[(size_t)j * taps + k] = (float)(s * window * scale);
    }
  return SSAK_OK;
}

extern "C" int ssak_pcm_to_mono_f32(const void* raw, const int64_t* byte_offsets, const int32_t* nframes, int B, int channels,
                                    int sample_width, int Tmax, float* out, void* stream) {
  SSAK_REQUIRE(raw && byte_offsets && nframes && out, "pcm_to_mono: null pointer");
  SSAK_REQUIRE(B > 0 && Tmax > 0 && channels >= 1 && channels <= 16, "pcm_to_mono: bad shape B=%d Tmax=%d channels=%d", B, Tmax, channels);
  SSAK_REQUIRE(sample_width == 1 || sample_width == 2 || sample_width == 4, "pcm_to_mono: sample width %d (bytes) not supported", sample_width);
  pcm_to_mono_kernel<<<dim3(std::min(ssak_cdiv(Tmax, 256), 1024), B), 256, 0, (hipStream_t)stream>>>(
      (const uint8_t*)raw, byte_offsets, nframes, channels, sample_width, Tmax, out);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

extern "C" int ssak_resample_sinc(const float* in, const int32_t* in_lens, int B, int Tin, int orig_sr, int new_sr, const float* table,
                                  float* out, int Tout, int32_t* out_lens, void* stream) {
  SSAK_REQUIRE(in && table && out, "resample: null pointer");
  SSAK_REQUIRE(B > 0 && Tin > 0 && Tout > 0, "resample: bad shape");
  int o, n, w, taps;
  if (int rc = ssak_resample_plan(orig_sr, new_sr, &o, &n, &w, &taps)) return rc;
  const size_t lds = ((size_t)(RS_BLOCK / n + 2) * o + taps) * sizeof(float);
  SSAK_REQUIRE(lds <= 64 * 1024, "resample: %d -> %d Hz needs %zu B of LDS per workgroup", orig_sr, new_sr, lds);
  resample_kernel<<<dim3(ssak_cdiv(Tout, RS_BLOCK), B), RS_BLOCK, lds, (hipStream_t)stream>>>(in, in_lens, Tin, o, n, w, taps, table, out, Tout,
                                                                                         out_lens);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
