// a1: per-utterance waveform normalisation (zero mean / unit variance, right zero padding).  gfx950.
//
// Stands behind Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm (transformers
// feature_extraction_wav2vec2.py:78-97), reached from ssak/utils/dataset.py:632 and
// ssak/infer/transformers_infer.py:216.  HBM-bound: 4 B/sample read + 4 B/sample written.
//
// Pass 1: every workgroup takes one 8192-sample chunk of one utterance into registers (float4 loads),
// computes the chunk's (count, mean, M2) with a two-pass in-register reduction and writes it out.
// Pass 2: every workgroup Chan-merges its utterance's chunk statistics (<= a few hundred triples), then
// re-reads its chunk (Infinity-Cache resident) and writes the normalised samples / zero padding / mask.
#include "kernels.h"

namespace {

constexpr int NCHUNK = 8192;  // samples per workgroup
constexpr int NTHREADS = 256;
constexpr int PER_THREAD = NCHUNK / NTHREADS;  // 32 samples = 8 x float4

__global__ __launch_bounds__(NTHREADS) void norm_stats_kernel(const float* __restrict__ in,
                                                             const int32_t* __restrict__ lens, int T, int nchunks,
                                                             float* __restrict__ stats) {
  __shared__ float red[16];
  const int b = blockIdx.y, ch = blockIdx.x;
  const int len = lens ? min(max(lens[b], 0), T) : T;
  const int base = ch * NCHUNK;
  const float* x = in + (size_t)b * T;
  float v[PER_THREAD];
  float s = 0.f;
  const bool vec_ok = ((T & 3) == 0);
#pragma unroll
  for (int j = 0; j < PER_THREAD / 4; ++j) {
    const int i = base + (j * NTHREADS + threadIdx.x) * 4;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec_ok && i + 3 < len) {
      q = *reinterpret_cast<const float4*>(x + i);
    } else {
      if (i < len) q.x = x[i];
      if (i + 1 < len) q.y = x[i + 1];
      if (i + 2 < len) q.z = x[i + 2];
      if (i + 3 < len) q.w = x[i + 3];
    }
    v[4 * j] = q.x;
    v[4 * j + 1] = q.y;
    v[4 * j + 2] = q.z;
    v[4 * j + 3] = q.w;
    s += (q.x + q.y) + (q.z + q.w);
  }
  const int n = max(0, min(NCHUNK, len - base));
  const float mean = (n > 0) ? block_sum(s, red) / (float)n : 0.f;
  float m2 = 0.f;
#pragma unroll
  for (int j = 0; j < PER_THREAD / 4; ++j) {
    const int i = base + (j * NTHREADS + threadIdx.x) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k < len) {
        const float d = v[4 * j + k] - mean;
        m2 += d * d;
      }
  }
  m2 = block_sum(m2, red);
  if (threadIdx.x == 0) {
    float* o = stats + ((size_t)b * nchunks + ch) * 3;
    o[0] = (float)n;
    o[1] = mean;
    o[2] = m2;
  }
}

__global__ __launch_bounds__(NTHREADS) void norm_apply_kernel(const float* __restrict__ in,
                                                             const int32_t* __restrict__ lens, int T, int nchunks,
                                                             const float* __restrict__ stats, float* __restrict__ out,
                                                             int32_t* __restrict__ mask) {
  __shared__ float sh[2];
  const int b = blockIdx.y, ch = blockIdx.x;
  const int len = lens ? min(max(lens[b], 0), T) : T;
  if (threadIdx.x < 64) {
    // Chan et al. pairwise merge, in double: lanes stride over the chunks, then a butterfly over lanes
    double n = 0.0, mu = 0.0, m2 = 0.0;
    for (int c = threadIdx.x; c < nchunks; c += 64) {
      const float* s = stats + ((size_t)b * nchunks + c) * 3;
      const double nb = s[0], mb = s[1], qb = s[2];
      if (nb > 0.0) {
        const double nt = n + nb, d = mb - mu;
        mu += d * nb / nt;
        m2 += qb + d * d * n * nb / nt;
        n = nt;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mu, o, 64), qb = __shfl_xor(m2, o, 64);
      const double nt = n + nb;
      if (nt > 0.0) {
        const double d = mb - mu;
        const double mun = (n * mu + nb * mb) / nt;
        m2 = m2 + qb + d * d * n * nb / nt;
        mu = mun;
      }
      n = nt;
    }
    if (threadIdx.x == 0) {
      const double var = (n > 0.0) ? m2 / n : 0.0;
      sh[0] = (float)mu;
      sh[1] = (float)(1.0 / sqrt(var + 1e-7));
    }
  }
  __syncthreads();
  const float mu = sh[0], rs = sh[1];
  const int base = ch * NCHUNK;
  const float* x = in + (size_t)b * T;
  float* y = out + (size_t)b * T;
  int32_t* mk = mask ? mask + (size_t)b * T : nullptr;
  const bool vec_ok = ((T & 3) == 0);
#pragma unroll
  for (int j = 0; j < PER_THREAD / 4; ++j) {
    const int i = base + (j * NTHREADS + threadIdx.x) * 4;
    if (i >= T) continue;
    if (vec_ok && i + 3 < len) {
      const float4 q = *reinterpret_cast<const float4*>(x + i);
      *reinterpret_cast<float4*>(y + i) = make_float4((q.x - mu) * rs, (q.y - mu) * rs, (q.z - mu) * rs, (q.w - mu) * rs);
      if (mk) *reinterpret_cast<int4*>(mk + i) = make_int4(1, 1, 1, 1);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i + k < T) {
          const bool in_r = (i + k) < len;
          y[i + k] = in_r ? (x[i + k] - mu) * rs : 0.f;
          if (mk) mk[i + k] = in_r ? 1 : 0;
        }
    }
  }
}

}  // namespace

extern "C" size_t ssak_wave_normalize_workspace_bytes(int B, int T) {
  return (size_t)B * ssak_cdiv(T, NCHUNK) * 3 * sizeof(float);
}

extern "C" int ssak_wave_normalize(const float* in, const int32_t* lens, int B, int T, float* out, int32_t* mask,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(in && out && workspace, "wave_normalize: null pointer");
  SSAK_REQUIRE(B > 0 && T > 0, "wave_normalize: bad shape B=%d T=%d", B, T);
  SSAK_REQUIRE(workspace_bytes >= ssak_wave_normalize_workspace_bytes(B, T), "wave_normalize: workspace too small");
  SSAK_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0, "wave_normalize: buffers must be 16-byte aligned");
  const int nch = ssak_cdiv(T, NCHUNK);
  dim3 grid(nch, B);
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof_scope(PROF_WAVE_NORM, (double)B * T * 8.0, st);  // 1.28 MB / 10 s utterance: read once, write once
  norm_stats_kernel<<<grid, NTHREADS, 0, st>>>(in, lens, T, nch, (float*)workspace);
  SSAK_LAUNCH_CHECK();
  norm_apply_kernel<<<grid, NTHREADS, 0, st>>>(in, lens, T, nch, (const float*)workspace, out, mask);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
