// CTC loss + gradient w.r.t. logits (fused log-softmax), and greedy CTC decode.  gfx950.
//
// Stands behind log_softmax(fp32) -> F.ctc_loss -> autograd of Wav2Vec2ForCTC.forward
// (transformers modeling_wav2vec2.py:1705-1728; configured by ssak/train/transformers/wav2vec_train.py:319,325)
// and behind argmax + batch_decode's collapse (ssak/infer/transformers_infer.py:84-85).
//
// One 512-thread workgroup per utterance.  Waves 0-3 run the alpha recursion forwards while waves 4-7 run
// the beta recursion backwards, one workgroup barrier per frame; the current lattice row lives in LDS
// (ping-pong), the full lattices go to an L2-resident scratch.  The gradient pass then walks the frames one
// wave per frame, lanes over lattice states, summing posteriors per symbol with LDS float atomics.
// Latency-bound by construction (F sequential frames); all arithmetic fp32 in the log domain.
#include "common.h"

namespace {

constexpr int CTC_THREADS = 512;
constexpr int CTC_HALF = 256;

__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  if (m == -INFINITY) return -INFINITY;
  return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

// LDS carve (dynamic): ext[Smax] int | row_a[2][Smax] | row_b[2][Smax] | bins[8][Vpad] | misc[16]
__global__ __launch_bounds__(CTC_THREADS) void ctc_kernel(const float* __restrict__ logits,
                                                          const int32_t* __restrict__ in_lens,
                                                          const int32_t* __restrict__ labels, int B, int F, int V,
                                                          int Lmax, int blank, int reduction, int zero_inf,
                                                          float grad_scale, float* __restrict__ nll_out,
                                                          float* __restrict__ wnll_out, float* __restrict__ dlogits,
                                                          float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Smax = 2 * Lmax + 1;
  const int Vpad = (V + 63) & ~63;
  int* ext = reinterpret_cast<int*>(smem);
  float* row_a = reinterpret_cast<float*>(ext + Smax);
  float* row_b = row_a + 2 * Smax;
  float* bins = row_b + 2 * Smax;
  float* misc = bins + 8 * Vpad;
  int* imisc = reinterpret_cast<int*>(misc + 8);

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int T = in_lens ? min(max(in_lens[b], 0), F) : F;
  const float* lg = logits + (size_t)b * F * V;
  float* lp = ws + (size_t)b * ((size_t)F * V + 2 * (size_t)F * Smax);
  float* A = lp + (size_t)F * V;
  float* Bt = A + (size_t)F * Smax;

  // ---- phase 0: log-softmax rows (one wave per frame) and label compaction (thread 0 of the last wave)
  for (int t = wave; t < T; t += CTC_THREADS / 64) {
    float mx = -INFINITY;
    for (int c = lane; c < V; c += 64) mx = fmaxf(mx, lg[(size_t)t * V + c]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int c = lane; c < V; c += 64) sm += __expf(lg[(size_t)t * V + c] - mx);
    sm = wave_sum(sm);
    const float lz = mx + __logf(sm);
    for (int c = lane; c < V; c += 64) lp[(size_t)t * V + c] = lg[(size_t)t * V + c] - lz;
  }
  if (tid == CTC_THREADS - 1) {
    int L = 0, bad_label = 0;
    for (int i = 0; i < Lmax; ++i) {
      int v = labels[(size_t)b * Lmax + i];
      if (v >= V) {  // "Label values must be <= vocab_size": flagged here, surfaced as a NaN loss (no host sync)
        bad_label = 1;
        v = blank;
      }
      if (v >= 0) {
        ext[2 * L] = blank;
        ext[2 * L + 1] = v;
        ++L;
      }
    }
    ext[2 * L] = blank;
    imisc[0] = L;
    imisc[2] = bad_label;
  }
  __syncthreads();
  const int L = imisc[0];
  const int S = 2 * L + 1;

  // ---- phase 1: alpha (threads 0..255) and beta (threads 256..511), one barrier per frame
  const bool is_beta = tid >= CTC_HALF;
  const int ht = tid & (CTC_HALF - 1);
  float* row = is_beta ? row_b : row_a;
  float* lat = is_beta ? Bt : A;
  for (int i = 0; i < T; ++i) {
    const int t = is_beta ? (T - 1 - i) : i;
    const float* prev = row + ((i + 1) & 1) * Smax;
    float* cur = row + (i & 1) * Smax;
    for (int s = ht; s < S; s += CTC_HALF) {
      const int c = ext[s];
      const float e = lp[(size_t)t * V + c];
      float v;
      if (i == 0) {
        if (!is_beta)
          v = (s <= 1) ? e : -INFINITY;
        else
          v = (s >= S - 2) ? e : -INFINITY;
      } else if (!is_beta) {
        const float a0 = prev[s];
        const float a1 = (s >= 1) ? prev[s - 1] : -INFINITY;
        const float a2 = (s >= 2 && (s & 1) && ext[s - 2] != c) ? prev[s - 2] : -INFINITY;
        v = lse3(a0, a1, a2) + e;
      } else {
        const float b0 = prev[s];
        const float b1 = (s + 1 < S) ? prev[s + 1] : -INFINITY;
        const float b2 = (s + 2 < S && (s & 1) && ext[s + 2] != c) ? prev[s + 2] : -INFINITY;
        v = lse3(b0, b1, b2) + e;
      }
      cur[s] = v;
      lat[(size_t)t * Smax + s] = v;
    }
    __syncthreads();
  }

  // ---- log-likelihood
  if (tid == 0) {
    float ll;
    if (T == 0)
      ll = (L == 0) ? 0.f : -INFINITY;
    else {
      const float* last = row_a + ((T - 1) & 1) * Smax;
      ll = (S == 1) ? last[0] : lse2(last[S - 1], last[S - 2]);
    }
    float nll = -ll;
    const bool bad = !(nll < INFINITY);  // inf or nan
    float w = (reduction == SSAK_REDUCTION_MEAN) ? 1.f / ((float)max(L, 1) * (float)B) : 1.f;
    if (bad && zero_inf) nll = 0.f;
    if (imisc[2]) nll = NAN;
    misc[0] = ll;
    misc[1] = w;
    imisc[1] = bad ? 1 : 0;
    if (nll_out) nll_out[b] = nll;
    wnll_out[b] = nll * w;
  }
  __syncthreads();
  if (!dlogits) return;
  const float ll = misc[0];
  const float w = misc[1] * grad_scale;
  const bool bad = imisc[1] != 0;

  // ---- phase 2: gradient, one wave per frame
  float* gb = dlogits + (size_t)b * F * V;
  float* mybins = bins + wave * Vpad;
  for (int c = lane; c < Vpad; c += 64) mybins[c] = 0.f;
  for (int t = wave; t < F; t += CTC_THREADS / 64) {
    if (t >= T || bad) {
      for (int c = lane; c < V; c += 64) gb[(size_t)t * V + c] = 0.f;
      continue;
    }
    float blank_sum = 0.f;
    for (int s = lane; s < S; s += 64) {
      const int c = ext[s];
      const float term = __expf(A[(size_t)t * Smax + s] + Bt[(size_t)t * Smax + s] - lp[(size_t)t * V + c] - ll);
      if (s & 1)
        atomicAdd(&mybins[c], term);
      else
        blank_sum += term;  // every even state is the blank: reduce across lanes instead of 32-way atomics
    }
    blank_sum = wave_sum(blank_sum);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) atomicAdd(&mybins[blank], blank_sum);
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS atomics of this wave have landed
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < V; c += 64) {
      gb[(size_t)t * V + c] = (__expf(lp[(size_t)t * V + c]) - mybins[c]) * w;
      mybins[c] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void ctc_sum_kernel(const float* __restrict__ wnll, int B, float* __restrict__ loss) {
  __shared__ float red[16];
  float v = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) v += wnll[i];
  v = block_sum(v, red);
  if (threadIdx.x == 0) loss[0] = v;
}

// greedy decode: one wave per utterance; frames in chunks of 64, lanes = frames
__global__ void greedy_kernel(const float* __restrict__ logits, const int32_t* __restrict__ in_lens, int B, int F, int V,
                              int blank, int32_t* __restrict__ ids, int32_t* __restrict__ out_lens) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int T = in_lens ? min(max(in_lens[b], 0), F) : F;
  int count = 0;
  int prev_last = -1;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    int best = -1;
    if (t < T) {
      const float* r = logits + ((size_t)b * F + t) * V;
      float bv = r[0];
      best = 0;
      for (int c = 1; c < V; ++c) {
        const float x = r[c];
        if (x > bv) {  // first maximum wins, as torch.argmax
          bv = x;
          best = c;
        }
      }
    }
    int prev = __shfl_up(best, 1, 64);
    if (lane == 0) prev = prev_last;
    const bool emit = (t < T) && best != prev && best != blank;
    const unsigned long long m = __ballot(emit);
    const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
    if (emit) ids[(size_t)b * F + pos] = best;
    count += __popcll(m);
    prev_last = __shfl(best, 63, 64);
  }
  for (int i = count + lane; i < F; i += 64) ids[(size_t)b * F + i] = -1;
  if (lane == 0) out_lens[b] = count;
}

size_t ctc_lds_bytes(int V, int Lmax) {
  const int Smax = 2 * Lmax + 1;
  const int Vpad = (V + 63) & ~63;
  return (size_t)Smax * 4 + 4 * (size_t)Smax * 4 + 8 * (size_t)Vpad * 4 + 16 * 4;
}

}  // namespace

extern "C" size_t ssak_ctc_workspace_bytes(int B, int F, int V, int Lmax) {
  const size_t Smax = 2 * (size_t)Lmax + 1;
  return ((size_t)B * ((size_t)F * V + 2 * (size_t)F * Smax) + (size_t)B) * sizeof(float);
}

extern "C" int ssak_ctc_loss_fwd_bwd(const float* logits, const int32_t* in_lens, const int32_t* labels, int B, int F,
                                     int V, int Lmax, int blank, int reduction, int zero_infinity, float grad_scale,
                                     float* loss, float* nll, float* dlogits, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  SSAK_REQUIRE(logits && labels && loss && workspace, "ctc: null pointer");
  SSAK_REQUIRE(B > 0 && F > 0 && V > 0 && Lmax >= 0, "ctc: bad shape B=%d F=%d V=%d Lmax=%d", B, F, V, Lmax);
  SSAK_REQUIRE(blank >= 0 && blank < V, "ctc: blank %d outside [0,%d)", blank, V);
  SSAK_REQUIRE(reduction == SSAK_REDUCTION_SUM || reduction == SSAK_REDUCTION_MEAN, "ctc: bad reduction %d", reduction);
  SSAK_REQUIRE(workspace_bytes >= ssak_ctc_workspace_bytes(B, F, V, Lmax), "ctc: workspace too small");
  const size_t lds = ctc_lds_bytes(V, Lmax);
  SSAK_REQUIRE(lds <= 160 * 1024, "ctc: Lmax=%d needs %zu B of LDS (> 160 KiB)", Lmax, lds);
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  float* wnll = ws + (size_t)B * ((size_t)F * V + 2 * (size_t)F * (2 * (size_t)Lmax + 1));
  if (lds > 64 * 1024)
    SSAK_HIP(hipFuncSetAttribute((const void*)ctc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ctc_kernel<<<B, CTC_THREADS, lds, st>>>(logits, in_lens, labels, B, F, V, Lmax, blank, reduction, zero_infinity,
                                          grad_scale, nll, wnll, dlogits, ws);
  SSAK_LAUNCH_CHECK();
  ctc_sum_kernel<<<1, 256, 0, st>>>(wnll, B, loss);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

extern "C" int ssak_ctc_greedy_decode(const float* logits, const int32_t* in_lens, int B, int F, int V, int blank,
                                      int32_t* ids, int32_t* out_lens, void* stream) {
  SSAK_REQUIRE(logits && ids && out_lens, "greedy: null pointer");
  SSAK_REQUIRE(B > 0 && F > 0 && V > 0, "greedy: bad shape");
  greedy_kernel<<<B, 64, 0, (hipStream_t)stream>>>(logits, in_lens, B, F, V, blank, ids, out_lens);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
