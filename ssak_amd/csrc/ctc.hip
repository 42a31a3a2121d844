// CTC loss + gradient w.r.t. logits (fused log-softmax), and greedy CTC decode.  gfx950.
//
// Stands behind log_softmax(fp32) -> F.ctc_loss -> autograd of Wav2Vec2ForCTC.forward
// (transformers modeling_wav2vec2.py:1705-1728; configured by ssak/train/transformers/wav2vec_train.py:319,325)
// and behind argmax + batch_decode's collapse (ssak/infer/transformers_infer.py:84-85).
//
// Three launches per step: ctc_lsm (log-softmax, one wave per (utterance, frame) over the whole chip), ctc_lat<SPL> (the
// alpha / beta recursion, WAVE-RESIDENT: one workgroup per utterance, alpha on wave 0 and beta on wave 1, a lane owns
// SPL = 4 / 8 / 16 consecutive lattice states in registers, the neighbour's edge state arrives by one DPP wave shift per
// frame, no LDS and no barrier inside the frame loop, log-probabilities gathered 8 frames ahead into a register ring) and
// ctc_grad (one wave per (utterance, frame), lanes over lattice states; posteriors per symbol are summed in wave-private LDS
// bins with LDS float atomics -- order within a wave-instruction is fixed by the hardware, so runs are bit-reproducible in
// practice, but it is an atomic, not a fixed-order tree).  Label sequences beyond 511 tokens fall back to the monolithic
// kernel at the end of this file: one 512-thread workgroup per utterance, waves 0-3 alpha / waves 4-7 beta, one barrier per
// frame, lattice rows in LDS (ping-pong), full lattices in an L2-resident scratch.
// Latency-bound by construction (F sequential frames); all arithmetic fp32 in the log domain.
#include <stdlib.h>

#include "kernels.h"

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

// branch-free log-sum-exp of three (the lattice kernel's inner operation): all -inf in -> -inf out without a divergent
// early return (exp2(-inf) = 0, log2(0) = -inf), raw v_exp_f32 / v_log_f32
__device__ __forceinline__ float lse3_nb(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  const float ms = (m == -INFINITY) ? 0.f : m;
  const float k = -ms * 1.4426950408889634f;
  const float s = __builtin_amdgcn_exp2f(fmaf(a, 1.4426950408889634f, k)) + __builtin_amdgcn_exp2f(fmaf(b, 1.4426950408889634f, k)) +
                  __builtin_amdgcn_exp2f(fmaf(c, 1.4426950408889634f, k));
  return fmaf(__builtin_amdgcn_logf(s), 0.6931471805599453f, ms);
}

// ------------------------------------------------------------------------------------------------------------------
// Wave-resident lattice (the path taken whenever 2 * Lmax + 1 <= 1024).  The monolithic kernel above spends its time on
// things that are not the recursion: the log-softmax and the gradient are run by 8 waves over ~500 frames, and each of the
// ~500 lattice steps pays a workgroup barrier, an LDS round trip and an un-prefetched gather of the log-probabilities.
// Here the work is three launches:
//   ctc_lsm_kernel   log-softmax, one wave per (utterance, frame)                       -- all CUs
//   ctc_lat_kernel   alpha on wave 0, beta on wave 1 of one workgroup per utterance.  A lane owns SPL consecutive lattice
//                    states in registers; the only cross-lane traffic is the neighbour's edge state, one DPP wave shift
//                    per step (two for beta).  No LDS, no barrier in the loop; the log-probabilities of step t+1 are
//                    gathered while step t is computed; the row is stored with one 16-byte store per lane.
//   ctc_grad_kernel  posterior sums and the gradient, one wave per (utterance, frame)    -- all CUs
// Same arithmetic as above (fp32 log domain, lse3 then + log-probability), so the same parity bars apply.
constexpr int DPP_WAVE_SHR1 = 0x138, DPP_WAVE_SHL1 = 0x130;

__global__ __launch_bounds__(256) void ctc_lsm_kernel(const float* __restrict__ logits, const int32_t* __restrict__ in_lens, int B, int F,
                                                      int V, float* __restrict__ lp) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * F) return;
  const int b = (int)(row / F), t = (int)(row % F);
  const int T = in_lens ? min(max(in_lens[b], 0), F) : F;
  if (t >= T) return;
  const float* lg = logits + row * V;
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, lg[c]);
  mx = wave_max(mx);
  float sm = 0.f;
  for (int c = lane; c < V; c += 64) sm += __expf(lg[c] - mx);
  sm = wave_sum(sm);
  const float lz = mx + __logf(sm);
  for (int c = lane; c < V; c += 64) lp[row * V + c] = lg[c] - lz;
}

// info per utterance (floats): ll | w | bad | L
template <int SPL>
__global__ __launch_bounds__(128) void ctc_lat_kernel(const float* __restrict__ lp_all, const int32_t* __restrict__ in_lens,
                                                      const int32_t* __restrict__ labels, int B, int F, int V, int Lmax, int Spad,
                                                      int blank, int reduction, int zero_inf, float* __restrict__ nll_out,
                                                      float* __restrict__ wnll_out, float* __restrict__ lat_all,
                                                      float* __restrict__ info) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const bool is_beta = threadIdx.x >= 64;
  const int T = in_lens ? min(max(in_lens[b], 0), F) : F;
  const float* lp = lp_all + (size_t)b * F * V;
  float* lat = lat_all + ((size_t)b * 2 + (is_beta ? 1 : 0)) * (size_t)F * Spad;
  __shared__ int s_ext[2 * 512 + 2];  // extended label sequence (SPL <= 16: S <= 1024)
  __shared__ int s_misc[4];
  // label compaction by one thread (tens of labels), as in the monolithic kernel
  if (threadIdx.x == 0) {
    int L = 0, bad_label = 0;
    for (int i = 0; i < Lmax; ++i) {
      int v = labels[(size_t)b * Lmax + i];
      if (v >= V) {
        bad_label = 1;
        v = blank;
      }
      if (v >= 0) {
        s_ext[2 * L] = blank;
        s_ext[2 * L + 1] = v;
        ++L;
      }
    }
    s_ext[2 * L] = blank;
    s_misc[0] = L;
    s_misc[1] = bad_label;
  }
  __syncthreads();
  const int L = s_misc[0], S = 2 * L + 1;
  // this lane's states s = SPL * lane + k; odd k are labels.  lab[h] = class of state 2h+1; skip[h] = the 2-step move is allowed
  int lab[SPL / 2];
  bool skip[SPL / 2];
#pragma unroll
  for (int h = 0; h < SPL / 2; ++h) {
    const int s = SPL * lane + 2 * h + 1;
    lab[h] = s < S ? s_ext[s] : blank;
    if (!is_beta)
      skip[h] = s < S && s >= 2 && s_ext[s - 2] != lab[h];
    else
      skip[h] = s + 2 < S && s_ext[s + 2] != lab[h];
  }
  // states beyond S are kept at -inf by an additive penalty instead of a select: a select of the whole log-sum-exp is
  // compiled into a divergent branch per state, which serialises the lane's SPL independent chains
  float pen[SPL];
#pragma unroll
  for (int k = 0; k < SPL; ++k) pen[k] = (SPL * lane + k < S) ? 0.f : -INFINITY;
  float v[SPL];
  // log-probabilities are gathered PD steps ahead into a register ring: a step is ~0.15 us of arithmetic, an L2 round trip
  // several times that, and waiting for the gather of the very next step made the recursion run at memory latency
  constexpr int PD = 8;
  float ring_lab[PD][SPL / 2], ring_blank[PD];
  const int Tc = max(T, 1);  // T == 0: the loop does not run, the priming gathers read row 0 (F >= 1)
  auto frame_of = [&](int i) { return is_beta ? (Tc - 1 - i) : i; };
  auto gather = [&](int t, float* el, float& eb) {
    const float* r = lp + (size_t)t * V;
    eb = r[blank];
#pragma unroll
    for (int h = 0; h < SPL / 2; ++h) el[h] = r[lab[h]];
  };
#pragma unroll
  for (int u = 0; u < PD; ++u) {
    ring_blank[u] = 0.f;
#pragma unroll
    for (int h = 0; h < SPL / 2; ++h) ring_lab[u][h] = 0.f;
    gather(frame_of(min(u, Tc - 1)), ring_lab[u], ring_blank[u]);
  }
  for (int i0 = 0; i0 < T; i0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int i = i0 + u;
      if (i >= T) break;
      const int t = frame_of(i);
      float e_lab[SPL / 2];
#pragma unroll
      for (int h = 0; h < SPL / 2; ++h) e_lab[h] = ring_lab[u][h];
      const float e_blank = ring_blank[u];
      // unconditional (index clamped): a conditional gather merges "loaded" and "old" in a copy placed right after the
      // load, which makes the wave wait for the load it has just issued
      gather(frame_of(min(i + PD, Tc - 1)), ring_lab[u], ring_blank[u]);
      if (i == 0) {
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
          const int s = SPL * lane + k;
          const float e = (k & 1) ? e_lab[k >> 1] : e_blank;
          const bool on = is_beta ? (s >= S - 2 && s < S) : (s <= 1 && s < S);
          v[k] = on ? e : -INFINITY;
        }
      } else if (!is_beta) {
        // the left neighbour's last state feeds this lane's first two (s-1 of k = 0, s-2 of k = 1)
        const float left = dpp_f<DPP_WAVE_SHR1, 0xf>(-INFINITY, v[SPL - 1]);
        float nv[SPL];
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
          const float a0 = v[k];
          const float a1 = k >= 1 ? v[k - 1] : left;
          float a2 = -INFINITY;
          if (k & 1) a2 = skip[k >> 1] ? (k >= 2 ? v[k - 2] : left) : -INFINITY;
          const float e = ((k & 1) ? e_lab[k >> 1] : e_blank) + pen[k];
          nv[k] = lse3_nb(a0, a1, a2) + e;
        }
#pragma unroll
        for (int k = 0; k < SPL; ++k) v[k] = nv[k];
      } else {
        // the right neighbour's first two states feed this lane's last one (s+1 and s+2 of k = SPL-1)
        const float r0 = dpp_f<DPP_WAVE_SHL1, 0xf>(-INFINITY, v[0]);
        const float r1 = dpp_f<DPP_WAVE_SHL1, 0xf>(-INFINITY, v[1]);
        float nv[SPL];
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
          const float b0 = v[k];
          const float b1 = k + 1 < SPL ? v[k + 1] : r0;
          float b2 = -INFINITY;
          if (k & 1) b2 = skip[k >> 1] ? (k + 2 < SPL ? v[k + 2] : r1) : -INFINITY;
          const float e = ((k & 1) ? e_lab[k >> 1] : e_blank) + pen[k];
          nv[k] = lse3_nb(b0, b1, b2) + e;
        }
#pragma unroll
        for (int k = 0; k < SPL; ++k) v[k] = nv[k];
      }
      // row t of the lattice: SPL consecutive floats per lane, 16-byte stores (Spad is a multiple of 4)
      if (SPL * lane < Spad) {
        float* dst = lat + (size_t)t * Spad + SPL * lane;
#pragma unroll
        for (int k = 0; k < SPL; k += 4)
          if (SPL * lane + k < Spad) *reinterpret_cast<f32x4*>(dst + k) = (f32x4){v[k], v[k + 1], v[k + 2], v[k + 3]};
      }
    }
  }
  // ---- log-likelihood from the last alpha row (wave 0): states S-1 and S-2 sit in one or two lanes
  if (!is_beta) {
    float a_last = -INFINITY, a_prev = -INFINITY;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      const int s = SPL * lane + k;
      if (s == S - 1) a_last = v[k];
      if (s == S - 2) a_prev = v[k];
    }
    a_last = wave_max(a_last);  // exactly one lane holds each (the others are -inf)
    a_prev = wave_max(a_prev);
    if (lane == 0) {
      float ll;
      if (T == 0)
        ll = (L == 0) ? 0.f : -INFINITY;
      else
        ll = (S == 1) ? a_last : lse2(a_last, a_prev);
      float nll = -ll;
      const bool bad = !(nll < INFINITY);
      const float w = (reduction == SSAK_REDUCTION_MEAN) ? 1.f / ((float)max(L, 1) * (float)B) : 1.f;
      if (bad && zero_inf) nll = 0.f;
      if (s_misc[1]) nll = NAN;
      info[b * 4 + 0] = ll;
      info[b * 4 + 1] = w;
      info[b * 4 + 2] = bad ? 1.f : 0.f;
      info[b * 4 + 3] = (float)L;
      if (nll_out) nll_out[b] = nll;
      wnll_out[b] = nll * w;
    }
  }
}

// gradient: one wave per (utterance, frame); 4 waves per workgroup share the extended label sequence in LDS
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ lp_all, const int32_t* __restrict__ in_lens,
                                                       const int32_t* __restrict__ labels, int B, int F, int V, int Lmax, int Spad,
                                                       int blank, float grad_scale, const float* __restrict__ lat_all,
                                                       const float* __restrict__ info, float* __restrict__ dlogits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Vpad = (V + 63) & ~63;
  int* ext = reinterpret_cast<int*>(smem);                    // [2 * Lmax + 1]
  float* bins = reinterpret_cast<float*>(ext + 2 * Lmax + 2);  // [4][Vpad]
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int T = in_lens ? min(max(in_lens[b], 0), F) : F;
  const int L = (int)info[b * 4 + 3];
  const int S = 2 * L + 1;
  // the same compaction as the lattice kernel, done cooperatively: position of each kept label = prefix count
  if (threadIdx.x == 0) {
    int n = 0;
    for (int i = 0; i < Lmax; ++i) {
      int v = labels[(size_t)b * Lmax + i];
      if (v >= V) v = blank;
      if (v >= 0) {
        ext[2 * n] = blank;
        ext[2 * n + 1] = v;
        ++n;
      }
    }
    ext[2 * n] = blank;
  }
  float* mybins = bins + wave * Vpad;
  for (int c = lane; c < Vpad; c += 64) mybins[c] = 0.f;
  __syncthreads();
  const float ll = info[b * 4 + 0];
  const float w = info[b * 4 + 1] * grad_scale;
  const bool bad = info[b * 4 + 2] != 0.f;
  const float* lp = lp_all + (size_t)b * F * V;
  const float* A = lat_all + (size_t)b * 2 * (size_t)F * Spad;
  const float* Bt = A + (size_t)F * Spad;
  float* gb = dlogits + (size_t)b * F * V;
  const int t0 = blockIdx.x * 16;
  for (int t = t0 + wave; t < min(t0 + 16, F); t += 4) {
    if (t >= T || bad) {
      for (int c = lane; c < V; c += 64) gb[(size_t)t * V + c] = 0.f;
      continue;
    }
    float blank_sum = 0.f;
    for (int s = lane; s < S; s += 64) {
      const int c = ext[s];
      const float term = __expf(A[(size_t)t * Spad + s] + Bt[(size_t)t * Spad + s] - lp[(size_t)t * V + c] - ll);
      if (s & 1)
        atomicAdd(&mybins[c], term);
      else
        blank_sum += term;
    }
    blank_sum = wave_sum(blank_sum);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) atomicAdd(&mybins[blank], blank_sum);
    __builtin_amdgcn_s_waitcnt(0xc07f);
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
  const size_t Spad = (2 * (size_t)Lmax + 1 + 3) & ~(size_t)3;  // lattice rows padded to 16 bytes
  return ((size_t)B * ((size_t)F * V + 2 * (size_t)F * Spad) + (size_t)B * 5) * sizeof(float);
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
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof_scope(PROF_CTC, (double)B * F * V * 4.0 * (dlogits ? 2 : 1), st);  // logits in, gradient out
  float* ws = (float*)workspace;
  const int Smax = 2 * Lmax + 1;
  static const bool env_mono = SSAK_DEV_ENV("SSAK_CTC_MONOLITHIC") != nullptr;  // development switch
  if (Smax <= 1024 && Lmax >= 1 && !env_mono) {
    // wave-resident lattice: lp [B,F,V] | lattices [B][2][F][Spad] | wnll [B] | info [B][4]
    const int Spad = (Smax + 3) & ~3;
    float* lp = ws;
    float* lat = lp + (size_t)B * F * V;
    float* wnll = lat + (size_t)B * 2 * (size_t)F * Spad;
    float* info = wnll + B;
    ctc_lsm_kernel<<<ssak_cdiv((long)B * F, 4), 256, 0, st>>>(logits, in_lens, B, F, V, lp);
    SSAK_LAUNCH_CHECK();
    if (Smax <= 256)
      ctc_lat_kernel<4><<<B, 128, 0, st>>>(lp, in_lens, labels, B, F, V, Lmax, Spad, blank, reduction, zero_infinity, nll, wnll, lat, info);
    else if (Smax <= 512)
      ctc_lat_kernel<8><<<B, 128, 0, st>>>(lp, in_lens, labels, B, F, V, Lmax, Spad, blank, reduction, zero_infinity, nll, wnll, lat, info);
    else
      ctc_lat_kernel<16><<<B, 128, 0, st>>>(lp, in_lens, labels, B, F, V, Lmax, Spad, blank, reduction, zero_infinity, nll, wnll, lat, info);
    SSAK_LAUNCH_CHECK();
    if (dlogits) {
      const size_t lds = (size_t)(2 * Lmax + 2) * 4 + 4 * (size_t)((V + 63) & ~63) * 4;
      SSAK_REQUIRE(lds <= 64 * 1024, "ctc: V=%d needs %zu B of LDS in the gradient pass", V, lds);
      ctc_grad_kernel<<<dim3(ssak_cdiv(F, 16), B), 256, lds, st>>>(lp, in_lens, labels, B, F, V, Lmax, Spad, blank, grad_scale, lat,
                                                                  info, dlogits);
      SSAK_LAUNCH_CHECK();
    }
    ctc_sum_kernel<<<1, 256, 0, st>>>(wnll, B, loss);
    SSAK_LAUNCH_CHECK();
    return SSAK_OK;
  }
  const size_t lds = ctc_lds_bytes(V, Lmax);
  SSAK_REQUIRE(lds <= 160 * 1024, "ctc: Lmax=%d needs %zu B of LDS (> 160 KiB)", Lmax, lds);
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
