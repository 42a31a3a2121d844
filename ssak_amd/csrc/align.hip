// CTC forced alignment (Viterbi trellis + backtrack) for gfx950 -- SURVEY.md section 8f-1.
//
// Stands behind get_trellis / backtrack of ssak/utils/align_transcriptions.py:27-70,79-123 (module constants USE_MAX and
// USE_CHAR_REPEATED, :24-25), which the reference runs as a Python loop of torch ops over the frames.  The recursion is
// sequential in time and parallel over the transcript, so one workgroup owns one utterance: thread `tid` owns the
// trellis columns j = 1 + tid + NT * i, a trellis row lives in LDS (double buffered, one barrier per frame), every
// thread's emission gathers for frame t+1 are in flight while frame t is computed, and the whole [F+1, L+1] trellis is
// written to HBM once (row-contiguous, coalesced) because the reference returns it.  Each cell also leaves a 2-bit code
// (changed > stayed | changed < stayed) -- exactly the two comparisons the reference's backtrack re-derives from the
// trellis -- so the backtrack is a walk over one byte per frame.  All arithmetic is the reference's: fp32 add and max
// (bit-exact), column 0 = running sum of the blank log-probability accumulated in double and rounded per element
// (torch.cumsum's accumulation type on CPU), the +inf / -inf borders of :42-43 included.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int ALIGN_MAX_K = 16;  // trellis columns per thread (L <= 16 * 1024)

struct AlignParams {
  const float* em;      // [F, V] log-probabilities
  const int32_t* tokens;  // [L]
  const float* col0;    // [F+1] trellis column 0 given by the caller (first_as_garbage) or NULL = blank cumsum
  float* trellis;       // [F+1, ld] with ld >= L+1, or NULL (batched callers that only want the path)
  float* lastcol;       // [F+1] trellis column L (the backtrack's argmax reads it; kept whether or not the trellis is)
  uint8_t* bp;          // [F, L] codes, then [F] codes along the path
  int32_t* path_token;  // [F] indexed by time
  float* path_logp;     // [F] indexed by time
  int32_t* path_info;   // [2]: number of points (-1 = failed to align), time index of the first point
  int F, V, L, blank;
  int tld;              // row stride of the trellis
};

// One launch, many utterances (tools/align_audio_transcript.py aligns a Kaldi folder: the kernel is one workgroup per
// utterance and the chip has 256 CUs).  Padded layouts: emission [B, Fmax, V], tokens [B, Lmax], col0 [B, Fmax+1],
// trellis [B, Fmax+1, Lmax+1] (utterance b uses the top-left (F_b+1) x (L_b+1) corner), paths [B, Fmax], info [B, 2].
struct AlignBatch {
  const float* em;
  const int32_t* frame_lens;  // [B] or NULL (= Fmax)
  const int32_t* tokens;
  const int32_t* token_lens;  // [B] or NULL (= Lmax)
  const float* col0;
  float* trellis;
  float* lastcol;             // [B, Fmax+1]
  uint8_t* bp;                // [B] x (Fmax * Lmax + Fmax)
  int32_t* path_token;
  float* path_logp;
  int32_t* path_info;
  int Fmax, V, Lmax, blank;
};

__device__ __forceinline__ void align_one(const AlignParams& p, float* rows);

__global__ __launch_bounds__(1024) void align_kernel(const AlignBatch q) {
  extern __shared__ float rows[];  // 2 x (Lmax + 1) floats (>= 4 KiB: the backtrack's code band reuses it)
  const int b = blockIdx.x;
  AlignParams p;
  p.F = q.frame_lens ? q.frame_lens[b] : q.Fmax;
  p.L = q.token_lens ? q.token_lens[b] : q.Lmax;
  p.V = q.V;
  p.blank = q.blank;
  p.tld = q.Lmax + 1;
  p.em = q.em + (long)b * q.Fmax * q.V;
  p.tokens = q.tokens + (long)b * q.Lmax;
  p.col0 = q.col0 ? q.col0 + (long)b * (q.Fmax + 1) : nullptr;
  p.trellis = q.trellis ? q.trellis + (long)b * (q.Fmax + 1) * (q.Lmax + 1) : nullptr;
  p.lastcol = q.lastcol + (long)b * (q.Fmax + 1);
  p.bp = q.bp + (long)b * ((long)q.Fmax * q.Lmax + q.Fmax);
  p.path_token = q.path_token + (long)b * q.Fmax;
  p.path_logp = q.path_logp + (long)b * q.Fmax;
  p.path_info = q.path_info + 2 * b;
  if (p.F < 1 || p.F > q.Fmax || p.L < 1 || p.L > q.Lmax) {  // lengths outside the padded layout: refuse this utterance
    if (threadIdx.x == 0) {
      p.path_info[0] = -3;
      p.path_info[1] = 0;
    }
    return;
  }
  align_one(p, rows);
}

__device__ __forceinline__ void align_one(const AlignParams& p, float* rows) {
  __shared__ float red_v[16];
  __shared__ int red_i[16];
  const int tid = threadIdx.x, NT = blockDim.x;
  const int F = p.F, V = p.V, L = p.L, W = L + 1, TW = p.tld;
  const int K = (L + NT - 1) / NT;
  float* r0 = rows;
  float* r1 = rows + W;
  // rows r >= inf_from of column 0 are +inf (trellis[-L:, 0] = inf, align_transcriptions.py:43)
  const int inf_from = max(0, F + 1 - L);
  int tok[ALIGN_MAX_K];
  float et[ALIGN_MAX_K];
#pragma unroll
  for (int i = 0; i < ALIGN_MAX_K; ++i) {
    const int s = tid + NT * i;
    tok[i] = (i < K && s < L) ? p.tokens[s] : 0;
    et[i] = (i < K && s < L && tok[i] >= 0 && tok[i] < V) ? p.em[tok[i]] : 0.f;
  }
  // a token outside the vocabulary would index past the emission row: refuse the whole call (status in path_info[0])
  {
    int bad = 0;
#pragma unroll
    for (int i = 0; i < ALIGN_MAX_K; ++i) bad |= (tok[i] < 0 || tok[i] >= V);
    if (__syncthreads_or(bad)) {
      if (tid == 0) {
        p.path_info[0] = -2;
        p.path_info[1] = 0;
      }
      return;
    }
  }
  // row 0: trellis[0, 0] = 0 (or the caller's), trellis[0, 1:] = -inf
  for (int j = tid + 1; j < W; j += NT) {
    r0[j] = -INFINITY;
    if (p.trellis) p.trellis[j] = -INFINITY;
    if (j == L) p.lastcol[0] = -INFINITY;
  }
  double run = 0.0;  // thread 0: running sum of the blank log-probability
  if (tid == 0) {
    float c = p.col0 ? p.col0[0] : 0.f;
    if (0 >= inf_from) c = INFINITY;
    r0[0] = c;
    if (p.trellis) p.trellis[0] = c;
  }
  __syncthreads();
  for (int t = 0; t < F; ++t) {
    float* cur = (t & 1) ? r1 : r0;
    float* nxt = (t & 1) ? r0 : r1;
    const float eb = p.em[(long)t * V + p.blank];
    // gathers of the next frame first: their latency hides behind this frame's work and the barrier
    float etn[ALIGN_MAX_K];
#pragma unroll
    for (int i = 0; i < ALIGN_MAX_K; ++i) {
      const int s = tid + NT * i;
      etn[i] = (i < K && s < L && t + 1 < F) ? p.em[(long)(t + 1) * V + tok[i]] : 0.f;
    }
    float* trow = p.trellis ? p.trellis + (long)(t + 1) * TW : nullptr;
    uint8_t* brow = p.bp + (long)t * L;
#pragma unroll
    for (int i = 0; i < ALIGN_MAX_K; ++i) {
      const int s = tid + NT * i;
      if (i < K && s < L) {
        const float a = cur[s + 1], b = cur[s];
        const float stay_blank = a + eb, stay_tok = a + et[i], change = b + et[i];
        const float stayed = fmaxf(stay_blank, stay_tok);
        const float v = fmaxf(stay_blank, fmaxf(stay_tok, change));
        nxt[s + 1] = v;
        if (trow) trow[s + 1] = v;
        if (s + 1 == L) p.lastcol[t + 1] = v;
        brow[s] = (uint8_t)((change > stayed ? 1 : 0) | (change < stayed ? 2 : 0));
      }
    }
    if (tid == 0) {
      float c;
      if (p.col0) {
        c = p.col0[t + 1];
      } else {
        run += (double)eb;
        c = (float)run;
      }
      if (t + 1 >= inf_from) c = INFINITY;
      nxt[0] = c;
      if (trow) trow[0] = c;
    }
#pragma unroll
    for (int i = 0; i < ALIGN_MAX_K; ++i) et[i] = etn[i];
    __syncthreads();
  }
  // ---- t_start = argmax(trellis[:, L]) (first maximum), align_transcriptions.py:88
  float best = -INFINITY;
  int best_i = 0x7fffffff;
  for (int r = tid; r <= F; r += NT) {
    const float v = p.lastcol[r];
    if (v > best || (v == best && r < best_i)) {
      best = v;
      best_i = r;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(best_i, o, 64);
    if (ov > best || (ov == best && oi < best_i)) {
      best = ov;
      best_i = oi;
    }
  }
  if ((tid & 63) == 0) {
    red_v[tid >> 6] = best;
    red_i[tid >> 6] = best_i;
  }
  __syncthreads();
  // ---- backtrack (align_transcriptions.py:90-123), one point per frame from t_start down until token 0 is entered.
  // The walk itself is serial (the column at frame t-1 depends on the decision at frame t) but reads only the 2-bit
  // codes, so wave 0 fetches them 64 frames ahead -- lane d takes the <= 64 columns row t-1-d can still reach -- into
  // LDS (the trellis rows are no longer needed) and lane 0 walks the band at LDS latency: ~80 ns per frame instead of
  // ~3 us of dependent global loads.  The log-probabilities of the points are filled in afterwards by all threads.
  __shared__ int walk[4];  // count | first | ok | t_start
  uint8_t* band = reinterpret_cast<uint8_t*>(rows);  // [64][64]
  uint8_t* pcode = p.bp + (long)F * L;               // [F] code of the point at each time index
  if (tid < 64) {
    for (int w = 1; w < (NT + 63) / 64; ++w)
      if (red_v[w] > best || (red_v[w] == best && red_i[w] < best_i)) {
        best = red_v[w];
        best_i = red_i[w];
      }
    if (best_i == 0x7fffffff) best_i = 0;  // every entry is -inf (or NaN): torch.argmax returns 0
    int t = __shfl(best_i, 0, 64), j = L, count = 0, first = 0, ok = 0;
    while (t >= 1 && !ok) {
      const int slo = max(0, j - 64);  // first column index s = j' - 1 the next 64 steps can touch
      const int width = j - slo;       // columns slo .. j - 1
      const int row = t - 1 - tid;     // lane d = tid: code row of step t - d
      {
        uint8_t v[64];  // all 64 byte loads in flight before the first one is consumed
        const uint8_t* src = p.bp + (long)max(row, 0) * L + slo;
#pragma unroll
        for (int c = 0; c < 64; ++c) v[c] = (row >= 0 && c < width) ? src[c] : (uint8_t)0;
#pragma unroll
        for (int c = 0; c < 64; c += 4)
          *reinterpret_cast<uint32_t*>(band + tid * 64 + c) = (uint32_t)v[c] | ((uint32_t)v[c + 1] << 8) | ((uint32_t)v[c + 2] << 16) | ((uint32_t)v[c + 3] << 24);
      }
      __builtin_amdgcn_s_waitcnt(0);   // the band is in LDS (single wave: no barrier needed)
      __builtin_amdgcn_wave_barrier();
      int steps = 0;
      if (tid == 0) {
        for (int d = 0; d < 64 && t - d >= 1; ++d) {
          const int tt = t - d;
          const int code = band[d * 64 + (j - 1 - slo)];
          p.path_token[tt - 1] = j - 1;
          pcode[tt - 1] = (uint8_t)code;
          ++count;
          first = tt - 1;
          ++steps;
          if (code & 1) {
            --j;
            if (j == 0) {
              ok = 1;
              break;
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      t -= __shfl(steps, 0, 64);
      j = __shfl(j, 0, 64);
      ok = __shfl(ok, 0, 64);
    }
    if (tid == 0) {
      walk[0] = count;
      walk[1] = first;
      walk[2] = ok;
      p.path_info[0] = ok ? count : -1;
      p.path_info[1] = first;
    }
  }
  __syncthreads();
  const int count = walk[0], first = walk[1];
  for (int i = tid; i < count; i += NT) {
    const int time = first + i, t = time + 1;
    const int tk = p.tokens[p.path_token[time]];
    const int code = pcode[time];
    const bool gt = code & 1, lt = code & 2;
    float lp;
    if (lt && t < F)
      lp = fmaxf(p.em[(long)(t - 1) * V], p.em[(long)t * V + tk]);  // emission[t-1, 0] -- index 0, as the reference writes it
    else
      lp = p.em[(long)(t - 1) * V + (gt ? tk : 0)];
    p.path_logp[time] = lp;
  }
}

}  // namespace

static size_t align_ws_per_utt(int F, int L) {
  // a code per trellis cell + the codes along the path, then the trellis' last column (F + 1 floats, 4-byte aligned)
  return (((size_t)F * (size_t)L + (size_t)F + 3) & ~(size_t)3) + (size_t)(F + 1) * sizeof(float);
}

extern "C" size_t ssak_ctc_align_workspace_bytes(int F, int L) {
  if (F <= 0 || L <= 0) return 0;
  return align_ws_per_utt(F, L);
}

extern "C" size_t ssak_ctc_align_batch_workspace_bytes(int B, int Fmax, int Lmax) {
  if (B <= 0 || Fmax <= 0 || Lmax <= 0) return 0;
  return (size_t)B * align_ws_per_utt(Fmax, Lmax);
}

static int align_launch(const AlignBatch& q, int B, hipStream_t st) {
  const int L = q.Lmax;
  const int nt = std::min(1024, std::max(64, (L + 63) / 64 * 64));
  const size_t lds = std::max<size_t>(2 * (size_t)(L + 1) * sizeof(float), 64 * 64);  // trellis rows, later the code band
  static bool attr_done = false;
  if (!attr_done) {
    SSAK_HIP(hipFuncSetAttribute((const void*)align_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (ALIGN_MAX_K * 1024 + 1) * 4));
    attr_done = true;
  }
  align_kernel<<<B, nt, lds, st>>>(q);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}

extern "C" int ssak_ctc_forced_align(const float* emission, const int32_t* tokens, int F, int V, int L, int blank, const float* col0,
                                     float* trellis, int32_t* path_token, float* path_logp, int32_t* path_info, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(emission && tokens && trellis && path_token && path_logp && path_info, "forced_align: null pointer");
  SSAK_REQUIRE(F >= 1 && V >= 1 && L >= 1, "forced_align: F=%d V=%d L=%d must be >= 1", F, V, L);
  SSAK_REQUIRE(L <= ALIGN_MAX_K * 1024, "forced_align: transcript of %d tokens exceeds %d", L, ALIGN_MAX_K * 1024);
  SSAK_REQUIRE(blank >= 0 && blank < V, "forced_align: blank id %d outside the vocabulary of %d", blank, V);
  SSAK_REQUIRE(workspace && workspace_bytes >= ssak_ctc_align_workspace_bytes(F, L), "forced_align: workspace too small");
  const size_t codes = ((size_t)F * L + F + 3) & ~(size_t)3;
  AlignBatch q{emission, nullptr, tokens, nullptr, col0, trellis, (float*)((char*)workspace + codes), (uint8_t*)workspace,
               path_token, path_logp, path_info, F, V, L, blank};
  return align_launch(q, 1, (hipStream_t)stream);
}

extern "C" int ssak_ctc_forced_align_batch(const float* emission, const int32_t* frame_lens, const int32_t* tokens,
                                           const int32_t* token_lens, int B, int Fmax, int V, int Lmax, int blank, const float* col0,
                                           float* trellis, int32_t* path_token, float* path_logp, int32_t* path_info,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(emission && tokens && path_token && path_logp && path_info, "forced_align_batch: null pointer");
  SSAK_REQUIRE(B >= 1 && Fmax >= 1 && V >= 1 && Lmax >= 1, "forced_align_batch: B=%d Fmax=%d V=%d Lmax=%d must be >= 1", B, Fmax, V, Lmax);
  SSAK_REQUIRE(Lmax <= ALIGN_MAX_K * 1024, "forced_align_batch: transcripts of up to %d tokens exceed %d", Lmax, ALIGN_MAX_K * 1024);
  SSAK_REQUIRE(blank >= 0 && blank < V, "forced_align_batch: blank id %d outside the vocabulary of %d", blank, V);
  SSAK_REQUIRE(workspace && workspace_bytes >= ssak_ctc_align_batch_workspace_bytes(B, Fmax, Lmax), "forced_align_batch: workspace too small");
  // workspace: B code blocks of (Fmax * Lmax + Fmax) bytes, then B last-column rows of Fmax + 1 floats
  const size_t codes = ((size_t)B * ((size_t)Fmax * Lmax + Fmax) + 3) & ~(size_t)3;
  SSAK_REQUIRE(codes + (size_t)B * (Fmax + 1) * sizeof(float) <= workspace_bytes, "forced_align_batch: workspace too small");
  AlignBatch q{emission, frame_lens, tokens, token_lens, col0, trellis, (float*)((char*)workspace + codes), (uint8_t*)workspace,
               path_token, path_logp, path_info, Fmax, V, Lmax, blank};
  return align_launch(q, B, (hipStream_t)stream);
}
