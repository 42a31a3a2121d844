// Word error counts on the device for the evaluation step.
//
// Stands behind compute_metrics of ssak/train/transformers/wav2vec_train.py:110-125: argmax + processor.batch_decode
// (greedy CTC; ssak_ctc_greedy_decode), label ids decoded without grouping, remove_special_words(glue_apostrophe=False)
// on both strings (ssak/utils/text_basic.py:91-110), then the "wer" metric = sum of word edits / sum of reference words.
// The reference pulls the full logits to the host at every eval step; here only two integers per utterance leave the
// device.  Strings never exist: a token-class table says what each id does to the word segmentation
//   0 letter, 1 word separator ("|" = space), 2 removed (pad / <s> / </s> / <unk>: "<...>" words are deleted from the
//   text, which glues their neighbours), 3 letter that also ends the word (the apostrophe under glue_apostrophe=False),
// words are compared as id sequences, and the Levenshtein recursion over words is exact integer arithmetic.
// One thread per utterance (tens of words each; the evaluation batch is the parallel dimension).
#include "common.h"
#include "kernels.h"

namespace {

struct WerParams {
  const int32_t* hyp;       // [B, F] collapsed ids from the greedy decode
  const int32_t* hyp_lens;  // [B]
  const int32_t* labels;    // [B, Lmax], negative = padding
  const uint8_t* cls;       // [V]
  int32_t* edits;           // [B]
  int32_t* ref_words;       // [B]
  int32_t* ws;              // per utterance: hyp word starts | lens [F], ref word starts | lens [Lmax], DP row [F + 1]
  int B, F, Lmax, V;
};

// segment seq[0, n) into words (start, len) over the COMPACTED letter sequence written to `letters`
__device__ int segment(const int32_t* seq, int n, const uint8_t* cls, int V, int32_t* letters, int32_t* starts, int32_t* lens) {
  int nw = 0, nl = 0, cur = 0;  // cur = letters in the open word
  for (int i = 0; i < n; ++i) {
    const int id = seq[i];
    if (id < 0 || id >= V) continue;  // padding
    const int c = cls[id];
    if (c == 2) continue;
    if (c == 1) {
      if (cur > 0) {
        starts[nw] = nl - cur;
        lens[nw++] = cur;
        cur = 0;
      }
      continue;
    }
    letters[nl++] = id;
    ++cur;
    if (c == 3) {
      starts[nw] = nl - cur;
      lens[nw++] = cur;
      cur = 0;
    }
  }
  if (cur > 0) {
    starts[nw] = nl - cur;
    lens[nw++] = cur;
  }
  return nw;
}

__global__ void wer_kernel(const WerParams p) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  const long per = 3L * p.F + 3L * p.Lmax + (p.F + 1);
  int32_t* w = p.ws + (long)b * per;
  int32_t *hl = w, *hs = hl + p.F, *hn = hs + p.F;            // hyp letters | word starts | word lens
  int32_t *rl = hn + p.F, *rs = rl + p.Lmax, *rn = rs + p.Lmax;  // ref
  int32_t* row = rn + p.Lmax;                                  // DP row over hyp words
  const int nh = segment(p.hyp + (long)b * p.F, min(max(p.hyp_lens[b], 0), p.F), p.cls, p.V, hl, hs, hn);
  const int nr = segment(p.labels + (long)b * p.Lmax, p.Lmax, p.cls, p.V, rl, rs, rn);
  for (int j = 0; j <= nh; ++j) row[j] = j;
  for (int i = 1; i <= nr; ++i) {
    int diag = row[0];  // D[i-1][0]
    row[0] = i;
    const int32_t* rw = rl + rs[i - 1];
    const int rlen = rn[i - 1];
    for (int j = 1; j <= nh; ++j) {
      bool same = hn[j - 1] == rlen;
      if (same) {
        const int32_t* hw = hl + hs[j - 1];
        for (int k = 0; k < rlen; ++k)
          if (hw[k] != rw[k]) {
            same = false;
            break;
          }
      }
      const int up = row[j];  // D[i-1][j]
      const int v = min(min(up + 1, row[j - 1] + 1), diag + (same ? 0 : 1));
      diag = up;
      row[j] = v;
    }
  }
  p.edits[b] = row[nh];
  p.ref_words[b] = nr;
}

}  // namespace

extern "C" size_t ssak_ctc_wer_workspace_bytes(int B, int F, int Lmax) {
  if (B <= 0 || F <= 0 || Lmax <= 0) return 0;
  return (size_t)B * (3 * (size_t)F + 3 * (size_t)Lmax + (size_t)F + 1) * sizeof(int32_t);
}

extern "C" int ssak_ctc_wer(const int32_t* hyp_ids, const int32_t* hyp_lens, const int32_t* labels, const uint8_t* token_class, int B,
                            int F, int Lmax, int V, int32_t* edits, int32_t* ref_words, void* workspace, size_t workspace_bytes,
                            void* stream) {
  SSAK_REQUIRE(hyp_ids && hyp_lens && labels && token_class && edits && ref_words, "wer: null pointer");
  SSAK_REQUIRE(B > 0 && F > 0 && Lmax > 0 && V > 0, "wer: bad shape");
  SSAK_REQUIRE(workspace && workspace_bytes >= ssak_ctc_wer_workspace_bytes(B, F, Lmax), "wer: workspace too small");
  WerParams p{hyp_ids, hyp_lens, labels, token_class, edits, ref_words, (int32_t*)workspace, B, F, Lmax, V};
  wer_kernel<<<ssak_cdiv(B, 64), 64, 0, (hipStream_t)stream>>>(p);
  SSAK_LAUNCH_CHECK();
  return SSAK_OK;
}
