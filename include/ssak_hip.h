/* ssak_hip.h -- C ABI of libssak_hip.so: the MI355X (gfx950) acoustic-model hot path for SSAK.
 *
 * linto-ai/ssak has no FFI or plugin interface of its own: its hot path is three lines of Python
 * glue around third-party objects.  The entry points below are what a binding for
 * that path would call; each one names the reference call site it stands behind.  All pointers are
 * DEVICE pointers unless marked "host"; every entry point is asynchronous on `stream`
 * (a hipStream_t passed as void*), takes caller-owned buffers, keeps no global state and returns
 * SSAK_OK or a negative status (ssak_last_error() gives the message).  No torch types anywhere.
 *
 * Dtypes: bf16 = 16-bit brain float (uint16_t storage), activations row-major [rows, channels].
 */
#ifndef SSAK_HIP_H
#define SSAK_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSAK_OK 0
#define SSAK_ERR_INVALID (-1) /* bad argument (shape, alignment, null pointer, workspace too small) */
#define SSAK_ERR_LAUNCH (-2)  /* HIP runtime / kernel launch failure */
#define SSAK_ERR_STATE (-3)   /* engine used out of order (e.g. backward before forward) */

#define SSAK_REDUCTION_SUM 0
#define SSAK_REDUCTION_MEAN 1

int ssak_version(void);
const char* ssak_last_error(void); /* host string, valid until the next failing call on this thread */

/* ---- a1: waveform normalisation -------------------------------------------------------------
 * Replaces Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm + right zero padding, reached from
 * ssak/utils/dataset.py:632 (training) and ssak/infer/transformers_infer.py:216 (inference).
 * in  [B, T] fp32 raw samples (anything beyond lens[b] is ignored), lens [B] int32 (NULL = all T)
 * out [B, T] fp32: (x - mean) / sqrt(var + 1e-7) over the first lens[b] samples, 0 beyond;
 * mask [B, T] int32 attention mask (1 valid / 0 pad) or NULL.  workspace >= ssak_wave_normalize_workspace_bytes. */
size_t ssak_wave_normalize_workspace_bytes(int B, int T);
int ssak_wave_normalize(const float* in, const int32_t* lens, int B, int T, float* out, int32_t* mask,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- a9: CTC loss + gradient ----------------------------------------------------------------
 * Replaces log_softmax(fp32) -> F.ctc_loss(blank=pad_token_id, reduction, zero_infinity) and its
 * autograd backward, reached from Wav2Vec2ForCTC.forward (transformers modeling_wav2vec2.py:1705-1728)
 * under ssak/train/transformers/wav2vec_train.py:319,325,415.
 * logits [B, F, V] fp32 (raw, pre-softmax); in_lens [B] int32 valid frames; labels [B, Lmax] int32,
 * negative = padding (-100, wav2vec_train.py:100; target length = count(label >= 0)).
 * loss [1] fp32; nll [B] fp32 per-utterance negative log-likelihood (after zero_infinity) or NULL;
 * dlogits [B, F, V] fp32 = grad_scale * d loss / d logits (0 for frames >= in_lens[b]) or NULL.
 * "mean" = mean over batch of nll_b / max(target_len_b, 1).  Infeasible alignments give nll = +inf,
 * zeroed (loss and gradient) iff zero_infinity. */
size_t ssak_ctc_workspace_bytes(int B, int F, int V, int Lmax);
int ssak_ctc_loss_fwd_bwd(const float* logits, const int32_t* in_lens, const int32_t* labels, int B, int F, int V,
                          int Lmax, int blank, int reduction, int zero_infinity, float grad_scale, float* loss,
                          float* nll, float* dlogits, void* workspace, size_t workspace_bytes, void* stream);

/* ---- a12: greedy CTC decode -----------------------------------------------------------------
 * Replaces torch.argmax(logits, -1) + the collapse step of processor.batch_decode
 * (ssak/infer/transformers_infer.py:84-85): argmax over V, merge repeats, drop blank.
 * ids [B, F] int32 receives the collapsed ids left-aligned, out_lens [B] their counts. */
int ssak_ctc_greedy_decode(const float* logits, const int32_t* in_lens, int B, int F, int V, int blank, int32_t* ids,
                           int32_t* out_lens, void* stream);

/* ---- dense contraction (MFMA bf16, fp32 accumulate) -----------------------------------------
 * The one GEMM behind every Linear / Conv1d / attention product of a3-a10.  C = alpha * op(A) * op(B)
 * (+ epilogue).  Operand layouts: *_kmajor = 0 -> stored [rows, K] with K contiguous (A: [M,K], B: [N,K],
 * i.e. an nn.Linear weight); 1 -> stored [K, rows] (A: [K,M], B: [K,N]).  ld* are element strides of the
 * stored matrix (multiples of 8; ldA < K is allowed and gives the overlapping-row "Toeplitz" operand that
 * makes a channels-last Conv1d a plain GEMM).  Two batch levels: z = z1 * nb2 + z2 with element
 * strides s?1 / s?2 per operand.  Epilogue: bias (fp32 [N] or NULL), then `epilogue` selects
 * NONE / GELU (optionally saving the pre-activation to `aux_out`) / MUL_GELU_GRAD (C *= gelu'(aux_in)).
 * out_f32 selects fp32 vs bf16 C.  split_k > 1 needs workspace >= split_k*batch*M*N*4 bytes and is summed
 * deterministically by a second kernel. */
#define SSAK_EPI_NONE 0
#define SSAK_EPI_GELU 1
#define SSAK_EPI_MUL_GELU_GRAD 2
typedef struct {
  int M, N, K;
  int a_kmajor, b_kmajor;
  long lda, ldb, ldc;
  int nb1, nb2;
  long sa1, sa2, sb1, sb2, sc1, sc2;
  float alpha;
  int epilogue;
  int out_f32;
  int accumulate; /* C += result (fp32 out only) */
  int split_k;
} ssak_gemm_desc;
int ssak_gemm_bf16(const ssak_gemm_desc* desc /*host*/, const void* A, const void* B, void* C, const float* bias,
                   const void* aux_in, void* aux_out, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
