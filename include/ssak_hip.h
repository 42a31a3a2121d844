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

/* ---- a13: Whisper log-mel features ---------------------------------------------------------
 * Replaces WhisperFeatureExtractor (transformers feature_extraction_whisper.py:95-168), reached from
 * ssak/utils/dataset.py:632-637 with a Whisper processor (ssak/train/transformers/whisper_train.py:356-367).
 * wav [B, T] fp32 (lens [B] valid samples or NULL), padded / trimmed to n_samples (480000 = 30 s) -> mel
 * [B, 80, n_samples/160] fp32 (or NULL).  mel_cl_bf16 (or NULL): the same values as bf16 channels-last
 * [B, cl_rows, 80] written from row cl_lead (what the Whisper encoder's first conv reads; pad rows untouched).
 * tables: ssak_logmel_table_floats() floats, 16-byte aligned, filled once by ssak_logmel_init_tables (the FFT's twiddles, the
 * periodic Hann window and the Slaney mel filters, computed in double on the host). */
size_t ssak_logmel_table_floats(void);
int ssak_logmel_init_tables(float* tables /*device*/);
size_t ssak_logmel_workspace_bytes(int B, int n_samples);
int ssak_logmel_whisper(const float* wav, const int32_t* lens, int B, int T, int n_samples, const float* tables, float* mel,
                        void* mel_cl_bf16, int cl_rows, int cl_lead, void* workspace, size_t workspace_bytes, void* stream);

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

/* ---- f2: audio ingest (PCM decode, mono mix, sample-rate conversion) ---------------------------
 * Replaces load_audio / conform_audio of ssak/utils/audio.py:24-154 for PCM input, reached from the Kaldi-folder loader
 * (ssak/utils/dataset.py:27-424,630-645): the segment cut is a byte range chosen by the caller (offset = int(start*sr),
 * audio.py:85-92), channels are averaged (librosa.to_mono, :118) and the rate change is torchaudio.transforms.Resample
 * with its defaults (:134): Hann-windowed sinc interpolation, lowpass_filter_width 6, rolloff 0.99.
 * ssak_pcm_to_mono_f32: raw = interleaved little-endian PCM bytes of all utterances (device), byte_offsets [B] int64 /
 *   nframes [B] int32 (device), sample_width in bytes (1 unsigned, 2 / 4 signed) -> out [B, Tmax] fp32, zero padded.
 * ssak_resample_plan / _table (host): reduced rates, filter half-width and taps; table [new_r][taps] fp32 computed in
 *   double like torchaudio's kernel.  ssak_resample_sinc: in [B, Tin] (in_lens [B] or NULL) -> out [B, Tout] with
 *   out_lens[b] = ceil(new_r * len / orig_r) valid samples (0 beyond); table on the device. */
/* ssak_read_ranges (host): n byte ranges (path, file offset, length) read with pread by `threads` native threads (a process-wide pool that outlives the call) into dst[i] (the
 *   caller's slices of a pinned staging buffer).  SSAK_ERR_INVALID with the path in ssak_last_error() for a file that cannot be
 *   opened or ends inside its range. */
int ssak_read_ranges(const char* const* paths, const int64_t* file_offsets, const int64_t* nbytes, void* const* dst, int n,
                     int threads);
/* ssak_drop_file_cache (host, measurement aid): fdatasync + posix_fadvise(DONTNEED) on each file, so that the next read comes from the
 *   storage device (bench.py's cold-cache ingest figure); returns the number of files it could not open / advise. */
int ssak_drop_file_cache(const char* const* paths, int n);
int ssak_pcm_to_mono_f32(const void* raw, const int64_t* byte_offsets, const int32_t* nframes, int B, int channels, int sample_width,
                         int Tmax, float* out, void* stream);
int ssak_resample_plan(int orig_sr, int new_sr, int* orig_r, int* new_r, int* width, int* taps);
int ssak_resample_table(int orig_sr, int new_sr, float* table /*host*/);
int ssak_resample_sinc(const float* in, const int32_t* in_lens, int B, int Tin, int orig_sr, int new_sr, const float* table /*device*/,
                       float* out, int Tout, int32_t* out_lens, void* stream);

/* ---- f3: word error counts for the evaluation step ---------------------------------------------
 * Replaces compute_metrics of ssak/train/transformers/wav2vec_train.py:110-125 after the argmax (ssak_ctc_greedy_decode):
 * batch_decode of predictions and (ungrouped) labels, remove_special_words(glue_apostrophe=False), the "wer" metric.
 * hyp_ids [B, F] / hyp_lens [B] as ssak_ctc_greedy_decode writes them; labels [B, Lmax] int32, negative = padding;
 * token_class [V] uint8: 0 letter, 1 word separator ("|"), 2 removed from the text (pad and the other "<...>" tokens),
 * 3 letter that also ends its word (the apostrophe).  edits [B] = substitutions + deletions + insertions between the
 * word sequences, ref_words [B] = reference words; WER = sum(edits) / sum(ref_words). */
size_t ssak_ctc_wer_workspace_bytes(int B, int F, int Lmax);
int ssak_ctc_wer(const int32_t* hyp_ids, const int32_t* hyp_lens, const int32_t* labels, const uint8_t* token_class, int B, int F,
                 int Lmax, int V, int32_t* edits, int32_t* ref_words, void* workspace, size_t workspace_bytes, void* stream);

/* ---- f1: CTC forced alignment (Viterbi trellis + backtrack) ----------------------------------
 * Replaces get_trellis + backtrack of ssak/utils/align_transcriptions.py:27-70,79-123 (USE_MAX, USE_CHAR_REPEATED),
 * reached from compute_alignment (:294-402) under tools/align_audio_transcript.py:121,335.
 * emission [F, V] fp32 log-probabilities (compute_log_probas, ssak/infer/general.py:99-101); tokens [L] int32 in [0, V);
 * col0 [F+1] = trellis column 0 supplied by the caller (the first_as_garbage variant, :38) or NULL for the running sum of
 * the blank log-probability (:40).  Outputs: trellis [F+1, L+1] fp32, bit-identical to the reference's; the path as
 * arrays indexed by time frame -- path_token[t] = token index, path_logp[t] = log of Point.score -- valid for
 * t in [path_info[1], path_info[1] + path_info[0]); path_info[0] = -1 when the walk ends without reaching token 0
 * (the reference raises RuntimeError("Failed to align ...")), -2 when a token id lies outside [0, V).
 * workspace >= ssak_ctc_align_workspace_bytes (one byte per trellis cell + one per frame).  One utterance per call, L <= 16384. */
size_t ssak_ctc_align_workspace_bytes(int F, int L);
int ssak_ctc_forced_align(const float* emission, const int32_t* tokens, int F, int V, int L, int blank, const float* col0,
                          float* trellis, int32_t* path_token, float* path_logp, int32_t* path_info /*[2]*/, void* workspace,
                          size_t workspace_bytes, void* stream);
/* The same for B utterances in ONE launch (one workgroup each): what tools/align_audio_transcript.py:121-335
 * (split_long_audio_kaldifolder) does utterance by utterance through compute_alignment.  Padded layouts: emission
 * [B, Fmax, V], frame_lens [B] (NULL = Fmax), tokens [B, Lmax], token_lens [B] (NULL = Lmax), col0 [B, Fmax+1] or NULL,
 * trellis [B, Fmax+1, Lmax+1] or NULL (utterance b fills the top-left (F_b+1) x (L_b+1) corner; callers that only cut audio
 * at word boundaries do not need it), path_token / path_logp [B, Fmax], path_info [B, 2] (as above; -3 = a length outside
 * [1, Fmax] / [1, Lmax]).  Every utterance's results are bit-identical to its single-utterance call. */
size_t ssak_ctc_align_batch_workspace_bytes(int B, int Fmax, int Lmax);
int ssak_ctc_forced_align_batch(const float* emission, const int32_t* frame_lens, const int32_t* tokens, const int32_t* token_lens,
                                int B, int Fmax, int V, int Lmax, int blank, const float* col0, float* trellis, int32_t* path_token,
                                float* path_logp, int32_t* path_info /*[B,2]*/, void* workspace, size_t workspace_bytes, void* stream);

/* ---- dense contraction (MFMA bf16, fp32 accumulate) -----------------------------------------
 * The one GEMM behind every Linear / Conv1d / attention product of a3-a10.  C = alpha * op(A) * op(B)
 * (+ epilogue).  Operand layouts: *_kmajor = 0 -> stored [rows, K] with K contiguous (A: [M,K], B: [N,K],
 * i.e. an nn.Linear weight); 1 -> stored [K, rows] (A: [K,M], B: [K,N]).  ld* are element strides of the
 * stored matrix (multiples of 8; ldA < K is allowed and gives the overlapping-row "Toeplitz" operand that
 * makes a channels-last Conv1d a plain GEMM).  Two batch levels: z = z1 * nb2 + z2 with element
 * strides s?1 / s?2 per operand.  Epilogue: bias (fp32 [N] or NULL), then `epilogue` selects
 * NONE / GELU (optionally saving the pre-activation to `aux_out`) / MUL_GELU_GRAD (C *= gelu'(aux_in)).
 * out_f32 selects fp32 vs bf16 C.  split_k > 1 needs workspace >= split_k*batch*M*N*4 bytes and is summed
 * deterministically by a second kernel; split_k = 0 lets the library size the split (<= 32, and only as far as
 * `workspace_bytes` allows; plain epilogue only, otherwise 1) -- the weight-gradient form, long K and few tiles. */
#define SSAK_EPI_NONE 0
#define SSAK_EPI_GELU 1
#define SSAK_EPI_MUL_GELU_GRAD 2
/* The feed-forward pair of the encoder layers (Wav2Vec2FeedForward, modeling_wav2vec2.py:565-572, and its autograd): the
 * forward GEMM computes y = dropout(gelu(x)) and, instead of the pre-activation x, saves the backward's whole elementwise
 * factor f = gelu'(x) * keep / (1 - p) to `aux_out` (same element offsets as C); the backward GEMM then only multiplies,
 * C *= aux_in -- no exp, no reciprocal and no mask hash in the epilogue that used to be the slowest of the layer.
 * ssak_gemm_bf16 stores the factor as ONE BYTE per element (aux buffers are uint8 [.., ldc]): code 26 = exactly 0 (also a
 * dropped element), otherwise gelu'(x) on the uniform grid (code - 26) * 1.26 / 254 over [-0.129, 1.136] (rounding error
 * <= 0.0025, bf16's own error for |f| >= 0.6); 1 / (1 - p) is applied when the codes are read, so the SSAK_EPI_MUL_AUX product
 * is given the forward's drop_p (it draws no mask).  ssak_gemm_f32 keeps a float factor with mask and scale folded in. */
#define SSAK_EPI_GELU_SAVE_GRAD 3
#define SSAK_EPI_MUL_AUX 4
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
  float drop_p;        /* > 0: dropout fused into the epilogue (after GELU / on the GELU-grad product) */
  uint32_t drop_stream; /* keep(drop_seed, drop_stream, output row z * M + m, output column n): replayable in backward */
  uint64_t drop_seed;
  long bias_s2; /* bias element stride per second-level batch index (grouped conv: one bias slice per group) */
  int pads_are_zero; /* caller guarantees that elements between the logical extent and the next multiple of 8
                        (K for K-contiguous operands, rows for K-major ones) are zero in memory: lets operands whose
                        extent is not a multiple of 8 (e.g. 499 frames) take the direct-to-LDS path */
  int colsum; /* != 0: `aux_out` is a float [N] vector that receives += the column sums of the stored C (after the epilogue:
                 the bias gradient of the Linear that produced the GEMM's A operand side), summed in a fixed order;
                 needs workspace >= ceil(M / 64) * N * 4 bytes; not with SSAK_EPI_GELU, split_k or batches */
  int dynamic_tiles; /* tile order of the persistent kernels for THIS product.  0 (default): static stride over the workgroups;
                 1: every tile is drawn from per-XCD ticket counters, so that workgroups whose CU is held by another stream's
                 kernel for a while -- the RCCL all-reduce of a data-parallel step -- take fewer tiles instead of finishing
                 last.  Alone on the chip the static order is a few percent faster (no ticket round trip at the start of a
                 launch); results are bit-identical either way.  (A grouped launch reads it from descs[0].) */
  const void* b_fragments; /* optional (NULL = none): the FRAGMENT-ORDERED copy of this product's B operand made by
                 ssak_gemm_fragment_b for the same N, K -- for a B that is a weight, static between optimizer steps.  When the
                 library picks the persistent kernel's "B-direct" form for the shape (ssak_gemm_uses_fragments), every wave
                 loads its B fragments from this copy straight into registers and B never enters LDS (the LDS form is bound by
                 LDS read bandwidth; N = 768 products of the train step run 5-25 % faster); otherwise it is ignored and B is
                 read.  Both must describe the same matrix: results are bit-identical either way.  Not with B batch strides. */
  int plan_tile; /* 0 (default): the library's cost model picks kernel, tile height and split.  256 / 192 / 128: run the product
                 on the persistent 256-column-tile kernels with this tile height whenever it qualifies for them (M, N >= 256,
                 whole 16-byte chunks) -- for tests and tuning; results do not depend on it beyond the summation order.  (Round 5's
                 value 129, a co-resident 128-row-tile kernel that never won a shape, is gone: tools/probes/gemm_c4.hip.txt.) */
} ssak_gemm_desc;
/* Fragment-ordered copy of a B operand ([N, K] K-contiguous, or [K, N] with b_kmajor; ldb as in the descriptor):
 * out[(cb * nkt + kt)][j][kk][lane][8] = B(n = 64 cb + 16 j + (lane & 15), k = 64 kt + 32 kk + 8 (lane >> 4) + e), zeros
 * beyond N / K, cb < 4 * ceil(N / 256), nkt = ceil(K / 64): the 8 KB one wave column needs for one 64-deep K tile are
 * contiguous and each `v_mfma_f32_16x16x32_bf16` operand is one 1-KiB wave-level load of whole cache lines. */
size_t ssak_gemm_fragment_b_bytes(int N, int K);
int ssak_gemm_fragment_b(const void* B, long ldb, int N, int K, int b_kmajor, void* out, void* stream);
/* The same for `n` matrices in ONE launch (host arrays of length n): the engine refreshes the copies of every kept layer's
 * weights once per train step. */
int ssak_gemm_fragment_b_batched(int n, const void* const* B, const long* ldb, const int* N, const int* K, const int* b_kmajor,
                                 void* const* out, void* stream);
/* 1 when ssak_gemm_bf16 would take the B-direct form for this descriptor if b_fragments were given (callers that keep the
 * copies fresh only for products that use them), else 0. */
int ssak_gemm_uses_fragments(const ssak_gemm_desc* desc);
int ssak_gemm_bf16(const ssak_gemm_desc* desc /*host*/, const void* A, const void* B, void* C, const float* bias,
                   const void* aux_in, void* aux_out, void* workspace, size_t workspace_bytes, void* stream);
/* The same contraction with float operands, float results and float aux buffers on the fp32 matrix pipe
 * (v_mfma_f32_32x32x2_f32): the GEMM of the fp32-exact verification mode (ssak_w2v2_config.exact).  The reference computes
 * these products in fp32 (USE_MIXED_PRECISION = False, ssak/train/transformers/wav2vec_train.py:191-192).  out_f32, split_k,
 * pads_are_zero are ignored (always float, never split, every access bounds-checked); colsum adds with float atomics. */
int ssak_gemm_f32(const ssak_gemm_desc* desc /*host*/, const void* A, const void* B, void* C, const float* bias,
                  const void* aux_in, void* aux_out, void* stream);

/* Grouped form: n <= 48 plain products (no bias / activation / dropout / batches / split-K) that share K, the operand layouts,
 * alpha and the output type run as ONE persistent launch whose tiles are dealt over the whole chip -- the weight gradients
 * dW = dY^T X of the encoder layers (loss.backward() of wav2vec_train.py:415), which one at a time are too few tiles to fill
 * the GPU without split-K slabs: the engine launches two layers at a time (216 tiles, one round; under data parallelism their
 * gradient ranges become ready early for the all-reduce).  descs / A / B / C are host arrays of length n. */
int ssak_gemm_bf16_grouped(const ssak_gemm_desc* descs /*host*/, int n, const void* const* A, const void* const* B, void* const* C,
                           void* stream);

/* Per-launch timing for the roofline report (measurement aid, not on the reference's path): while enabled, launches are
 * bracketed by HIP events on their own stream; ssak_prof_collect waits for them and returns one entry per slot: the kernel
 * classes of the train step (attention, LayerNorm, conv0, AdamW, CTC, ...) and every GEMM instantiation launched so far
 * (named as rocprofv3 prints it), each with launches / summed ms / ALGORITHMIC work -- flops for SSAK_BOUND_MFMA slots,
 * bytes for SSAK_BOUND_HBM / SSAK_BOUND_LATENCY slots.  Timing is switched on PER STREAM (no process-wide state: launches on
 * other streams -- another handle, another thread -- are not touched and not reported): ssak_prof_enable(stream, 0) = off,
 * (stream, 1) = every launch on that stream, (stream, 2 + i) = only slot i:
 * an event pair keeps consecutive kernels from overlapping head to tail, and bracketing all launches of a train step costs
 * a few per cent of it, so a benchmark surveys all slots in warm-up steps and times only the slot it reports on inside its
 * timed region. */
#define SSAK_BOUND_MFMA 0
#define SSAK_BOUND_HBM 1
#define SSAK_BOUND_LATENCY 2
typedef struct {
  char name[112];
  long launches;
  double total_ms;
  double total_flops; /* flops, or bytes for the HBM / latency-bound slots */
  int bound;
} ssak_prof_entry;
int ssak_prof_enable(void* stream, int on);
/* as ssak_prof_enable(stream, 2 + i) for a LIST of slots (indices into ssak_prof_collect's list): a kernel that serves several
 * products has one slot per (N, K); a benchmark that reports on the kernel times all of them. */
int ssak_prof_enable_slots(void* stream, const int32_t* slots, int n);
int ssak_prof_collect(void* stream, ssak_prof_entry* out /*host*/, int cap); /* cap >= 128; returns the number of entries */

/* ---- a3 (part): first layer of the feature encoder -------------------------------------------
 * Conv1d(1, C, k=10, s=5, no bias) -> GroupNorm(C groups: per channel over time) -> GELU of Wav2Vec2GroupNormConvLayer
 * (transformers modeling_wav2vec2.py:302-323), waveform x [B, T] fp32 -> out [B, T0, C] bf16 channels-last, T0 = (T-10)/5+1.
 * w [C, 10], gamma / beta [C] fp32.  Statistics in fp64 from 65 input moments per utterance; for C = 512 the taps run on the
 * matrix cores with a three-term bf16 split (fp32-grade).  Exported for per-op parity tests; the engine calls the same code. */
size_t ssak_conv0_workspace_bytes(int B, int T, int C);
int ssak_conv0_gn_gelu(const float* x, const float* w, const float* gamma, const float* beta, void* out_bf16, void* workspace,
                       size_t workspace_bytes, int B, int T, int C, void* stream);
/* the same on RAW full-length waveforms, the zero-mean / unit-variance normalisation folded into the GroupNorm statistics:
 * out == ssak_conv0_gn_gelu(ssak_wave_normalize(x)) to fp32 rounding (SSAK_W2V2_OPT_RAW_INPUT uses it) */
int ssak_conv0_gn_gelu_raw(const float* x, const float* w, const float* gamma, const float* beta, void* out_bf16, void* workspace,
                       size_t workspace_bytes, int B, int T, int C, void* stream);

/* ---- a7 (part): fused self-attention, head_dim 64 --------------------------------------------
 * Replaces Wav2Vec2Attention's softmax(QK^T d^-0.5 + key mask) -> dropout -> .V and its autograd
 * (transformers modeling_wav2vec2.py:438-463,500-548) for one layer.  qkv [B*F, 3H] bf16 (q | k | v, heads
 * interleaved in the channel dimension), ctx [B*F, H] bf16, lse [B, nh, F] fp32 (saved for the backward),
 * klens [B] valid keys or NULL.  Backward: dctx [B*F, H] -> dqkv [B*F, 3H]; delta [B, nh, F] fp32 scratch.
 * Exported for per-op parity tests; the engine calls the same kernels. */
int ssak_attention_fwd(const void* qkv, void* ctx, float* lse, const int32_t* klens, int B, int F, int nh, int H, float drop_p,
                       uint64_t seed, uint32_t stream_id, void* stream);
int ssak_attention_bwd(const void* qkv, const void* ctx, const float* lse, const int32_t* klens, const void* dctx, float* delta,
                       void* dqkv, int B, int F, int nh, int H, float drop_p, uint64_t seed, uint32_t stream_id, int mode,
                       void* stream);
/* `mode` of the backward: SSAK_ATTN_BWD_DEFAULT = SSAK_ATTN_BWD_TWO_KERNEL = two kernels (dQ; dK + dV), each recomputing P from the
 * saved log-sum-exp.  (Value 2, a fused single-pass form that lost to this one at the train step's shape, was removed in ABI 400:
 * SSAK_ERR_ARG.) */
#define SSAK_ATTN_BWD_DEFAULT 0
#define SSAK_ATTN_BWD_TWO_KERNEL 1
/* As ssak_attention_bwd, and bias_grad[3H] += the column sums of dqkv -- the gradient of the q|k|v projection bias (the Linear layers
 * of Wav2Vec2Attention, modeling_wav2vec2.py:500-548) -- summed inside the two kernels from the rows they hold (deterministic: one
 * partial row per workgroup in `workspace`, one fixed-order second stage), instead of a pass that reads dqkv again. */
size_t ssak_attention_bwd_bias_workspace_bytes(int B, int F, int H);
int ssak_attention_bwd_bias(const void* qkv, const void* ctx, const float* lse, const int32_t* klens, const void* dctx, float* delta,
                            void* dqkv, float* bias_grad, int B, int F, int nh, int H, float drop_p, uint64_t seed,
                            uint32_t stream_id, void* workspace, size_t workspace_bytes, void* stream);

/* ---- a11: optimizer tail (clip_grad_norm_ -> AdamW), flat fp32 buffers ----------------------
 * Replaces torch.nn.utils.clip_grad_norm_(max 1.0) + torch.optim.AdamW.step as driven by HF Trainer
 * (docker/transformers_modified/trainer.py:1827-1855; ssak/train/transformers/wav2vec_train.py:353-384).
 * ssak_grad_sumsq: out[0] = sum(grads^2), summed in a fixed order (per-workgroup partials in `workspace`, >= 4096 bytes,
 * then one pass): the clip coefficient is reproducible run to run.  ssak_adamw_step: g' = grads * grad_scale * min(1, max_norm /
 * (sqrt(gnorm_sq[0]) * grad_scale + 1e-6)) (no clipping when gnorm_sq is NULL or max_norm <= 0), then the
 * AdamW update with bias correction for 1-based `step`; shadow_bf16 (or NULL) receives the bf16 copy of the
 * new parameters.  The clip coefficient is read on the device: no host synchronisation. */
int ssak_grad_sumsq(const float* grads, long n, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* out[0] += sum g^2: the joint norm over several buffers (the SpeechBrain recipe clips the wav2vec2 and head gradients as one
 * vector, speechbrain core Brain.check_gradients under ssak/train/speechbrain/wav2vec_train.py:113,125) */
int ssak_grad_sumsq_add(const float* grads, long n, float* out, void* workspace, size_t workspace_bytes, void* stream);
int ssak_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, long n,
                    const float* gnorm_sq, float max_norm, float grad_scale, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, void* stream);

/* ---- e: data-parallel gradient exchange ------------------------------------------------------
 * One process per GPU, utterance shards per rank, ONE sum all-reduce of the flat gradient buffer per step (in buckets, as the
 * backward announces them: ssak_w2v2_set_grad_ready_callback) -- what torch.nn.DataParallel's gather + loss.mean() does inside
 * HF Trainer for the reference (docker/transformers_modified/trainer.py:2532-2533; per_device_train_batch_size = batch_size //
 * num_devices, ssak/train/transformers/wav2vec_train.py:349,356).  RCCL over xGMI; librccl.so is dlopen'ed on the first call
 * (libssak_hip.so does not link it: a single-GPU host never loads it).  Rank 0 calls ssak_comm_unique_id and hands the 128
 * bytes to the other ranks by its own means (file, socket, MPI, torch.distributed); every rank then calls ssak_comm_create
 * (collective).  ssak_allreduce: buf[offset, offset + count) of `dtype` elements := the sum over ranks, in place, asynchronous
 * on `stream`; 1 / world is folded into ssak_adamw_step's grad_scale.  One communicator per process and device. */
typedef struct ssak_comm ssak_comm;
#define SSAK_DTYPE_F32 0
#define SSAK_DTYPE_BF16 1
int ssak_comm_unique_id(void* id128 /*host, 128 bytes out*/);
int ssak_comm_create(ssak_comm** out, int world, int rank, const void* id128 /*host*/);
int ssak_allreduce(ssak_comm* comm, void* buf, long offset, long count, int dtype, void* stream);
int ssak_comm_destroy(ssak_comm* comm);

/* ---- a3-a10: the Wav2Vec2-CTC acoustic model ------------------------------------------------
 * Replaces `model(input_values, attention_mask, labels)` / `loss.backward()` of transformers.Wav2Vec2ForCTC as
 * called at ssak/train/transformers/wav2vec_train.py:387-415 (through HF Trainer.training_step) and
 * ssak/infer/transformers_infer.py:235.  The handle owns only derived weight layouts; parameters, gradients,
 * the bf16 shadow and the workspace are caller-owned flat device buffers.
 *
 * Parameter layout: ssak_w2v2_param_info enumerates (HF state_dict name, element offset, shape) of every
 * tensor inside the flat buffers; [0, ssak_w2v2_num_trainable) is what the optimizer / all-reduce touch. */
typedef struct {
  int vocab_size, hidden_size, num_layers, num_heads, intermediate_size;
  int num_conv_layers;
  int conv_dim[8], conv_kernel[8], conv_stride[8];
  int conv_bias;            /* 0 (base) */
  int feat_extract_norm;    /* 0 = "group" (base), 1 = "layer" (XLSR) */
  int do_stable_layer_norm; /* 0 = post-LN (base) */
  int num_conv_pos_embeddings, num_conv_pos_embedding_groups;
  float layer_norm_eps;
  float attention_dropout, hidden_dropout, activation_dropout, feat_proj_dropout, final_dropout;
  int freeze_feature_encoder; /* wav2vec_train.py:326-327 */
  /* arch 1 = Whisper encoder + CTC head (BASELINE config 4; a composition of the build): conv1/conv2
   * front end on log-mel features, fixed sinusoidal positions, pre-LN layers (k_proj without bias), final LayerNorm,
   * Linear(d_model, vocab).  input_values is then the feature tensor [B, num_mel_bins, T] (T = 2 * frames, even). */
  int arch;
  int num_mel_bins, max_source_positions;
  /* exact != 0: the fp32-exact VERIFICATION mode -- activations are stored in float, every product runs through
   * ssak_gemm_f32 on the fp32 master weights, GELU is the exact erf form, attention takes the unfused GEMM + softmax path;
   * the engine's sequencing, layouts, row kernels (the same templates as the bf16 mode) and the CTC kernels are then
   * comparable with the fp32 reference at 1e-4 instead of bf16's 1e-2.  wav2vec2 topologies with the frozen feature
   * encoder; a few per cent of the bf16 mode's speed.  Workspace sizes double. */
  int exact;
} ssak_w2v2_config;
typedef struct ssak_w2v2 ssak_w2v2;

int ssak_w2v2_create(const ssak_w2v2_config* cfg /*host*/, ssak_w2v2** out);
void ssak_w2v2_destroy(ssak_w2v2* h);
long ssak_w2v2_num_params(const ssak_w2v2* h);
long ssak_w2v2_num_trainable(const ssak_w2v2* h);
int ssak_w2v2_param_count(const ssak_w2v2* h);
int ssak_w2v2_param_info(const ssak_w2v2* h, int index, char* name /*host*/, int name_cap, long* offset, long* numel,
                         int* ndim, long* shape4 /*host [4]*/);
/* params/grads fp32 [num_params] (grads may be NULL for inference), shadow_bf16 uint16 [num_params] */
int ssak_w2v2_bind(ssak_w2v2* h, float* params, float* grads, void* shadow_bf16);
/* full != 0: rebuild the whole bf16 shadow + conv layouts from `params` (after loading weights);
 * full == 0: only the weight-normed positional-conv layouts (after an optimizer step that updated the shadow). */
int ssak_w2v2_sync_weights(ssak_w2v2* h, int full, void* stream);
int ssak_w2v2_num_frames(const ssak_w2v2* h, int T); /* floor((L-k)/s)+1 chain, modeling_wav2vec2.py:997-1016 */
size_t ssak_w2v2_workspace_bytes(const ssak_w2v2* h, int B, int T, int training);
/* input_values [B,T] fp32 normalised waveforms; lens [B] int32 valid samples or NULL (= no attention mask, the
 * group-norm/base convention); spec_mask [B,F] uint8 SpecAugment mask or NULL; layer_keep host uint8 [num_layers]
 * LayerDrop decisions or NULL; seed drives every dropout mask of this step; logits [B,F,V] fp32 out;
 * frame_lens [B] int32 out (or NULL).  training != 0 keeps the activations for ssak_w2v2_backward. */
int ssak_w2v2_forward(ssak_w2v2* h, const float* input_values, const int32_t* lens, int B, int T,
                      const uint8_t* spec_mask, const uint8_t* layer_keep /*host*/, uint64_t seed, int training,
                      float* logits, int32_t* frame_lens, void* workspace, size_t workspace_bytes, void* stream);
/* Optional: called (on the host, from inside ssak_w2v2_backward) each time all kernels writing a contiguous range
 * grads[offset, offset+count) have been enqueued on the stream -- once per encoder layer (top to bottom), then for
 * the rest.  The ranges are disjoint and cover [0, num_trainable).  A data-parallel caller launches one bucketed
 * all-reduce per announcement on a side stream so that the exchange overlaps the remaining backward
 * (the reference's nn.DataParallel reduces after backward, docker/transformers_modified/trainer.py:1345-1346). */
typedef void (*ssak_grad_ready_fn)(long offset, long count, void* user);
int ssak_w2v2_set_grad_ready_callback(ssak_w2v2* h, ssak_grad_ready_fn fn, void* user);
/* Optimizer on a side stream (north_star: "all-reduce ... overlapped with the optimizer on a side HIP stream"; the reference's
 * HF Trainer runs optimizer.step() in line, docker/transformers_modified/trainer.py:1827-1855): `params_ready` is a hipEvent_t the
 * caller records on its optimizer stream after the update of params / shadow (and ssak_w2v2_sync_weights(full = 0)) has been
 * enqueued there.  Every forward then waits for it on ITS stream at the first kernel that reads a trainable parameter -- with
 * a frozen feature encoder that is the feature projection, so the whole conv stack of step n+1 runs under the exchange tail and
 * the AdamW sweep of step n.  stall_begin / stall_end (hipEvent_t or NULL) are recorded around that wait: their distance is the
 * exposed part of the tail.  NULL params_ready removes the wait. */
int ssak_w2v2_set_param_event(ssak_w2v2* h, void* params_ready, void* stall_begin, void* stall_end);
/* Per-handle execution options (nothing here changes results beyond rounding; no process-wide switches):
 * SSAK_W2V2_OPT_DYNAMIC_TILES  0 / 1: ssak_gemm_desc.dynamic_tiles of every product the engine launches -- the data-parallel
 *                              trainers set it, RCCL's kernels share the chip with the persistent GEMMs;
 * SSAK_W2V2_OPT_ATTENTION_BWD  SSAK_ATTN_BWD_* (one form since ABI 400; kept so that callers need not change);
 * SSAK_W2V2_OPT_POSCONV_DIRECT  1 (default) / 0: the grouped positional convolution (forward, input gradient and weight gradient)
 *                              as direct convolutions with the input rows resident in LDS (group widths 48 and 64) or as the
 *                              Toeplitz GEMMs of rounds 1-2 (kept for other geometries and as the comparison path of the tests);
 * SSAK_W2V2_OPT_FRAGMENT_WEIGHTS  0 (default) / 1: every TRAINING forward makes fragment-ordered copies of the kept encoder
 *                              layers' projection weights (one batched launch after the optimizer's event, ssak_gemm_fragment_b)
 *                              and the forward / input-gradient products the library would run in its B-direct form take
 *                              them through ssak_gemm_desc.b_fragments; evaluation forwards never do (the copies would go
 *                              stale when the caller rewrites the shadow).  Bit-identical results; engine-owned memory (~2 x
 *                              the bf16 size of the layers' matrices).  OFF by default: with the operands warm the B-direct
 *                              form is 5-25 % faster on the N = 768 products, but in the train step a layer's weights are read
 *                              once per step -- from HBM -- and the gain is gone (3 591 -> 3 664 us per step for the six
 *                              products, + 160 us for the copies: profiles/r03_ab_fragments.log);
 * SSAK_W2V2_OPT_TRANSPOSED_WEIGHTS  1 (default) / 0: every TRAINING forward makes TRANSPOSED bf16 copies of the kept encoder layers'
 *                              four projection matrices (one batched launch after the optimizer's event, ~310 MB of traffic for
 *                              wav2vec2-base) and the backward's input-gradient products dX = dY W read them as a K-contiguous
 *                              B operand -- the layout of the four-wave GEMM (gemm_p4.hip) -- instead of the weight itself
 *                              K-major.  Same products, summation order aside; engine-owned memory (the bf16 size of the
 *                              layers' matrices).  Used when hidden and intermediate sizes are multiples of 256. */
#define SSAK_W2V2_OPT_DYNAMIC_TILES 1
#define SSAK_W2V2_OPT_ATTENTION_BWD 2
#define SSAK_W2V2_OPT_POSCONV_DIRECT 3
#define SSAK_W2V2_OPT_FRAGMENT_WEIGHTS 4
#define SSAK_W2V2_OPT_TRANSPOSED_WEIGHTS 5
/* SSAK_W2V2_OPT_RAW_INPUT  0 (default) / 1: `input_values` of ssak_w2v2_forward are RAW full-length waveforms (no padding inside the
 * batch): the feature extractor's zero-mean / unit-variance normalisation (a1, ssak/utils/dataset.py:632) is folded into the
 * GroupNorm statistics of the first conv layer -- conv0 is linear and bias-free, so only GroupNorm's epsilon changes (1e-5 sigma^2) --
 * and the train step needs no ssak_wave_normalize pass.  wav2vec2 group-norm feature encoder, frozen, only; logits equal the
 * two-pass form's to fp32 rounding.  Ragged batches and inference keep ssak_wave_normalize. */
#define SSAK_W2V2_OPT_RAW_INPUT 6
int ssak_w2v2_set_option(ssak_w2v2* h, int option, int value);
/* The gradient ranges ssak_w2v2_backward announces, in announcement order, from the configuration alone (host arithmetic, no
 * device): head matrix, one range per encoder layer from the last to the first (a layer's q|k|v|out|ffn matrices are
 * contiguous), the leading small matrices, then the vector region (biases, LayerNorm affines) [+ the feature encoder when it
 * is trained].  They are disjoint, start at multiples of 8 elements and cover [0, num_trainable): the buckets of the
 * data-parallel exchange.  Returns the number of ranges (<= cap written), negative on a bad configuration. */
int ssak_w2v2_grad_ranges(const ssak_w2v2_config* cfg /*host*/, long* offsets /*host*/, long* counts /*host*/, int cap);
/* dlogits [B,F,V] fp32 (e.g. from ssak_ctc_loss_fwd_bwd); overwrites grads[0, num_trainable). */
int ssak_w2v2_backward(ssak_w2v2* h, const float* dlogits, void* workspace, size_t workspace_bytes, void* stream);
/* The same model stopped at the encoder's last hidden state -- `self.modules.wav2vec2(wavs)` of the SpeechBrain recipe
 * (ssak/train/speechbrain/wav2vec_train.py:51; HuggingFaceWav2Vec2 returns Wav2Vec2Model(wav)[0]): hidden [B,F,H] bf16 out, no
 * final dropout, no lm_head.  ssak_w2v2_backward_hidden continues from d loss / d hidden [B,F,H] bf16 (the unfrozen case,
 * :95-137); the lm_head gradient stays zero.  A backward must match the kind of forward that preceded it. */
int ssak_w2v2_forward_hidden(ssak_w2v2* h, const float* input_values, const int32_t* lens, int B, int T,
                             const uint8_t* spec_mask, const uint8_t* layer_keep /*host*/, uint64_t seed, int training,
                             void* hidden_bf16, int32_t* frame_lens, void* workspace, size_t workspace_bytes, void* stream);
int ssak_w2v2_backward_hidden(ssak_w2v2* h, const void* dhidden_bf16, void* workspace, size_t workspace_bytes, void* stream);

/* ---- f4: the SpeechBrain recipe's acoustic head ------------------------------------------------
 * Row-wise pieces of `x = enc(feats); logits = ctc_lin(x)` (ssak/train/speechbrain/wav2vec_train.py:51-54) with the modules
 * of ssak/train/speechbrain/fr/hyperparameters_wav2vec_finetune_cv-fr.yaml:87-137; the Linears are ssak_gemm_bf16, log-softmax
 * + ctc_cost is ssak_ctc_loss_fwd_bwd.  Host-side composition: ssak_amd/sb_head.py.
 *
 * ssak_utt_norm_*: F.layer_norm(x, x.shape[1:]) without affine parameters -- the wrapper's waveform normalisation (fp32,
 * is_bf16 = 0) and its output_norm over (frames x features) (bf16, is_bf16 = 1).  x, y [B, n]; stats [B][2] = (mean, rstd) out
 * (may be NULL in the forward when no backward follows); the backward takes y, the forward's OUTPUT.  Any n (rows whose length
 * is a multiple of 8 bf16 / 4 fp32 elements take the 16-byte vector path). */
size_t ssak_utt_norm_workspace_bytes(int B);
int ssak_utt_norm_fwd(const void* x, void* y, int B, long n, int is_bf16, float eps, float* stats, void* workspace,
                      size_t workspace_bytes, void* stream);
int ssak_utt_norm_bwd(const void* dy, const void* y, void* dx, int B, long n, int is_bf16, const float* stats, void* workspace,
                      size_t workspace_bytes, void* stream);
/* BatchNorm1d over all M = B*T rows of x [M, C] bf16 (speechbrain BatchNorm1d on [B,T,C]: statistics over batch and time,
 * padding frames included), then LeakyReLU(leaky_slope) and dropout(drop_p), one fused apply pass: y [M, C] bf16.
 * training != 0: batch statistics (biased variance), running_mean / running_var (or NULL) updated with `momentum` and the
 * unbiased variance; training == 0: the running statistics, no dropout.  save_mean / save_rstd [C] out are what the backward
 * needs together with x.  Backward: dy -> dx, dgamma / dbeta [C] overwritten; the dropout mask is recomputed from
 * (seed, drop_stream, element index).  Sums are two-stage and fixed-order (deterministic).  C a multiple of 8. */
size_t ssak_batchnorm_workspace_bytes(int C);
int ssak_batchnorm_act_fwd(const void* x, void* y, int M, int C, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float momentum, float eps, int training, float leaky_slope, float drop_p,
                           uint64_t seed, uint32_t drop_stream, float* save_mean, float* save_rstd, const double* global_sums,
                           void* workspace, size_t workspace_bytes, void* stream);
int ssak_batchnorm_act_bwd(const void* dy, const void* x, void* dx, int M, int C, const float* gamma, const float* beta,
                           const float* save_mean, const float* save_rstd, float leaky_slope, float drop_p, uint64_t seed,
                           uint32_t drop_stream, float* dgamma, float* dbeta, double* local_sums_out, const double* global_sums,
                           void* workspace, size_t workspace_bytes, void* stream);
/* Synchronised statistics under data parallelism (the head's BatchNorm1d spans the global batch): the column
 * totals travel as doubles through one all-reduce of 2 C + 1 values per normalisation and direction (the last one is the
 * row count, so ranks may hold different numbers of rows and no host read is needed).
 *   forward : ssak_batchnorm_stats(x) -> sums[2C+1] = (sum x [C], sum x^2 [C], M) of the local rows; all-reduce(sum);
 *             ssak_batchnorm_act_fwd(..., global_sums = sums)
 *   backward: ssak_batchnorm_act_bwd(dx = NULL, local_sums_out = sums) -> local dgamma / dbeta (the parameter all-reduce sums
 *             them as usual) and sums[2C+1] = (sum g, sum g xhat, M); all-reduce(sum);
 *             ssak_batchnorm_act_bwd(dx, global_sums = sums) -> dx; dgamma / dbeta untouched (may be NULL).
 * With global_sums == NULL and local_sums_out == NULL both calls are the single-device form. */
int ssak_batchnorm_stats(const void* x, int M, int C, double* sums, void* workspace, size_t workspace_bytes, void* stream);
/* torch.optim.Adadelta (yaml :119-122: lr 1.0, rho 0.95, eps 1e-8): square_avg = rho sq + (1-rho) g^2;
 * delta = sqrt(acc_delta + eps) / sqrt(square_avg + eps) * g; acc_delta = rho acc + (1-rho) delta^2; p -= lr * delta.
 * g is first multiplied by grad_scale (1 / world size after a sum all-reduce) and by min(1, max_norm / (sqrt(*gnorm_sq) *
 * grad_scale + 1e-6)) when gnorm_sq != NULL and max_norm > 0 -- the convention of ssak_adamw_step. */
int ssak_adadelta_step(float* params, const float* grads, float* square_avg, float* acc_delta, void* shadow_bf16, long n,
                       const float* gnorm_sq, float max_norm, float grad_scale, float lr, float rho, float eps, float weight_decay,
                       void* stream);
int ssak_cast_f32_bf16(const float* src, void* dst_bf16, long n, void* stream);
/* the way back (the optional bf16 gradient exchange of the data-parallel trainer: 180 MB instead of 361 MB per step,
 * SURVEY.md 8e); 16-byte aligned buffers */
int ssak_cast_bf16_f32(const void* src_bf16, float* dst, long n, void* stream);
/* out[N] = column sums of X [M, N] bf16 (row stride ld): the bias gradient of a Linear from its output gradient.  Two-stage,
 * fixed-order; N and ld multiples of 8. */
size_t ssak_colsum_workspace_bytes(int N);
int ssak_colsum_bf16(const void* X, long ld, int M, int N, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- TEST-ONLY entries (not part of the product path; kept in the release library so that the parity tests run against the
 * library that ships): dropout bits of one site ------------------------------------------------------------------------
 * The engine stores no dropout mask: each site recomputes keep(seed, site, element) in its forward and backward kernels
 * (transformers draws torch's global generator at modeling_wav2vec2.py:433,458,568,571,596,692,1698; the reference passes the
 * probabilities at ssak/train/transformers/wav2vec_train.py:313-318).  These two entries write the bits out so that the CPU
 * restatement oracle/dropout_hash.py -- which feeds the SAME masks to transformers.Wav2Vec2ForCTC for the regularisers-on
 * goldens -- is pinned bit for bit against the device functions.  Since ABI 500 one definition serves every site:
 * keep(seed, site, row, col) = rowkey(seed, site, row) * colmul(col) mod 2^32 >= round(p * 65536) << 16, with (row, col) of the
 * site's row-major [rows, cols] tensor (cols <= 16 384) and, for attention, row = (b * nh + h) * F + q, col = key.  keep
 * [rows, cols] uint8; attention: keep [B, nh, F, F] (query-major).  site = the engine's stream id of the dropout site;
 * *scale_out (host, may be NULL) = the factor kept elements are multiplied by.  Never called on the hot path. */
int ssak_debug_dropout_mask(uint64_t seed, uint32_t site, float p, long rows, int cols, uint8_t* keep, float* scale_out /*host*/,
                            void* stream);
int ssak_debug_attention_dropout_mask(uint64_t seed, uint32_t site, float p, int B, int nh, int F, uint8_t* keep, void* stream);

#ifdef __cplusplus
}
#endif
#endif
