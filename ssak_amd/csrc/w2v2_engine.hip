// Wav2Vec2-CTC acoustic-model engine: forward, CTC loss + gradient, backward.  Host-side sequencing of
// the gfx950 kernels; everything runs asynchronously on one HIP stream and is hipGraph-capturable (no
// allocation, no synchronisation, no host reads inside forward/backward).
//
// Stands behind `model(input_values, attention_mask, labels)` + `loss.backward()` as the reference drives them
// through HF Trainer (ssak/train/transformers/wav2vec_train.py:387-415; module graph: transformers
// modeling_wav2vec2.py, Wav2Vec2ForCTC.forward :1667-1742) and `model(input_values).logits` for inference
// (ssak/infer/transformers_infer.py:235).
//
// MI355X-first choices (vs the reference's eager PyTorch graph):
//   * activations are bf16, channels-last everywhere; every Conv1d is a GEMM on overlapping rows (no im2col);
//   * 288 GB of HBM: every activation the backward needs is kept (no gradient checkpointing /
//     recompute, which the reference enables to fit 16 GB cards, wav2vec_train.py:329);
//   * q/k/v projections are one [3H,H] GEMM; bias, GELU, dropout and GELU-gradient live in GEMM epilogues;
//     residual + dropout + LayerNorm is one kernel; dropout masks are recomputed from a counter hash;
//   * weight gradients are deterministic split-K GEMMs reading both operands K-major (hardware transpose
//     read), written straight into one flat fp32 gradient buffer that RCCL all-reduces in one call.
#include <algorithm>
#include <string>
#include <vector>
#include <string.h>

#include "kernels.h"

namespace {

constexpr int WGRAD_SETS = 12;  // most buffer sets of the queued weight-gradient operands = layers per grouped launch (4 products each: ssak_gemm_bf16_grouped takes 48)
bool wgrad_all() {
  static const bool v = SSAK_DEV_ENV("SSAK_WGRAD_ALL") != nullptr;  // development switch: see the workspace plan
  return v;
}


struct PInfo {
  std::string name;
  std::vector<long> shape;
  long offset = 0, numel = 0;
  int region = 0;  // 0 trainable, 1 feature encoder
};

struct LayerP {
  long wqkv, bqkv, wo, bo, ln1w, ln1b, w1, b1, w2, b2, ln2w, ln2b;
};

inline long align_up(long x, long a) { return (x + a - 1) / a * a; }

struct Carver {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t o = off;
    off = (size_t)align_up((long)(off + bytes), 256);
    return o;
  }
};

struct LayerBuf {
  size_t qkv, P, Pd, ctx, r1, x1, f1pre, f1, r2, st;  // st: mean1|rstd1|mean2|rstd2 (4*M floats)
  size_t lse;  // fused attention: log-sum-exp per (b, head, query) instead of the probability matrices
};

struct Plan {
  int B = 0, T = 0, F = 0, M = 0, Fp = 0, training = 0;
  int Tl[8] = {0};
  long pg_rows = 0;
  size_t bufA, bufB, feat, stats0, ln0, st0, h0, pgx, pc_pre, pc, h1, stE, tmpH, S, xf, flens;
  std::vector<size_t> x;       // L+1 layer inputs/outputs
  std::vector<LayerBuf> lb;
  // backward temporaries
  size_t dlog, dA, dB, dY, dC, dI, dqkv, dSb, pgdy, dwf, slab, dln0, scratchH, lnpart, redring, redring_floats;
  // sets of the buffers the weight-gradient products read (dy of both LayerNorms, dI, dqkv): those products are queued and
  // launched several layers at a time (two under data parallelism, up to WGRAD_SETS on a single GPU), so a layer's set
  // must survive the backward of the layers queued after it
  size_t dY1 = 0, dYb[WGRAD_SETS] = {0}, dIb[WGRAD_SETS] = {0}, dqkvb[WGRAD_SETS] = {0}, dY1b[WGRAD_SETS] = {0};
  int wgrad_sets = 2;
  // Whisper front end: RS2 rows per utterance after conv2, RS1 = 2*RS2 before it
  int Tin = 0, RS1 = 0, RS2 = 0;
  bool fused_attn = false;
  size_t delta = 0;
  // --no_freeze: every conv activation / pre-activation is kept, plus backward temporaries
  bool fe_train = false;
  size_t act[8] = {0}, cpre[8] = {0}, fe_dp = 0, fe_da = 0, fe_db = 0, fe_dxcol = 0, fe_slab = 0, fe_dwr = 0, fe_c0 = 0;
  size_t fe_st[8] = {0};  // layer-norm feature encoder: per-frame mean | rstd of every conv layer's LayerNorm
  size_t melcl, h1pad, pre1, wpre2, we, dpre2pad, dxcol, dpre1pad, dwr;
  size_t slab_bytes = 0;
  size_t total = 0;
};

}  // namespace

struct ssak_w2v2 {
  ssak_w2v2_config cfg;
  std::vector<PInfo> params;
  long n_total = 0, n_train = 0;
  // parameter offsets (elements into the flat buffers)
  long p_mse, p_fpln_w, p_fpln_b, p_fp_w, p_fp_b, p_pc_b, p_pc_g, p_pc_v, p_eln_w, p_eln_b, p_lm_w, p_lm_b;
  long p_conv_w[8], p_conv_b[8], p_cln_w[8], p_cln_b[8];
  long p_c1w = 0, p_c1b = 0, p_c2w = 0, p_c2b = 0, p_pos = 0;  // Whisper front end  // conv weight / bias / per-layer norm (group: layer 0 only)
  std::vector<size_t> hres;  // stable-LN: residual-stream buffer that feeds the LayerNorm producing x[l]
  std::vector<LayerP> lp;
  // bound buffers (caller-owned)
  float* P = nullptr;
  float* G = nullptr;
  bf16* W = nullptr;
  // engine-owned derived weights
  // engine-owned derived weights, in the activation type of the mode (bf16, or float when cfg.exact)
  void* conv_w[8] = {nullptr};
  void* pc_wf = nullptr;
  void* pc_wb = nullptr;
  bf16* pc_wf_frag = nullptr;  // fragment-ordered copies for the direct positional-conv kernel (posconv.hip)
  bf16* pc_wb_frag = nullptr;
  float* pc_norms = nullptr;  // [2K]: ||v||^2 per tap | scratch
  Plan plan;
  bool have_fwd = false;
  bool fwd_hidden = false;  // the kept forward ended at the hidden state (ssak_w2v2_forward_hidden)
  uint64_t seed = 0;
  const uint8_t* spec_mask = nullptr;
  const int32_t* lens = nullptr;
  const float* last_input = nullptr;  // input_values of the last forward (conv0 backward recomputes from it)
  std::vector<int> keep;  // LayerDrop decisions of the last forward
  ssak_grad_ready_fn on_ready = nullptr;  // announces finished gradient ranges during backward (bucketed all-reduce)
  void* on_ready_user = nullptr;
  // optimizer on a side stream: the forward waits for this event before its first read of a trainable parameter
  hipEvent_t params_ready = nullptr, stall_begin = nullptr, stall_end = nullptr;
  // per-handle execution options (ssak_w2v2_set_option)
  int dynamic_tiles = 0;
  size_t xin[65] = {0};  // workspace offset of the INPUT of layer l (x[l]; a LayerDrop-skipped post-LN layer passes its input on: no copy)
  int raw_input = 0;  // SSAK_W2V2_OPT_RAW_INPUT: input_values are raw full-length waveforms (group-norm feature encoder, frozen)
  int attn_bwd_mode = SSAK_ATTN_BWD_DEFAULT;
  int posconv_direct = 1;
  int fragment_weights = 0;
  // Fragment-ordered copies of the encoder layers' projection weights for the B-direct GEMM form (gemm_p8.hip): engine-owned,
  // allocated by the first training forward and refreshed by every training forward (the optimizer rewrites the bf16 shadow
  // between steps).  Per layer six copies: qkv | out | ffn-down in the forward orientation, qkv | out | ffn-up in the
  // input-gradient orientation (the weight read K-major).
  bf16* wfrag = nullptr;
  size_t wfrag_layer = 0;      // elements per layer
  size_t wfrag_off[6] = {0};   // element offsets inside a layer's block
  bool wfrag_use[6] = {false}; // which of the six products take the B-direct form at wfrag_M rows
  int wfrag_M = 0;
  bool wfrag_valid = false;    // the copies match the shadow: set by a training forward, cleared by bind and evaluation forwards
  const bf16* frag(int layer, int which) const {
    return (wfrag_valid && wfrag_use[which] && keep[layer]) ? wfrag + (size_t)layer * wfrag_layer + wfrag_off[which] : nullptr;
  }
  // Transposed copies of the layers' projection matrices (qkv^T [H][3H] | out^T [H][H] | ffn-up^T [H][I] | ffn-down^T [I][H]):
  // the B operand of the backward's input-gradient products, K-contiguous (SSAK_W2V2_OPT_TRANSPOSED_WEIGHTS).  Engine-owned,
  // allocated by the first training forward, refreshed by every training forward for the layers it keeps.
  int transposed_weights = 1;
  bf16* wtr = nullptr;
  size_t wtr_layer = 0;
  size_t wtr_off[4] = {0};
  bool wtr_valid = false;
  const bf16* wt(int layer, int which) const {
    return (wtr_valid && keep[layer]) ? wtr + (size_t)layer * wtr_layer + wtr_off[which] : nullptr;
  }
};

namespace {

// stream ids of the dropout sites (mask bit = hash(seed, stream, element offset))
enum : uint32_t { DS_FEATPROJ = 1, DS_ENCIN = 2, DS_FINAL = 3, DS_LAYER0 = 16 };
inline uint32_t ds_attn(int l) { return DS_LAYER0 + 4 * l; }
inline uint32_t ds_hid1(int l) { return DS_LAYER0 + 4 * l + 1; }
inline uint32_t ds_act(int l) { return DS_LAYER0 + 4 * l + 2; }
inline uint32_t ds_hid2(int l) { return DS_LAYER0 + 4 * l + 3; }

int conv_len(int L, int k, int s) { return (L - k) / s + 1; }

void add_param(ssak_w2v2* e, long& cursor, const std::string& name, std::vector<long> shape, int region, long* off_out) {
  PInfo pi;
  pi.name = name;
  pi.shape = shape;
  pi.numel = 1;
  for (long d : shape) pi.numel *= d;
  pi.offset = cursor;
  pi.region = region;
  cursor = align_up(cursor + pi.numel, 8);
  if (off_out) *off_out = pi.offset;
  e->params.push_back(pi);
}

void build_param_table_whisper(ssak_w2v2* e) {
  const ssak_w2v2_config& c = e->cfg;
  const long H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size, NM = c.num_mel_bins;
  long cur = 0, dummy;
  e->lp.resize(c.num_layers);
  add_param(e, cur, "encoder.conv1.weight", {H, NM, 3}, 0, &e->p_c1w);
  add_param(e, cur, "encoder.conv2.weight", {H, H, 3}, 0, &e->p_c2w);
  for (int l = 0; l < c.num_layers; ++l) {
    const std::string p = "encoder.layers." + std::to_string(l) + ".";
    LayerP& L = e->lp[l];
    add_param(e, cur, p + "self_attn.q_proj.weight", {H, H}, 0, &L.wqkv);
    add_param(e, cur, p + "self_attn.k_proj.weight", {H, H}, 0, &dummy);
    add_param(e, cur, p + "self_attn.v_proj.weight", {H, H}, 0, &dummy);
    add_param(e, cur, p + "self_attn.out_proj.weight", {H, H}, 0, &L.wo);
    add_param(e, cur, p + "fc1.weight", {I, H}, 0, &L.w1);
    add_param(e, cur, p + "fc2.weight", {H, I}, 0, &L.w2);
  }
  add_param(e, cur, "ctc_head.weight", {V, H}, 0, &e->p_lm_w);
  add_param(e, cur, "encoder.conv1.bias", {H}, 0, &e->p_c1b);
  add_param(e, cur, "encoder.conv2.bias", {H}, 0, &e->p_c2b);
  for (int l = 0; l < c.num_layers; ++l) {
    const std::string p = "encoder.layers." + std::to_string(l) + ".";
    LayerP& L = e->lp[l];
    add_param(e, cur, p + "self_attn.q_proj.bias", {H}, 0, &L.bqkv);
    // Whisper's k_proj has no bias: the slot exists only so that q|k|v biases form one [3H] vector; it stays zero
    // (its gradient is cleared after every backward) and is not part of the HF state dict.
    add_param(e, cur, p + "self_attn.k_proj.bias", {H}, 0, &dummy);
    add_param(e, cur, p + "self_attn.v_proj.bias", {H}, 0, &dummy);
    add_param(e, cur, p + "self_attn.out_proj.bias", {H}, 0, &L.bo);
    add_param(e, cur, p + "self_attn_layer_norm.weight", {H}, 0, &L.ln1w);
    add_param(e, cur, p + "self_attn_layer_norm.bias", {H}, 0, &L.ln1b);
    add_param(e, cur, p + "fc1.bias", {I}, 0, &L.b1);
    add_param(e, cur, p + "fc2.bias", {H}, 0, &L.b2);
    add_param(e, cur, p + "final_layer_norm.weight", {H}, 0, &L.ln2w);
    add_param(e, cur, p + "final_layer_norm.bias", {H}, 0, &L.ln2b);
  }
  add_param(e, cur, "encoder.layer_norm.weight", {H}, 0, &e->p_eln_w);
  add_param(e, cur, "encoder.layer_norm.bias", {H}, 0, &e->p_eln_b);
  add_param(e, cur, "ctc_head.bias", {V}, 0, &e->p_lm_b);
  e->n_train = cur;
  add_param(e, cur, "encoder.embed_positions.weight", {c.max_source_positions, H}, 1, &e->p_pos);  // fixed sinusoids
  e->n_total = cur;
}

void build_param_table(ssak_w2v2* e) {
  if (e->cfg.arch == 1) return build_param_table_whisper(e);
  const ssak_w2v2_config& c = e->cfg;
  const long H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size, C = c.conv_dim[c.num_conv_layers - 1];
  const long K = c.num_conv_pos_embeddings, cg = H / c.num_conv_pos_embedding_groups;
  long cur = 0;
  e->lp.resize(c.num_layers);
  // region 0a: trainable matrices (weight-decayed set of HF Trainer, trainer.py:1013-1024)
  add_param(e, cur, "wav2vec2.masked_spec_embed", {H}, 0, &e->p_mse);
  add_param(e, cur, "wav2vec2.feature_projection.projection.weight", {H, C}, 0, &e->p_fp_w);
  add_param(e, cur, "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0", {1, 1, K}, 0, &e->p_pc_g);
  add_param(e, cur, "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1", {H, cg, K}, 0, &e->p_pc_v);
  for (int l = 0; l < c.num_layers; ++l) {
    const std::string p = "wav2vec2.encoder.layers." + std::to_string(l) + ".";
    LayerP& L = e->lp[l];
    long dummy;
    add_param(e, cur, p + "attention.q_proj.weight", {H, H}, 0, &L.wqkv);  // q,k,v contiguous = one [3H,H] matrix
    add_param(e, cur, p + "attention.k_proj.weight", {H, H}, 0, &dummy);
    add_param(e, cur, p + "attention.v_proj.weight", {H, H}, 0, &dummy);
    add_param(e, cur, p + "attention.out_proj.weight", {H, H}, 0, &L.wo);
    add_param(e, cur, p + "feed_forward.intermediate_dense.weight", {I, H}, 0, &L.w1);
    add_param(e, cur, p + "feed_forward.output_dense.weight", {H, I}, 0, &L.w2);
  }
  add_param(e, cur, "lm_head.weight", {V, H}, 0, &e->p_lm_w);
  // region 0b: trainable vectors (biases, LayerNorm affine: no weight decay)
  add_param(e, cur, "wav2vec2.feature_projection.layer_norm.weight", {C}, 0, &e->p_fpln_w);
  add_param(e, cur, "wav2vec2.feature_projection.layer_norm.bias", {C}, 0, &e->p_fpln_b);
  add_param(e, cur, "wav2vec2.feature_projection.projection.bias", {H}, 0, &e->p_fp_b);
  add_param(e, cur, "wav2vec2.encoder.pos_conv_embed.conv.bias", {H}, 0, &e->p_pc_b);
  add_param(e, cur, "wav2vec2.encoder.layer_norm.weight", {H}, 0, &e->p_eln_w);
  add_param(e, cur, "wav2vec2.encoder.layer_norm.bias", {H}, 0, &e->p_eln_b);
  for (int l = 0; l < c.num_layers; ++l) {
    const std::string p = "wav2vec2.encoder.layers." + std::to_string(l) + ".";
    LayerP& L = e->lp[l];
    long dummy;
    add_param(e, cur, p + "attention.q_proj.bias", {H}, 0, &L.bqkv);
    add_param(e, cur, p + "attention.k_proj.bias", {H}, 0, &dummy);
    add_param(e, cur, p + "attention.v_proj.bias", {H}, 0, &dummy);
    add_param(e, cur, p + "attention.out_proj.bias", {H}, 0, &L.bo);
    add_param(e, cur, p + "layer_norm.weight", {H}, 0, &L.ln1w);
    add_param(e, cur, p + "layer_norm.bias", {H}, 0, &L.ln1b);
    add_param(e, cur, p + "feed_forward.intermediate_dense.bias", {I}, 0, &L.b1);
    add_param(e, cur, p + "feed_forward.output_dense.bias", {H}, 0, &L.b2);
    add_param(e, cur, p + "final_layer_norm.weight", {H}, 0, &L.ln2w);
    add_param(e, cur, p + "final_layer_norm.bias", {H}, 0, &L.ln2b);
  }
  add_param(e, cur, "lm_head.bias", {V}, 0, &e->p_lm_b);
  e->n_train = cur;
  // region 1: feature encoder (frozen by default, wav2vec_train.py:326-327)
  long cin = 1;
  for (int i = 0; i < c.num_conv_layers; ++i) {
    const std::string p = "wav2vec2.feature_extractor.conv_layers." + std::to_string(i) + ".";
    add_param(e, cur, p + "conv.weight", {c.conv_dim[i], cin, c.conv_kernel[i]}, 1, &e->p_conv_w[i]);
    if (c.conv_bias) add_param(e, cur, p + "conv.bias", {c.conv_dim[i]}, 1, &e->p_conv_b[i]);
    if (i == 0 || c.feat_extract_norm == 1) {
      add_param(e, cur, p + "layer_norm.weight", {c.conv_dim[i]}, 1, &e->p_cln_w[i]);
      add_param(e, cur, p + "layer_norm.bias", {c.conv_dim[i]}, 1, &e->p_cln_b[i]);
    }
    cin = c.conv_dim[i];
  }
  e->n_total = cur;
}

int check_config(const ssak_w2v2_config& c) {
  if (c.arch == 1) {
    SSAK_REQUIRE(c.num_mel_bins > 0 && c.num_mel_bins % 8 == 0 && c.max_source_positions > 0, "whisper: num_mel_bins must be a multiple of 8");
    SSAK_REQUIRE(c.hidden_size % c.num_heads == 0 && (c.hidden_size / c.num_heads) % 8 == 0, "whisper: head_dim must be a multiple of 8");
    SSAK_REQUIRE(c.hidden_size % 8 == 0 && c.intermediate_size % 8 == 0 && c.vocab_size % 8 == 0, "whisper: d_model / ffn / vocab must be multiples of 8");
    SSAK_REQUIRE(c.num_layers >= 1 && c.num_layers <= 64, "whisper: num_layers out of range");
    return SSAK_OK;
  }
  SSAK_REQUIRE(c.arch == 0, "w2v2: arch must be 0 (wav2vec2) or 1 (whisper encoder)");
  SSAK_REQUIRE(c.num_conv_layers >= 2 && c.num_conv_layers <= 8, "w2v2: num_conv_layers %d unsupported", c.num_conv_layers);
  SSAK_REQUIRE(c.feat_extract_norm == 0 || c.feat_extract_norm == 1, "w2v2: feat_extract_norm must be 0 (group) or 1 (layer)");
  SSAK_REQUIRE(c.hidden_size % c.num_heads == 0 && (c.hidden_size / c.num_heads) % 8 == 0, "w2v2: head_dim must be a multiple of 8");
  SSAK_REQUIRE(c.hidden_size % 8 == 0 && c.intermediate_size % 8 == 0 && c.vocab_size % 8 == 0,
               "w2v2: hidden/intermediate/vocab sizes must be multiples of 8 (pad the vocabulary)");
  SSAK_REQUIRE(c.hidden_size % c.num_conv_pos_embedding_groups == 0 &&
                   (c.hidden_size / c.num_conv_pos_embedding_groups) % 8 == 0, "w2v2: pos-conv group width must be a multiple of 8");
  for (int i = 0; i < c.num_conv_layers; ++i) {
    SSAK_REQUIRE(c.conv_dim[i] % 8 == 0, "w2v2: conv_dim must be multiples of 8");
  }
  SSAK_REQUIRE(c.num_layers >= 1 && c.num_layers <= 64, "w2v2: num_layers out of range");
  return SSAK_OK;
}

int make_plan(const ssak_w2v2* e, int B, int T, int training, Plan& p) {
  const ssak_w2v2_config& c = e->cfg;
  p = Plan();
  p.B = B;
  p.T = T;
  p.training = training;
  const bool whisper = c.arch == 1;
  int L = T;
  if (!whisper) {
    for (int i = 0; i < c.num_conv_layers; ++i) {
      L = conv_len(L, c.conv_kernel[i], c.conv_stride[i]);
      SSAK_REQUIRE(L > 0, "w2v2: input of %d samples is too short for the feature encoder", T);
      p.Tl[i] = L;
    }
  } else {
    SSAK_REQUIRE(T >= 2 && (T & 1) == 0 && T / 2 <= c.max_source_positions, "whisper: %d feature frames (need even, <= %d)", T, 2 * c.max_source_positions);
    L = T / 2;
    p.Tin = T;
    p.RS2 = (int)align_up(L + 1, 4);
    p.RS1 = 2 * p.RS2;
  }
  p.F = L;
  p.M = B * L;
  p.Fp = (int)align_up(L, 8);
  const long H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size, nh = c.num_heads;
  const long C = whisper ? 8 : c.conv_dim[c.num_conv_layers - 1], K = whisper ? 8 : c.num_conv_pos_embeddings;
  const long G = whisper ? 1 : c.num_conv_pos_embedding_groups;
  const long M = p.M;
  p.pg_rows = K / 2 + (long)B * (p.F + K) + K;
  Carver cv;
  const size_t b2 = c.exact ? sizeof(float) : sizeof(bf16);  // activation element size
  if (whisper) {
    const long NM = c.num_mel_bins;
    p.melcl = cv.take(((size_t)B * p.RS1 + 8) * NM * b2);
    // (+ 8 zero rows: the stride-2 window of the last padding row of the weight-gradient GEMM reaches one row past B * RS1;
    // its dy row is zero, but 0 x NaN from an unwritten neighbour buffer would still poison the sum)
    p.h1pad = cv.take(((size_t)B * p.RS1 + 8) * H * b2);
    p.pre1 = cv.take((size_t)B * p.RS1 * H * b2);
    p.we = cv.take((size_t)M * H * b2);
    p.wpre2 = cv.take((size_t)M * H * b2);
    if (training) {
      p.dpre2pad = cv.take((size_t)B * p.RS2 * H * b2);
      p.dxcol = cv.take((size_t)M * 3 * H * b2);
      p.dpre1pad = cv.take((size_t)B * p.RS1 * H * b2);
      p.dwr = cv.take((size_t)H * 3 * std::max(H, NM) * sizeof(float));
    }
  }
  p.fe_train = !whisper && training && !c.freeze_feature_encoder;
  if (p.fe_train) {
    const int nc = c.num_conv_layers;
    size_t max_col = 0, max_w = 0;
    for (int i = 0; i < nc; ++i) {
      if (i < nc - 1) p.act[i] = cv.take((size_t)B * p.Tl[i] * c.conv_dim[i] * b2);
      if (c.feat_extract_norm == 1) {  // pre-LayerNorm conv outputs of ALL layers + their statistics
        if (i == 0) p.cpre[0] = cv.take((size_t)B * p.Tl[0] * c.conv_dim[0] * b2);
        p.fe_st[i] = cv.take((size_t)2 * B * p.Tl[i] * sizeof(float));
      }
      if (i > 0) {
        p.cpre[i] = cv.take((size_t)B * p.Tl[i] * c.conv_dim[i] * b2);
        max_col = std::max(max_col, (size_t)B * p.Tl[i] * c.conv_kernel[i] * c.conv_dim[i - 1]);
        max_w = std::max(max_w, (size_t)c.conv_dim[i] * c.conv_kernel[i] * c.conv_dim[i - 1]);
      }
    }
    if (c.feat_extract_norm != 1) p.fe_dp = cv.take((size_t)B * p.Tl[1] * c.conv_dim[1] * b2);
    p.fe_da = cv.take((size_t)B * p.Tl[0] * c.conv_dim[0] * b2);
    p.fe_db = cv.take((size_t)B * p.Tl[1] * c.conv_dim[1] * b2);
    p.fe_dxcol = cv.take(max_col * b2);
    p.fe_slab = cv.take((size_t)B * max_w * sizeof(float));
    p.fe_dwr = cv.take(max_w * sizeof(float));
    p.fe_c0 = cv.take(std::max(k_conv0_bwd_scratch_floats(B, p.Tl[0], c.conv_dim[0]),
                               k_conv0_wgrad_scratch_floats(B, c.conv_dim[0], c.conv_kernel[0])) * sizeof(float));
    if (c.feat_extract_norm == 1) p.fe_dp = cv.take((size_t)B * p.Tl[0] * c.conv_dim[0] * b2);  // (d conv0 output lives here too)
  }
  p.bufA = cv.take(whisper ? 256 : (size_t)B * p.Tl[0] * c.conv_dim[0] * b2);
  p.bufB = cv.take(whisper ? 256 : (size_t)B * p.Tl[1] * c.conv_dim[1] * b2);
  p.feat = cv.take((size_t)M * C * b2);
  p.stats0 = cv.take(whisper ? 256 : k_conv0_stats_doubles(B, p.Tl[0], c.conv_dim[0]) * sizeof(double));
  p.flens = cv.take((size_t)B * sizeof(int32_t));
  p.ln0 = cv.take((size_t)M * C * b2);
  p.st0 = cv.take((size_t)2 * M * sizeof(float));
  p.h0 = cv.take((size_t)M * H * b2);
  p.pgx = cv.take((size_t)G * p.pg_rows * (H / G) * b2);
  p.pc_pre = cv.take((size_t)M * H * b2);
  p.pc = cv.take((size_t)M * H * b2);
  p.h1 = cv.take((size_t)M * H * b2);
  p.stE = cv.take((size_t)2 * M * sizeof(float));
  p.tmpH = cv.take((size_t)M * H * b2);
  p.fused_attn = !c.exact && k_attention_supported((int)H, (int)nh);
  p.S = cv.take(p.fused_attn ? 256 : (size_t)B * nh * p.F * p.Fp * b2);
  p.xf = cv.take((size_t)M * H * b2);
  const int nl = c.num_layers;
  p.x.resize(nl + 1);
  p.lb.resize(nl);
  const bool drop_attn = training && c.attention_dropout > 0.f;
  for (int l = 0; l <= nl; ++l)
    p.x[l] = (training || l < 2) ? cv.take((size_t)M * H * b2) : p.x[l & 1];
  for (int l = 0; l < nl; ++l) {
    LayerBuf& lb = p.lb[l];
    if (training || l == 0) {
      lb.qkv = cv.take((size_t)M * 3 * H * b2);
      lb.P = cv.take(p.fused_attn ? 256 : (size_t)B * nh * p.F * p.Fp * b2);
      lb.Pd = (drop_attn && !p.fused_attn) ? cv.take((size_t)B * nh * p.F * p.Fp * b2) : lb.P;
      lb.lse = cv.take((size_t)B * nh * p.F * sizeof(float));
      lb.ctx = cv.take((size_t)M * H * b2);
      lb.r1 = cv.take((size_t)M * H * b2);
      lb.x1 = cv.take((size_t)M * H * b2);
      lb.f1pre = cv.take((size_t)M * I * b2);
      lb.f1 = cv.take((size_t)M * I * b2);
      lb.r2 = cv.take((size_t)M * H * b2);
      lb.st = cv.take((size_t)4 * M * sizeof(float));
    } else {
      lb = p.lb[0];
    }
  }
  if (training) {
    p.dlog = cv.take((size_t)M * V * b2);
    p.dA = cv.take((size_t)M * H * b2);
    p.dB = cv.take((size_t)M * H * b2);
    p.dY = cv.take((size_t)M * H * b2);
    p.dY1 = cv.take((size_t)M * H * b2);
    p.dC = cv.take((size_t)M * H * b2);
    p.scratchH = cv.take((size_t)M * H * b2);
    p.dI = cv.take((size_t)M * I * b2);
    p.dqkv = cv.take((size_t)M * 3 * H * b2);
    p.dYb[0] = p.dY;
    p.dY1b[0] = p.dY1;
    p.dIb[0] = p.dI;
    p.dqkvb[0] = p.dqkv;
    // two sets (layer pairs) by default; SSAK_WGRAD_ALL=1: up to WGRAD_SETS layers per grouped launch on a single GPU -- measured
    // SLOWER (eleven layers as one 1 188-tile launch: 2 064 us per step against 2 014 us in pairs, the same 0.46 of the roof per
    // tile: the better fill of the rounds is lost again to the operand panels of many layers competing for the L2)
    p.wgrad_sets = wgrad_all() ? std::max(2, std::min(WGRAD_SETS, c.num_layers)) : 3;  // (three: a layer's last product may wait for the launch after next)
    for (int s2 = 1; s2 < p.wgrad_sets; ++s2) {
      p.dYb[s2] = cv.take((size_t)M * H * b2);
      p.dY1b[s2] = cv.take((size_t)M * H * b2);
      p.dIb[s2] = cv.take((size_t)M * I * b2);
      p.dqkvb[s2] = cv.take((size_t)M * 3 * H * b2);
    }
    p.dSb = cv.take(p.fused_attn ? 256 : (size_t)B * nh * p.F * p.Fp * b2);
    p.delta = cv.take((size_t)B * nh * p.F * sizeof(float));
    p.pgdy = cv.take((size_t)G * p.pg_rows * (H / G) * b2);
    p.dwf = cv.take((size_t)H * K * (H / G) * sizeof(float));
    p.dln0 = cv.take((size_t)M * C * b2);
    p.lnpart = cv.take((size_t)LN_BWD_BLOCKS * 3 * std::max(H, C) * sizeof(float));
    // partial sums of the reductions whose second stage is deferred to one launch per pair of encoder layers (kernels.h:
    // ReduceSink): per layer two LayerNorm backward slabs, the FFN bias sums of the dX epilogue and the qkv bias sums
    p.redring_floats = 3 * ((size_t)2 * LN_BWD_BLOCKS * 3 * H + (size_t)std::max(ssak_cdiv(M, 64), 64) * I +
                              std::max((size_t)64 * 3 * H, (size_t)B * ssak_cdiv(p.F, 64) * 3 * H));  // (qkv bias partials: <= one slot per 64 frames)
    p.redring = cv.take(p.redring_floats * sizeof(float));
    // split-K slabs: the largest weight-gradient product is [I,H] (or [3H,H]); at most 32 slices
    const size_t big = (size_t)std::max(std::max(I * H, 3 * H * H), std::max(H * C, V * H));
    p.slab_bytes = big * 32 * sizeof(float);
    p.slab = cv.take(p.slab_bytes);
  }
  p.total = cv.off;
  return SSAK_OK;
}

// Options of the handle whose forward / backward is running on this thread: the product builder below reads them, so the
// ~60 call sites do not each pass them.  Scoped to one entry-point call (EngineCall), never process-wide.
thread_local int t_dynamic_tiles = 0;
struct EngineCall {
  int saved;
  explicit EngineCall(const ssak_w2v2* e) : saved(t_dynamic_tiles) { t_dynamic_tiles = e->dynamic_tiles; }
  ~EngineCall() { t_dynamic_tiles = saved; }
};

struct Gemm {
  ssak_gemm_desc d;
  const void* A = nullptr;
  const void* B = nullptr;
  void* C = nullptr;
  const float* bias = nullptr;
  const void* aux_in = nullptr;
  void* aux_out = nullptr;
  bool f32 = false;  // fp32-exact mode: float operands through ssak_gemm_f32
  Gemm(int M, int N, int K, bool exact = false) : f32(exact) {
    memset(&d, 0, sizeof(d));
    d.dynamic_tiles = t_dynamic_tiles;  // the calling handle's option (EngineCall)
    d.M = M;
    d.N = N;
    d.K = K;
    d.nb1 = d.nb2 = 1;
    d.alpha = 1.f;
    d.split_k = 1;
    d.pads_are_zero = 1;  // every engine buffer with a padded leading dimension (P, dS: ld = round_up(F, 8)) is written with zero pads
  }
  Gemm& a(const void* p, long ld, bool km = false) {
    A = p;
    d.lda = ld;
    d.a_kmajor = km;
    return *this;
  }
  Gemm& b(const void* p, long ld, bool km = false) {
    B = p;
    d.ldb = ld;
    d.b_kmajor = km;
    return *this;
  }
  // B of an input-gradient product dX = dY W: the transposed copy of W (K-contiguous) when there is one, else W itself K-major
  Gemm& b_wt(const void* wt, long ld_t, const void* w, long ld_w) { return wt ? b(wt, ld_t, false) : b(w, ld_w, true); }
  Gemm& c(void* p, long ld, bool f32 = false) {
    C = p;
    d.ldc = ld;
    d.out_f32 = f32;
    return *this;
  }
  Gemm& batch(int nb1, int nb2, long sa1, long sa2, long sb1, long sb2, long sc1, long sc2) {
    d.nb1 = nb1;
    d.nb2 = nb2;
    d.sa1 = sa1;
    d.sa2 = sa2;
    d.sb1 = sb1;
    d.sb2 = sb2;
    d.sc1 = sc1;
    d.sc2 = sc2;
    return *this;
  }
  Gemm& with_bias(const float* bp, long s2 = 0) {
    bias = bp;
    d.bias_s2 = s2;
    return *this;
  }
  Gemm& epi(int e, const void* ain = nullptr, void* aout = nullptr) {
    d.epilogue = e;
    aux_in = ain;
    aux_out = aout;
    return *this;
  }
  Gemm& alpha(float a_) {
    d.alpha = a_;
    return *this;
  }
  // += column sums of the stored C into `out` [N] (run() must be given a workspace of ceil(M / 64) * N floats)
  Gemm& colsum(float* out) {
    d.colsum = 1;
    aux_out = out;
    return *this;
  }
  // fragment-ordered copy of B (ssak_gemm_desc.b_fragments; null = none)
  Gemm& bfrag(const void* frag) {
    if (!f32) d.b_fragments = frag;
    return *this;
  }
  Gemm& drop(float p, uint32_t stream, uint64_t seed) {
    d.drop_p = p;
    d.drop_stream = stream;
    d.drop_seed = seed;
    return *this;
  }
  int run(hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0) {
    if (f32) return ssak_gemm_f32(&d, A, B, C, bias, aux_in, aux_out, (void*)st);
    return ssak_gemm_bf16(&d, A, B, C, bias, aux_in, aux_out, ws, ws_bytes, (void*)st);
  }
  // weight-gradient form: long K, few output tiles -> deterministic split-K, sized by the library's cost model
  int run_wgrad(hipStream_t st, void* slab, size_t slab_bytes) {
    d.split_k = 0;
    return run(st, slab, slab_bytes);
  }
};

template <bool EXACT>
struct GemmX : Gemm {
  GemmX(int M, int N, int K) : Gemm(M, N, K, EXACT) {}
};

// Weight-gradient products are not on the critical path of the backward: they are queued and launched together as ONE grouped
// GEMM (ssak_gemm_bf16_grouped) instead of four launches per layer that each need split-K slabs and a reduction pass.  The
// queue is packed greedily into launches of at most one round of 256 x 256 tiles (see flush_gemms in the backward): ~2.4 base
// layers per launch (rounds 1-2: pairs of layers, 216 tiles, with the odd layer LayerDrop leaves over as four split-K
// launches); under data parallelism a layer's gradient range is announced for the bucketed all-reduce as soon as its last
// product has been launched.  All kept layers in one multi-round launch (SSAK_WGRAD_ALL=1; 1 188 tiles for eleven layers = 4.6
// rounds) was measured slower on one GPU: see the plan.
struct WgradQueue {
  static constexpr int CAP = 4 * WGRAD_SETS;  // four products per encoder layer
  ssak_gemm_desc d[CAP];
  const void* A[CAP];
  const void* B[CAP];
  void* C[CAP];
  int set[CAP];  // buffer set the product's operands live in
  int n = 0, layers = 0, tiles = 0;
  long pushed = 0, launched = 0;  // products ever queued / launched
  bool f32 = false;
  // gradient ranges to announce once the queued products have been launched, in LAYER ORDER: a data-parallel caller
  // pairs the k-th announcement of every rank in one collective, and LayerDrop decisions differ between ranks, so a
  // dropped layer's (zero) range must not overtake the kept layers still waiting here
  long ann_off[64];
  long ann_cnt[64];   // elements of the range (a layer's four matrices; the lm_head matrix)
  long ann_last[64];  // the range's last product (sequence number), -1: nothing to wait for (a dropped layer)
  int n_ann = 0;
  static int tiles_of(const ssak_gemm_desc& g) { return ssak_cdiv(g.M, 256) * ssak_cdiv(g.N, 256); }
  bool uses_set(int s) const {
    for (int i = 0; i < n; ++i)
      if (set[i] == s) return true;
    return false;
  }
  void push(const Gemm& g, int buffer_set) {
    set[n] = buffer_set;
    ++pushed;
    d[n] = g.d;
    A[n] = g.A;
    B[n] = g.B;
    C[n] = g.C;
    f32 = g.f32;
    tiles += tiles_of(g.d);
    ++n;
  }
  // launch what is queued: grouped when it fills at least 5/8 of a round of 256 workgroups, else one by one (split-K: the last,
  // partial launch of a backward and tiny test models)
  int flush(hipStream_t st, void* slab, size_t slab_bytes) {
    int rc = SSAK_OK;
    if (n > 0) {
      if (f32) {
        for (int i = 0; i < n && rc == SSAK_OK; ++i) rc = ssak_gemm_f32(&d[i], A[i], B[i], C[i], nullptr, nullptr, nullptr, (void*)st);
      } else if (tiles >= 160) {
        rc = ssak_gemm_bf16_grouped(d, n, A, B, C, (void*)st);
      } else {
        for (int i = 0; i < n && rc == SSAK_OK; ++i) {
          d[i].split_k = 0;
          rc = ssak_gemm_bf16(&d[i], A[i], B[i], C[i], nullptr, nullptr, nullptr, slab, slab_bytes, (void*)st);
        }
      }
    }
    launched += n;
    n = 0;
    tiles = 0;
    return rc;
  }
};

#define TRY(x)                    \
  do {                            \
    int rc__ = (x);               \
    if (rc__ != SSAK_OK) return rc__; \
  } while (0)

struct ConvChain {
  int n;
  int k[8], s[8];
};
__global__ void frame_lens_kernel(const int32_t* __restrict__ lens, int B, int T, ConvChain cc, int32_t* __restrict__ flens) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int L = min(max(lens[b], 0), T);
  for (int i = 0; i < cc.n; ++i) L = (L >= cc.k[i]) ? (L - cc.k[i]) / cc.s[i] + 1 : 0;
  flens[b] = L;
}

// The six per-layer products that can take fragment-ordered weights: N, K, orientation and the weight they read
struct FragProduct {
  int N, K, b_km;
  long ldb;
  long LayerP::*w;
};
int refresh_weight_fragments(ssak_w2v2* e, int M, const uint8_t* layer_keep, hipStream_t st) {
  const ssak_w2v2_config& c = e->cfg;
  const int H = c.hidden_size, I = c.intermediate_size, L = c.num_layers;
  const FragProduct pr[6] = {{3 * H, H, 0, H, &LayerP::wqkv}, {H, H, 0, H, &LayerP::wo}, {H, I, 0, I, &LayerP::w2},
                             {H, 3 * H, 1, H, &LayerP::wqkv}, {H, H, 1, H, &LayerP::wo}, {H, I, 1, H, &LayerP::w1}};
  if (!e->wfrag) {
    size_t off = 0;
    for (int i = 0; i < 6; ++i) {
      e->wfrag_off[i] = off;
      off += ssak_gemm_fragment_b_bytes(pr[i].N, pr[i].K) / sizeof(bf16);
    }
    e->wfrag_layer = off;
    SSAK_HIP(hipMalloc((void**)&e->wfrag, (size_t)L * off * sizeof(bf16)));
  }
  if (e->wfrag_M != M) {
    for (int i = 0; i < 6; ++i) {
      ssak_gemm_desc d;
      memset(&d, 0, sizeof(d));
      d.M = M, d.N = pr[i].N, d.K = pr[i].K;
      d.b_kmajor = pr[i].b_km;
      d.lda = pr[i].K, d.ldb = pr[i].ldb, d.ldc = pr[i].N;
      d.nb1 = d.nb2 = 1;
      d.alpha = 1.f;
      d.split_k = 1;
      d.pads_are_zero = 1;
      e->wfrag_use[i] = ssak_gemm_uses_fragments(&d) != 0;
    }
    e->wfrag_M = M;
  }
  std::vector<const void*> src;
  std::vector<void*> dst;
  std::vector<long> ldb;
  std::vector<int> N, K, km;
  for (int l = 0; l < L; ++l) {
    if (layer_keep && !layer_keep[l]) continue;
    for (int i = 0; i < 6; ++i) {
      if (!e->wfrag_use[i]) continue;
      src.push_back(e->W + e->lp[l].*(pr[i].w));
      dst.push_back(e->wfrag + (size_t)l * e->wfrag_layer + e->wfrag_off[i]);
      ldb.push_back(pr[i].ldb);
      N.push_back(pr[i].N);
      K.push_back(pr[i].K);
      km.push_back(pr[i].b_km);
    }
  }
  if (!src.empty())
    TRY(ssak_gemm_fragment_b_batched((int)src.size(), src.data(), ldb.data(), N.data(), K.data(), km.data(), dst.data(), (void*)st));
  e->wfrag_valid = true;
  return SSAK_OK;
}

// qkv | out | ffn-up | ffn-down, each [rows = out features][cols = in features] in the shadow -> [in][out]
int refresh_weight_transposes(ssak_w2v2* e, const uint8_t* layer_keep, hipStream_t st) {
  const ssak_w2v2_config& c = e->cfg;
  const int H = c.hidden_size, I = c.intermediate_size, L = c.num_layers;
  const int rows[4] = {3 * H, H, I, H}, cols[4] = {H, H, H, I};
  long LayerP::*const w[4] = {&LayerP::wqkv, &LayerP::wo, &LayerP::w1, &LayerP::w2};
  if (!e->wtr) {
    size_t off = 0;
    for (int i = 0; i < 4; ++i) {
      e->wtr_off[i] = off;
      off += (size_t)rows[i] * cols[i];
    }
    e->wtr_layer = off;
    SSAK_HIP(hipMalloc((void**)&e->wtr, (size_t)L * off * sizeof(bf16)));
  }
  std::vector<const bf16*> src;
  std::vector<bf16*> dst;
  std::vector<int> R, Cc;
  for (int l = 0; l < L; ++l) {
    if (layer_keep && !layer_keep[l]) continue;
    for (int i = 0; i < 4; ++i) {
      src.push_back(e->W + e->lp[l].*(w[i]));
      dst.push_back(e->wtr + (size_t)l * e->wtr_layer + e->wtr_off[i]);
      R.push_back(rows[i]);
      Cc.push_back(cols[i]);
    }
  }
  if (!src.empty()) TRY(k_transpose_bf16_batched((int)src.size(), src.data(), dst.data(), R.data(), Cc.data(), st));
  e->wtr_valid = true;
  return SSAK_OK;
}

}  // namespace

// =================================================================================================== C ABI
extern "C" int ssak_w2v2_create(const ssak_w2v2_config* cfg, ssak_w2v2** out) {
  SSAK_REQUIRE(cfg && out, "w2v2_create: null pointer");
  TRY(check_config(*cfg));
  ssak_w2v2* e = new ssak_w2v2();
  e->cfg = *cfg;
  build_param_table(e);
  const ssak_w2v2_config& c = e->cfg;
  const long H = c.hidden_size, K = c.num_conv_pos_embeddings;
  const long cg = c.arch == 1 ? 8 : H / c.num_conv_pos_embedding_groups;
  if (c.arch == 1) {
    const size_t wsz = c.exact ? sizeof(float) : sizeof(bf16);
    SSAK_HIP(hipMalloc((void**)&e->conv_w[1], (size_t)H * 3 * c.num_mel_bins * wsz));
    SSAK_HIP(hipMalloc((void**)&e->conv_w[2], (size_t)H * 3 * H * wsz));
    *out = e;
    return SSAK_OK;
  }
  long cin = 1;
  const size_t esz = c.exact ? sizeof(float) : sizeof(bf16);
  for (int i = 0; i < c.num_conv_layers; ++i) {
    if (i > 0) SSAK_HIP(hipMalloc((void**)&e->conv_w[i], (size_t)c.conv_dim[i] * cin * c.conv_kernel[i] * esz));
    cin = c.conv_dim[i];
  }
  SSAK_HIP(hipMalloc((void**)&e->pc_wf, (size_t)H * K * cg * esz));
  SSAK_HIP(hipMalloc((void**)&e->pc_wb, (size_t)H * K * cg * esz));
  if (!c.exact && k_posconv_direct_supported((int)H, c.num_conv_pos_embedding_groups, (int)K)) {
    SSAK_HIP(hipMalloc((void**)&e->pc_wf_frag, (size_t)H * K * cg * sizeof(bf16)));
    SSAK_HIP(hipMalloc((void**)&e->pc_wb_frag, (size_t)H * K * cg * sizeof(bf16)));
  }
  SSAK_HIP(hipMalloc((void**)&e->pc_norms, (size_t)(2 + c.hidden_size) * K * sizeof(float)));  // norms | dot | [H][K] partials
  *out = e;
  return SSAK_OK;
}

extern "C" void ssak_w2v2_destroy(ssak_w2v2* e) {
  if (!e) return;
  for (int i = 0; i < 8; ++i)
    if (e->conv_w[i]) (void)hipFree(e->conv_w[i]);
  if (e->pc_wf) (void)hipFree(e->pc_wf);
  if (e->pc_wb) (void)hipFree(e->pc_wb);
  if (e->pc_wf_frag) (void)hipFree(e->pc_wf_frag);
  if (e->wfrag) (void)hipFree(e->wfrag);
  if (e->wtr) (void)hipFree(e->wtr);
  if (e->pc_wb_frag) (void)hipFree(e->pc_wb_frag);
  if (e->pc_norms) (void)hipFree(e->pc_norms);
  delete e;
}

extern "C" long ssak_w2v2_num_params(const ssak_w2v2* e) { return e ? e->n_total : 0; }
extern "C" long ssak_w2v2_num_trainable(const ssak_w2v2* e) {
  if (!e) return 0;
  if (e->cfg.arch == 1) return e->n_train;  // everything but the fixed position table
  return e->cfg.freeze_feature_encoder ? e->n_train : e->n_total;
}
extern "C" int ssak_w2v2_param_count(const ssak_w2v2* e) { return e ? (int)e->params.size() : 0; }
extern "C" int ssak_w2v2_param_info(const ssak_w2v2* e, int index, char* name, int name_cap, long* offset, long* numel,
                                    int* ndim, long* shape4) {
  SSAK_REQUIRE(e && index >= 0 && index < (int)e->params.size(), "w2v2_param_info: bad index %d", index);
  const PInfo& pi = e->params[index];
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", pi.name.c_str());
  if (offset) *offset = pi.offset;
  if (numel) *numel = pi.numel;
  if (ndim) *ndim = (int)pi.shape.size();
  if (shape4)
    for (size_t i = 0; i < 4; ++i) shape4[i] = i < pi.shape.size() ? pi.shape[i] : 1;
  return SSAK_OK;
}

extern "C" int ssak_w2v2_bind(ssak_w2v2* e, float* params, float* grads, void* shadow_bf16) {
  SSAK_REQUIRE(e && params && shadow_bf16, "w2v2_bind: null pointer");
  SSAK_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)shadow_bf16) & 15) == 0, "w2v2_bind: buffers must be 16-byte aligned");
  e->P = params;
  e->G = grads;
  e->W = (bf16*)shadow_bf16;
  e->wfrag_valid = false;
  e->wtr_valid = false;
  return SSAK_OK;
}

extern "C" int ssak_w2v2_set_option(ssak_w2v2* e, int option, int value) {
  SSAK_REQUIRE(e, "w2v2_set_option: null handle");
  if (option == SSAK_W2V2_OPT_DYNAMIC_TILES) {
    e->dynamic_tiles = value ? 1 : 0;
  } else if (option == SSAK_W2V2_OPT_ATTENTION_BWD) {
    SSAK_REQUIRE(value == SSAK_ATTN_BWD_DEFAULT || value == SSAK_ATTN_BWD_TWO_KERNEL, "w2v2_set_option: attention backward form %d (the single-pass form was removed in ABI 400)", value);
    e->attn_bwd_mode = value;
  } else if (option == SSAK_W2V2_OPT_POSCONV_DIRECT) {
    e->posconv_direct = value ? 1 : 0;
  } else if (option == SSAK_W2V2_OPT_FRAGMENT_WEIGHTS) {
    e->fragment_weights = value ? 1 : 0;
  } else if (option == SSAK_W2V2_OPT_TRANSPOSED_WEIGHTS) {
    e->transposed_weights = value ? 1 : 0;
  } else if (option == SSAK_W2V2_OPT_RAW_INPUT) {
    SSAK_REQUIRE(!value || (e->cfg.arch == 0 && e->cfg.feat_extract_norm == 0), "w2v2_set_option: raw input is folded into the GroupNorm of the wav2vec2 group-norm feature encoder only");
    e->raw_input = value ? 1 : 0;
  } else {
    ssak_set_error("w2v2_set_option: unknown option %d", option);
    return SSAK_ERR_INVALID;
  }
  return SSAK_OK;
}

extern "C" int ssak_w2v2_set_grad_ready_callback(ssak_w2v2* e, ssak_grad_ready_fn fn, void* user) {
  SSAK_REQUIRE(e, "w2v2_set_grad_ready_callback: null handle");
  e->on_ready = fn;
  e->on_ready_user = user;
  return SSAK_OK;
}

extern "C" int ssak_w2v2_grad_ranges(const ssak_w2v2_config* cfg, long* offsets, long* counts, int cap) {
  SSAK_REQUIRE(cfg && offsets && counts && cap > 0, "w2v2_grad_ranges: null pointer");
  TRY(check_config(*cfg));
  ssak_w2v2 e;
  e.cfg = *cfg;
  build_param_table(&e);
  const ssak_w2v2_config& c = e.cfg;
  const long H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size;
  const long n_grad = (c.arch == 1 || c.freeze_feature_encoder) ? e.n_train : e.n_total;
  const long layer_span = (e.lp[0].w2 + H * I) - e.lp[0].wqkv;
  int n = 0;
  auto put = [&](long off, long cnt) {
    if (cnt <= 0) return;
    if (n < cap) {
      offsets[n] = off;
      counts[n] = cnt;
    }
    ++n;
  };
  put(e.p_lm_w, V * H);
  for (int l = c.num_layers - 1; l >= 0; --l) put(e.lp[l].wqkv, layer_span);
  put(0, e.lp[0].wqkv);
  put(e.p_lm_w + V * H, n_grad - (e.p_lm_w + V * H));
  return n;
}

extern "C" int ssak_w2v2_set_param_event(ssak_w2v2* e, void* params_ready, void* stall_begin, void* stall_end) {
  SSAK_REQUIRE(e, "w2v2_set_param_event: null handle");
  e->params_ready = (hipEvent_t)params_ready;
  e->stall_begin = (hipEvent_t)stall_begin;
  e->stall_end = (hipEvent_t)stall_end;
  return SSAK_OK;
}

extern "C" int ssak_w2v2_sync_weights(ssak_w2v2* e, int full, void* stream) {
  SSAK_REQUIRE(e && e->P && e->W, "w2v2_sync_weights: bind the parameter buffers first");
  hipStream_t st = (hipStream_t)stream;
  const ssak_w2v2_config& c = e->cfg;
  if (c.arch == 1) {
    if (full) TRY(k_cast_f32_bf16(e->P, e->W, e->n_total, st));
    // the conv weights are trainable here: their [Co][k][Ci] GEMM layouts follow every optimizer step
    if (c.exact) {
      TRY(k_conv_weight_rearrange_t<float>(e->P + e->p_c1w, (float*)e->conv_w[1], c.hidden_size, c.num_mel_bins, 3, st));
      TRY(k_conv_weight_rearrange_t<float>(e->P + e->p_c2w, (float*)e->conv_w[2], c.hidden_size, c.hidden_size, 3, st));
    } else {
      TRY(k_conv_weight_rearrange_t<bf16>(e->P + e->p_c1w, (bf16*)e->conv_w[1], c.hidden_size, c.num_mel_bins, 3, st));
      TRY(k_conv_weight_rearrange_t<bf16>(e->P + e->p_c2w, (bf16*)e->conv_w[2], c.hidden_size, c.hidden_size, 3, st));
    }
    return SSAK_OK;
  }
  if (full) {
    TRY(k_cast_f32_bf16(e->P, e->W, e->n_total, st));
    long cin = c.conv_dim[0];
    for (int i = 1; i < c.num_conv_layers; ++i) {
      if (c.exact)
        TRY(k_conv_weight_rearrange_t<float>(e->P + e->p_conv_w[i], (float*)e->conv_w[i], c.conv_dim[i], (int)cin, c.conv_kernel[i], st));
      else
        TRY(k_conv_weight_rearrange_t<bf16>(e->P + e->p_conv_w[i], (bf16*)e->conv_w[i], c.conv_dim[i], (int)cin, c.conv_kernel[i], st));
      cin = c.conv_dim[i];
    }
  }
  if (!full && !c.freeze_feature_encoder) {  // trainable feature encoder: the conv GEMM layouts follow the optimizer
    long cin2 = c.conv_dim[0];
    for (int i = 1; i < c.num_conv_layers; ++i) {
      if (c.exact)
        TRY(k_conv_weight_rearrange_t<float>(e->P + e->p_conv_w[i], (float*)e->conv_w[i], c.conv_dim[i], (int)cin2, c.conv_kernel[i], st));
      else
        TRY(k_conv_weight_rearrange_t<bf16>(e->P + e->p_conv_w[i], (bf16*)e->conv_w[i], c.conv_dim[i], (int)cin2, c.conv_kernel[i], st));
      cin2 = c.conv_dim[i];
    }
  }
  if (c.exact)
    TRY(k_posconv_prepare_t<float>(e->P + e->p_pc_g, e->P + e->p_pc_v, (float*)e->pc_wf, (float*)e->pc_wb, e->pc_norms, c.hidden_size,
                                   c.num_conv_pos_embedding_groups, c.num_conv_pos_embeddings, st));
  else {
    TRY(k_posconv_prepare_t<bf16>(e->P + e->p_pc_g, e->P + e->p_pc_v, (bf16*)e->pc_wf, (bf16*)e->pc_wb, e->pc_norms, c.hidden_size,
                                  c.num_conv_pos_embedding_groups, c.num_conv_pos_embeddings, st));
    if (e->pc_wf_frag) {
      TRY(k_posconv_frag_weights((const bf16*)e->pc_wf, e->pc_wf_frag, c.hidden_size, c.num_conv_pos_embedding_groups, c.num_conv_pos_embeddings, st));
      TRY(k_posconv_frag_weights((const bf16*)e->pc_wb, e->pc_wb_frag, c.hidden_size, c.num_conv_pos_embedding_groups, c.num_conv_pos_embeddings, st));
    }
  }
  return SSAK_OK;
}

extern "C" size_t ssak_w2v2_workspace_bytes(const ssak_w2v2* e, int B, int T, int training) {
  if (!e || B <= 0 || T <= 0) return 0;
  Plan p;
  if (make_plan(e, B, T, training, p) != SSAK_OK) return 0;
  return p.total;
}

extern "C" int ssak_w2v2_num_frames(const ssak_w2v2* e, int T) {
  if (!e) return 0;
  if (e->cfg.arch == 1) return (T >= 2 && (T & 1) == 0) ? T / 2 : 0;
  int L = T;
  for (int i = 0; i < e->cfg.num_conv_layers; ++i) {
    if (L < e->cfg.conv_kernel[i]) return 0;
    L = conv_len(L, e->cfg.conv_kernel[i], e->cfg.conv_stride[i]);
  }
  return L;
}

// logits != NULL: through final dropout + lm_head (Wav2Vec2ForCTC); hidden != NULL: the encoder's last hidden state
// (Wav2Vec2Model()[0], what the SpeechBrain recipe's wav2vec2 module returns) and no head.
template <typename AT>
static int forward_impl(ssak_w2v2* e, const float* input_values, const int32_t* lens, int B, int T, const uint8_t* spec_mask,
                        const uint8_t* layer_keep /*host*/, uint64_t seed, int training, float* logits, bf16* hidden,
                        int32_t* frame_lens, void* workspace, size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(e && input_values && (logits || hidden) && workspace, "w2v2_forward: null pointer");
  SSAK_REQUIRE(e->P && e->W, "w2v2_forward: bind + sync_weights first");
  SSAK_REQUIRE(B > 0 && T > 0, "w2v2_forward: bad shape B=%d T=%d", B, T);
  SSAK_REQUIRE(((uintptr_t)workspace & 255) == 0, "w2v2_forward: workspace must be 256-byte aligned");
  constexpr bool EXACT = sizeof(AT) == 4;  // fp32-exact verification mode (ssak_w2v2_config.exact)
  EngineCall engine_call(e);
  Plan& p = e->plan;
  e->have_fwd = false;
  TRY(make_plan(e, B, T, training, p));
  SSAK_REQUIRE(workspace_bytes >= p.total, "w2v2_forward: workspace too small (%zu < %zu)", workspace_bytes, p.total);
  const ssak_w2v2_config& c = e->cfg;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  auto BF = [&](size_t off) { return (AT*)(ws + off); };
  auto FP = [&](size_t off) { return (float*)(ws + off); };
  const int H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size, nh = c.num_heads, hd = H / nh;
  const bool whisper = c.arch == 1;
  const int nc = whisper ? 1 : c.num_conv_layers, C = whisper ? 8 : c.conv_dim[nc - 1], K = whisper ? 8 : c.num_conv_pos_embeddings;
  const int G = whisper ? 1 : c.num_conv_pos_embedding_groups, cg = H / G;
  const int F = p.F, M = p.M, Fp = p.Fp;
  const float* P = e->P;
  const AT* W = EXACT ? (const AT*)e->P : (const AT*)e->W;  // exact mode: products read the fp32 master weights
  e->seed = seed;
  e->spec_mask = spec_mask;
  e->lens = lens;
  e->last_input = input_values;
  const bool tr = training != 0;
  auto DS = [&](float prob, uint32_t stream_id) {
    DropSpec d;
    d.seed = seed;
    d.stream = stream_id;
    d.p = tr ? prob : 0.f;
    return d;
  };
  const DropSpec none;
  // the optimizer may still be updating params / shadow on its own stream (ssak_w2v2_set_param_event)
  bool waited = false;
  auto wait_params = [&]() -> int {
    if (waited || !e->params_ready) return SSAK_OK;
    waited = true;
    if (e->stall_begin) SSAK_HIP(hipEventRecord(e->stall_begin, st));
    SSAK_HIP(hipStreamWaitEvent(st, e->params_ready, 0));
    if (e->stall_end) SSAK_HIP(hipEventRecord(e->stall_end, st));
    return SSAK_OK;
  };
  // conv1 of the Whisper front end and an unfrozen feature encoder are trainable: wait before anything runs
  if (whisper || !c.freeze_feature_encoder) TRY(wait_params());

  int32_t* flens = nullptr;
  const bool stable = whisper || c.do_stable_layer_norm != 0;
  if (whisper) {
   {
    // ---- a14 front end: mel [B, NM, Tin] -> channels-last padded -> conv1+GELU -> conv2(s2)+GELU -> + positions
    const int NM = c.num_mel_bins, Tin = p.Tin, RS1 = p.RS1;
    SSAK_REQUIRE(!lens, "whisper: fixed-length windows, no attention mask (modeling_whisper.py:605-607)");
    SSAK_HIP(hipMemsetAsync(ws + p.melcl, 0, ((size_t)B * RS1 + 8) * NM * sizeof(AT), st));
    SSAK_HIP(hipMemsetAsync(ws + p.h1pad, 0, ((size_t)B * RS1 + 8) * H * sizeof(AT), st));
    TRY(k_mel_to_cl_t<AT>(input_values, BF(p.melcl), B, NM, Tin, RS1, 1, st));
    TRY(GemmX<EXACT>(Tin, H, 3 * NM).a(BF(p.melcl), NM).b(e->conv_w[1], 3 * NM).c(BF(p.h1pad) + H, H)
            .batch(B, 1, (long)RS1 * NM, 0, 0, 0, (long)RS1 * H, 0).with_bias(P + e->p_c1b)
            .epi(SSAK_EPI_GELU, nullptr, BF(p.pre1) + H).run(st));
    TRY(GemmX<EXACT>(F, H, 3 * H).a(BF(p.h1pad), 2 * H).b(e->conv_w[2], 3 * H).c(BF(p.we), H)
            .batch(B, 1, (long)RS1 * H, 0, 0, 0, (long)F * H, 0).with_bias(P + e->p_c2b)
            .epi(SSAK_EPI_GELU, nullptr, BF(p.wpre2)).run(st));
    TRY(k_add_rowvec_t<AT>(BF(p.we), W + e->p_pos, BF(p.h1), B, F, H, st));
    // residual stream r = dropout(conv + pos); x0 = self_attn_layer_norm of layer 0
    TRY(k_layernorm_fwd_t<AT>(BF(p.h1), nullptr, P + e->lp[0].ln1w, P + e->lp[0].ln1b, BF(p.h1), BF(p.x[0]), FP(p.stE),
                        FP(p.stE) + M, M, H, c.layer_norm_eps, none, none, st, DS(c.hidden_dropout, DS_ENCIN)));
   }
  } else {
  // frame lengths (attention / CTC masks) from sample lengths: integer floor-div chain (modeling_wav2vec2.py:997-1016)
  if (lens) {
    flens = (int32_t*)(ws + p.flens);
    ConvChain cc;
    cc.n = nc;
    for (int i = 0; i < nc; ++i) {
      cc.k[i] = c.conv_kernel[i];
      cc.s[i] = c.conv_stride[i];
    }
    frame_lens_kernel<<<ssak_cdiv(B, 64), 64, 0, st>>>(lens, B, T, cc, flens);
    SSAK_LAUNCH_CHECK();
    if (frame_lens) SSAK_HIP(hipMemcpyAsync(frame_lens, flens, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  }

  // ---- a3: feature encoder (frozen: forward only)
  const bool ln_fe = c.feat_extract_norm == 1;
  SSAK_REQUIRE(!(e->raw_input && p.fe_train), "w2v2_forward: raw input (normalisation folded into conv0's GroupNorm) needs the frozen feature encoder");
  if (!ln_fe) {
    TRY(k_conv0_gn_gelu_t<AT>(input_values, P + e->p_conv_w[0], P + e->p_cln_w[0], P + e->p_cln_b[0],
                        p.fe_train ? BF(p.act[0]) : BF(p.bufA), (double*)(ws + p.stats0), B, T, p.Tl[0], c.conv_dim[0],
                        c.conv_kernel[0], c.conv_stride[0], st, e->raw_input != 0));
  } else {
    // layer-norm variant (XLSR, modeling_wav2vec2.py:275-299): conv + bias -> LayerNorm over channels -> GELU
    // (--no_freeze keeps the pre-LayerNorm conv output and the statistics of every layer for the backward)
    AT* pre0 = p.fe_train ? BF(p.cpre[0]) : BF(p.bufA);
    TRY(k_conv0_bias_t<AT>(input_values, P + e->p_conv_w[0], c.conv_bias ? P + e->p_conv_b[0] : nullptr, pre0, B, T,
                     p.Tl[0], c.conv_dim[0], c.conv_kernel[0], c.conv_stride[0], st));
    TRY(k_layernorm_fwd_t<AT>(pre0, nullptr, P + e->p_cln_w[0], P + e->p_cln_b[0], nullptr, p.fe_train ? BF(p.act[0]) : BF(p.bufA),
                        p.fe_train ? FP(p.fe_st[0]) : nullptr, p.fe_train ? FP(p.fe_st[0]) + (size_t)B * p.Tl[0] : nullptr,
                        B * p.Tl[0], c.conv_dim[0], 1e-5f, none, none, st, none, true));
  }
  {
    AT* src = p.fe_train ? BF(p.act[0]) : BF(p.bufA);
    for (int i = 1; i < nc; ++i) {
      AT* dst = (i == nc - 1) ? BF(p.feat) : (p.fe_train ? BF(p.act[i]) : ((i & 1) ? BF(p.bufB) : BF(p.bufA)));
      const int Ci = c.conv_dim[i - 1], Co = c.conv_dim[i], k = c.conv_kernel[i], s = c.conv_stride[i];
      AT* conv_out = (ln_fe && p.fe_train) ? BF(p.cpre[i]) : dst;  // layer-norm variant, training: pre-LN values are kept
      GemmX<EXACT> g(p.Tl[i], Co, k * Ci);
      g.a(src, (long)s * Ci).b(e->conv_w[i], (long)k * Ci).c(conv_out, Co)
          .batch(B, 1, (long)p.Tl[i - 1] * Ci, 0, 0, 0, (long)p.Tl[i] * Co, 0);
      if (c.conv_bias) g.with_bias(P + e->p_conv_b[i]);
      if (!ln_fe) g.epi(SSAK_EPI_GELU, nullptr, p.fe_train ? BF(p.cpre[i]) : nullptr);
      TRY(g.run(st));
      if (ln_fe)
        TRY(k_layernorm_fwd_t<AT>(conv_out, nullptr, P + e->p_cln_w[i], P + e->p_cln_b[i], nullptr, dst,
                            p.fe_train ? FP(p.fe_st[i]) : nullptr, p.fe_train ? FP(p.fe_st[i]) + (size_t)B * p.Tl[i] : nullptr,
                            B * p.Tl[i], Co, 1e-5f, none, none, st, none, true));
      src = dst;
    }
  }
  // ---- a4: feature projection  LN -> Linear (+ feat_proj_dropout): the first read of trainable parameters
  TRY(wait_params());
  TRY(k_layernorm_fwd_t<AT>(BF(p.feat), nullptr, P + e->p_fpln_w, P + e->p_fpln_b, nullptr, BF(p.ln0), FP(p.st0),
                      FP(p.st0) + M, M, C, c.layer_norm_eps, none, none, st));
  TRY(GemmX<EXACT>(M, H, C).a(BF(p.ln0), C).b(W + e->p_fp_w, C).c(BF(p.h0), H).with_bias(P + e->p_fp_b)
          .run(st));
    if (tr && c.feat_proj_dropout > 0.f)  // same row kernel (and mask generator) as its replay in the backward
      TRY(k_layernorm_fwd_t<AT>(BF(p.h0), nullptr, nullptr, nullptr, BF(p.h0), nullptr, nullptr, nullptr, M, H, 0.f,
                          DS(c.feat_proj_dropout, DS_FEATPROJ), none, st));
  // ---- a5: SpecAugment scatter + zeroing of padded frames
  TRY(k_specaug_fwd_t<AT>(BF(p.h0), spec_mask, flens, P + e->p_mse, B, F, H, st));
  // ---- a6: positional conv (grouped, weight-normed) + GELU, residual, LayerNorm, dropout
  TRY(k_posconv_pack_t<AT>(BF(p.h0), BF(p.pgx), B, F, H, G, K, st));
  bool pc_direct = false;
  if constexpr (!EXACT) pc_direct = e->posconv_direct && e->pc_wf_frag != nullptr;
  if (pc_direct) {
    // direct convolution: the 512 + 127 input rows of a workgroup's frames stay in LDS (posconv.hip)
    if constexpr (!EXACT)
      TRY(k_posconv_direct(BF(p.pgx), p.pg_rows, 0, e->pc_wf_frag, P + e->p_pc_b, BF(p.pc), BF(p.pc_pre), B, F, H, G, K, 1, st));
  } else {
    TRY(GemmX<EXACT>(F, cg, K * cg)
            .a(BF(p.pgx), cg)
            .b(e->pc_wf, (long)K * cg)
            .c(BF(p.pc), H)
            .batch(B, G, (long)(F + K) * cg, p.pg_rows * cg, 0, (long)cg * K * cg, (long)F * H, cg)
            .with_bias(P + e->p_pc_b, cg)
            .epi(SSAK_EPI_GELU, nullptr, BF(p.pc_pre))
            .run(st));
  }
  if (!stable) {
    // post-LN (base): x0 = dropout(LN(h0 + pos))                                       (modeling_wav2vec2.py:694-697)
    TRY(k_layernorm_fwd_t<AT>(BF(p.pc), BF(p.h0), P + e->p_eln_w, P + e->p_eln_b, BF(p.h1), BF(p.x[0]), FP(p.stE), FP(p.stE) + M,
                        M, H, c.layer_norm_eps, none, DS(c.hidden_dropout, DS_ENCIN), st));
  } else {
    // stable-LN (XLSR): residual stream r = dropout(h0 + pos); x0 = LN1 of layer 0 applied to r   (:763-771, :631-640)
    TRY(k_layernorm_fwd_t<AT>(BF(p.pc), BF(p.h0), P + e->lp[0].ln1w, P + e->lp[0].ln1b, BF(p.h1), BF(p.x[0]), FP(p.stE),
                        FP(p.stE) + M, M, H, c.layer_norm_eps, none, none, st, DS(c.hidden_dropout, DS_ENCIN)));
  }
  }
  // ---- a7: encoder layers with LayerDrop: post-LN (base, :591-608) or pre-LN "stable layer norm" (XLSR, :631-654)
  e->keep.assign(c.num_layers, 1);
  e->wfrag_valid = false;
  if constexpr (!EXACT) {
    // fragment-ordered weights of the layers that will run, for the products that take the B-direct GEMM form (the optimizer's
    // event has been waited for above: the shadow is final)
    if (tr && e->fragment_weights) TRY(refresh_weight_fragments(e, M, layer_keep, st));
  }
  e->wtr_valid = false;
  if constexpr (!EXACT) {
    // transposed weights for the backward's input-gradient products (the four-wave GEMM wants both operands K-contiguous)
    if (tr && e->transposed_weights && c.hidden_size % 256 == 0 && c.intermediate_size % 256 == 0) TRY(refresh_weight_transposes(e, layer_keep, st));
  }
  e->hres.assign(c.num_layers + 1, p.h1);
  e->xin[0] = p.x[0];
  const float scale = 1.f / sqrtf((float)hd);
  for (int l = 0; l < c.num_layers; ++l) {
    const LayerP& L = e->lp[l];
    const LayerBuf& lb = p.lb[l];
    float* stl = FP(lb.st);
    // LayerNorm that produces the next layer's input in stable mode (next layer's LN1, or the encoder's final LN)
    const float* nxt_w = P + ((l + 1 < c.num_layers) ? e->lp[l + 1].ln1w : e->p_eln_w);
    const float* nxt_b = P + ((l + 1 < c.num_layers) ? e->lp[l + 1].ln1b : e->p_eln_b);
    if (tr && layer_keep && !layer_keep[l]) {
      e->keep[l] = 0;
      if (!stable) {
        // skipped layer: output = input (modeling_wav2vec2.py:701-712) -- the next layer reads this layer's input where it lies
        // (it used to be a 24.5 MB device copy per dropped layer)
        e->xin[l + 1] = e->xin[l];
      } else {
        // the residual stream passes through; the next LayerNorm still has to be applied to it
        TRY(k_layernorm_fwd_t<AT>(BF(e->hres[l]), nullptr, nxt_w, nxt_b, nullptr, BF(p.x[l + 1]), stl + 2 * M, stl + 3 * M, M, H,
                            c.layer_norm_eps, none, none, st));
        e->hres[l + 1] = e->hres[l];
        e->xin[l + 1] = p.x[l + 1];
      }
      continue;
    }
    const AT* x = BF(e->xin[l]);
    e->xin[l + 1] = p.x[l + 1];
    AT* qkv = BF(lb.qkv);
    TRY(GemmX<EXACT>(M, 3 * H, H).a(x, H).b(W + L.wqkv, H).bfrag(e->frag(l, 0)).c(qkv, 3 * H).with_bias(P + L.bqkv).run(st));
    if (p.fused_attn) {
      // scores never leave the MFMA accumulators (attention.hip); only ctx and the per-row log-sum-exp are written
      if constexpr (!EXACT)
        TRY(k_attention_fwd(qkv, BF(lb.ctx), FP(lb.lse), flens, B, F, nh, H, DS(c.attention_dropout, ds_attn(l)), st));
    } else {
      TRY(GemmX<EXACT>(F, F, hd).a(qkv, 3 * H).b(qkv + H, 3 * H).c(BF(p.S), Fp).alpha(scale)
              .batch(B, nh, (long)F * 3 * H, hd, (long)F * 3 * H, hd, (long)nh * F * Fp, (long)F * Fp).run(st));
      TRY(k_softmax_fwd_t<AT>(BF(p.S), BF(lb.P), lb.Pd != lb.P ? BF(lb.Pd) : nullptr, flens, B * nh * F, F, Fp, nh * F,
                        DS(c.attention_dropout, ds_attn(l)), st));
      TRY(GemmX<EXACT>(F, hd, F).a(BF(lb.Pd), Fp).b(qkv + 2 * H, 3 * H, true).c(BF(lb.ctx), H)
              .batch(B, nh, (long)nh * F * Fp, (long)F * Fp, (long)F * 3 * H, hd, (long)F * H, hd).run(st));
    }
    TRY(GemmX<EXACT>(M, H, H).a(BF(lb.ctx), H).b(W + L.wo, H).bfrag(e->frag(l, 1)).c(BF(p.tmpH), H).with_bias(P + L.bo).run(st));
    if (!stable) {
      TRY(k_layernorm_fwd_t<AT>(BF(p.tmpH), x, P + L.ln1w, P + L.ln1b, BF(lb.r1), BF(lb.x1), stl, stl + M, M, H,
                          c.layer_norm_eps, DS(c.hidden_dropout, ds_hid1(l)), none, st));
    } else {
      // r1 = r + drop(attn);  x1 = final_layer_norm(r1)
      TRY(k_layernorm_fwd_t<AT>(BF(p.tmpH), BF(e->hres[l]), P + L.ln2w, P + L.ln2b, BF(lb.r1), BF(lb.x1), stl, stl + M, M, H,
                          c.layer_norm_eps, DS(c.hidden_dropout, ds_hid1(l)), none, st));
    }
    TRY(GemmX<EXACT>(M, I, H).a(BF(lb.x1), H).b(W + L.w1, H).c(BF(lb.f1), I).with_bias(P + L.b1)
            .epi(tr ? SSAK_EPI_GELU_SAVE_GRAD : SSAK_EPI_GELU, nullptr, tr ? BF(lb.f1pre) : nullptr)  // f1pre := 8-bit codes of gelu'(pre) * mask (float values incl. 1 / (1 - p) in the exact mode)
            .drop(tr ? c.activation_dropout : 0.f, ds_act(l), seed).run(st));
    TRY(GemmX<EXACT>(M, H, I).a(BF(lb.f1), I).b(W + L.w2, I).bfrag(e->frag(l, 2)).c(BF(p.tmpH), H).with_bias(P + L.b2).run(st));
    if (!stable) {
      TRY(k_layernorm_fwd_t<AT>(BF(p.tmpH), BF(lb.x1), P + L.ln2w, P + L.ln2b, BF(lb.r2), BF(p.x[l + 1]), stl + 2 * M, stl + 3 * M,
                          M, H, c.layer_norm_eps, DS(c.hidden_dropout, ds_hid2(l)), none, st));
    } else {
      // r2 = r1 + drop(ffn);  x[l+1] = (next layer's LN1 | encoder LN)(r2)
      TRY(k_layernorm_fwd_t<AT>(BF(p.tmpH), BF(lb.r1), nxt_w, nxt_b, BF(lb.r2), BF(p.x[l + 1]), stl + 2 * M, stl + 3 * M, M, H,
                          c.layer_norm_eps, DS(c.hidden_dropout, ds_hid2(l)), none, st));
      e->hres[l + 1] = lb.r2;
    }
  }
  // ---- a8: final dropout + lm_head -> fp32 logits
  const AT* xl = BF(e->xin[c.num_layers]);
  if (hidden) {
    SSAK_REQUIRE(!EXACT, "w2v2_forward_hidden: the hidden-state interface is bf16 (not built for the fp32-exact mode)");
    SSAK_HIP(hipMemcpyAsync(hidden, xl, (size_t)M * H * sizeof(bf16), hipMemcpyDeviceToDevice, st));
    e->have_fwd = tr;
    e->fwd_hidden = true;
    return SSAK_OK;
  }
  e->fwd_hidden = false;
  if (tr && c.final_dropout > 0.f) {
    TRY(k_layernorm_fwd_t<AT>(xl, nullptr, nullptr, nullptr, BF(p.xf), nullptr, nullptr, nullptr, M, H, 0.f,
                        DS(c.final_dropout, DS_FINAL), none, st));
    xl = BF(p.xf);
  }
  TRY(GemmX<EXACT>(M, V, H).a(xl, H).b(W + e->p_lm_w, H).c(logits, V, true).with_bias(P + e->p_lm_b).run(st));
  e->have_fwd = tr;
  return SSAK_OK;
}

extern "C" int ssak_w2v2_forward(ssak_w2v2* e, const float* input_values, const int32_t* lens, int B, int T,
                                 const uint8_t* spec_mask, const uint8_t* layer_keep /*host*/, uint64_t seed,
                                 int training, float* logits, int32_t* frame_lens, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(e && logits, "w2v2_forward: null pointer");
  if (e->cfg.exact)
    return forward_impl<float>(e, input_values, lens, B, T, spec_mask, layer_keep, seed, training, logits, nullptr, frame_lens, workspace,
                               workspace_bytes, stream);
  return forward_impl<bf16>(e, input_values, lens, B, T, spec_mask, layer_keep, seed, training, logits, nullptr, frame_lens, workspace,
                      workspace_bytes, stream);
}

extern "C" int ssak_w2v2_forward_hidden(ssak_w2v2* e, const float* input_values, const int32_t* lens, int B, int T,
                                        const uint8_t* spec_mask, const uint8_t* layer_keep /*host*/, uint64_t seed,
                                        int training, void* hidden_bf16, int32_t* frame_lens, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(e && hidden_bf16, "w2v2_forward_hidden: null pointer");
  SSAK_REQUIRE(!e->cfg.exact, "w2v2_forward_hidden: not built for the fp32-exact mode");
  return forward_impl<bf16>(e, input_values, lens, B, T, spec_mask, layer_keep, seed, training, nullptr, (bf16*)hidden_bf16, frame_lens,
                      workspace, workspace_bytes, stream);
}

// dlogits (after ssak_w2v2_forward) or dhidden (after ssak_w2v2_forward_hidden): exactly one is non-null
template <typename AT>
static int backward_impl(ssak_w2v2* e, const float* dlogits, const bf16* dhidden, void* workspace, size_t workspace_bytes,
                         void* stream) {
  SSAK_REQUIRE(e && (dlogits || dhidden) && workspace, "w2v2_backward: null pointer");
  if (!e->have_fwd || e->fwd_hidden != (dhidden != nullptr)) {
    ssak_set_error("w2v2_backward: no matching training-mode forward to differentiate");
    return SSAK_ERR_STATE;
  }
  constexpr bool EXACT = sizeof(AT) == 4;
  EngineCall engine_call(e);
  SSAK_REQUIRE(e->G, "w2v2_backward: no gradient buffer bound");
  Plan& p = e->plan;
  SSAK_REQUIRE(workspace_bytes >= p.total, "w2v2_backward: workspace too small");
  const ssak_w2v2_config& c = e->cfg;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  auto BF = [&](size_t off) { return (AT*)(ws + off); };
  auto FP = [&](size_t off) { return (float*)(ws + off); };
  const int H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size, nh = c.num_heads, hd = H / nh;
  const bool whisper = c.arch == 1;
  const int nc = whisper ? 1 : c.num_conv_layers, C = whisper ? 8 : c.conv_dim[nc - 1], K = whisper ? 8 : c.num_conv_pos_embeddings;
  const int G = whisper ? 1 : c.num_conv_pos_embedding_groups, cg = H / G;
  const int B = p.B, F = p.F, M = p.M, Fp = p.Fp;
  const float* P = e->P;
  float* Gd = e->G;
  const AT* W = EXACT ? (const AT*)e->P : (const AT*)e->W;  // exact mode: products read the fp32 master weights
  const uint64_t seed = e->seed;
  const int32_t* flens = e->lens ? (const int32_t*)(ws + p.flens) : nullptr;
  auto DS = [&](float prob, uint32_t stream_id) {
    DropSpec d;
    d.seed = seed;
    d.stream = stream_id;
    d.p = prob;
    return d;
  };
  const DropSpec none;
  void* slab = ws + p.slab;
  const size_t cs_floats = (size_t)LN_BWD_BLOCKS * 3 * std::max(H, C);  // p.lnpart doubles as the column-sum scratch
  const float scale = 1.f / sqrtf((float)hd);

  const long n_grad = (e->cfg.arch == 1 || e->cfg.freeze_feature_encoder) ? e->n_train : e->n_total;
  {
    // Zero what is ACCUMULATED into (bias / LayerNorm / embedding gradients: everything outside the layers' weight matrices)
    // and the matrices of dropped layers; the four matrices of a kept layer are written whole by its weight-gradient products
    // (94 % of the buffer: 40 us of memset per step).  Falls back to the whole buffer if the layers are not laid out back to back.
    const long span = (e->lp[0].w2 + (long)c.hidden_size * c.intermediate_size) - e->lp[0].wqkv;
    bool packed = true;
    for (int l = 1; l < c.num_layers; ++l) packed &= e->lp[l].wqkv == e->lp[l - 1].wqkv + span;
    const long w0 = e->lp[0].wqkv, w1 = e->lp[c.num_layers - 1].wqkv + span;
    if (!packed || w1 > n_grad) {
      SSAK_HIP(hipMemsetAsync(Gd, 0, (size_t)n_grad * sizeof(float), st));
    } else {
      if (w0 > 0) SSAK_HIP(hipMemsetAsync(Gd, 0, (size_t)w0 * sizeof(float), st));
      for (int l = 0; l < c.num_layers; ++l)
        if (!e->keep[l]) SSAK_HIP(hipMemsetAsync(Gd + e->lp[l].wqkv, 0, (size_t)span * sizeof(float), st));
      if (n_grad > w1) SSAK_HIP(hipMemsetAsync(Gd + w1, 0, (size_t)(n_grad - w1) * sizeof(float), st));
    }
  }
  // ---- lm_head (its gradient stays zero when the backward starts from the hidden state)
  AT* dlog = BF(p.dlog);
  if (dlogits) {
    if constexpr (EXACT)
      dlog = const_cast<float*>(dlogits);  // the CTC gradient is consumed in fp32 as it is
    else
      TRY(k_cast_f32_bf16(dlogits, dlog, (long)M * V, st));
    // (the lm_head weight gradient -- V x H, K = M: three tiles -- rides in the first grouped weight-gradient launch below instead of
    // running alone as a split-K product with its slab reduction: queued once the queue exists)
    TRY(k_colsum_t<AT>(dlog, V, M, V, Gd + e->p_lm_b, st, FP(p.lnpart), cs_floats));
  }
  auto announce = [&](long off, long cnt) {
    if (e->on_ready && cnt > 0) e->on_ready(off, cnt, e->on_ready_user);
  };
  const long layer_span = (e->lp[0].w2 + (long)H * I) - e->lp[0].wqkv;  // wqkv|wo|w1|w2 of one layer are contiguous
  if (!dlogits) announce(e->p_lm_w, (long)V * H);  // (hidden-state interface: the head's gradient stays zero, its range is complete)
  AT* gA = BF(p.dA);  // gradient w.r.t. the current layer output = gA (+ gB)
  AT* gB = nullptr;
  if (dhidden)
    SSAK_HIP(hipMemcpyAsync(gA, dhidden, (size_t)M * H * sizeof(bf16), hipMemcpyDeviceToDevice, st));
  else {
    // dx = (dlogits W) * mask / (1 - p): the final-dropout mask is replayed in this product's epilogue ((row, column) of the [M, H]
    // output = the site's indices; through round 5 a pass of its own over gA, 14 us per step)
    GemmX<EXACT> g(M, H, V);
    g.a(dlog, V).b(W + e->p_lm_w, H, true).c(gA, H);
    if (c.final_dropout > 0.f) g.drop(c.final_dropout, DS_FINAL, seed);
    TRY(g.run(st));
  }
  // ---- encoder layers, last to first.  gA (+gB) = gradient w.r.t. x[l+1], the layer output (post-LN) or the
  // normalised input of the next layer (stable-LN); Gres = gradient of the residual stream (stable-LN only).
  const bool stable = whisper || c.do_stable_layer_norm != 0;
  const AT* Gres = nullptr;
  auto free_buf = [&](const AT* u1, const AT* u2, const AT* u3) {
    AT* cand[3] = {BF(p.dC), BF(p.dA), BF(p.scratchH)};
    for (AT* cnd : cand)
      if (cnd != u1 && cnd != u2 && cnd != u3) return cnd;
    return (AT*)nullptr;
  };
  WgradQueue wq;
  // second stages of the layers' column reductions: queued, launched together at every weight-gradient flush
  ReduceSink sink;
  struct SinkGuard {
    SinkGuard(ReduceSink* s) { g_reduce_sink = s; }
    ~SinkGuard() { g_reduce_sink = nullptr; }
  };
  size_t red_used = 0;
  auto red_take = [&](size_t nfloats) -> float* {  // a partial-sum region that stays untouched until the next flush
    nfloats = (nfloats + 63) & ~(size_t)63;
    if (red_used + nfloats > p.redring_floats) return nullptr;
    float* r = FP(p.redring) + red_used;
    red_used += nfloats;
    return r;
  };
  auto red_get = [&](size_t nfloats, float** out) -> int {
    float* r = red_take(nfloats);
    if (!r) {  // ring full (cannot happen with the flush cadence of two layers; kept for safety): run what is queued
      TRY(k_reduce_flush(sink, st));
      red_used = 0;
      r = red_take(nfloats);
    }
    if (!r) {
      ssak_set_error("w2v2_backward: reduction ring too small");
      return SSAK_ERR_STATE;
    }
    *out = r;
    return SSAK_OK;
  };
  const size_t ln_part_floats = (size_t)LN_BWD_BLOCKS * 3 * H;
  int kept = 0;  // layers that ran so far: selects the buffer set their weight-gradient operands live in
  auto flush_reductions = [&]() -> int {
    TRY(k_reduce_flush(sink, st));
    red_used = 0;
    return SSAK_OK;
  };
  // Weight-gradient products are packed GREEDILY into launches of at most one round of 256 x 256 tiles (a base layer has 108:
  // pairs of layers filled 216 of 256 workgroups and LayerDrop left an odd layer over in half of the steps, which then ran as
  // four split-K launches): a product that does not fit any more starts the next launch, so launches hold ~2.4 layers and a
  // layer's products may go out in two launches.  Its gradient range is announced when the last of them has been launched.
  auto flush_gemms = [&]() -> int {
    TRY(wq.flush(st, slab, p.slab_bytes));
    int done = 0;
    while (done < wq.n_ann && wq.ann_last[done] < wq.launched) {
      announce(wq.ann_off[done], wq.ann_cnt[done]);
      ++done;
    }
    for (int i = done; i < wq.n_ann; ++i) {
      wq.ann_off[i - done] = wq.ann_off[i];
      wq.ann_cnt[i - done] = wq.ann_cnt[i];
      wq.ann_last[i - done] = wq.ann_last[i];
    }
    wq.n_ann -= done;
    return SSAK_OK;
  };
  auto wq_push = [&](const Gemm& g, int buffer_set) -> int {
    const bool full = !wgrad_all() && wq.tiles + WgradQueue::tiles_of(g.d) > 256;  // (SSAK_WGRAD_ALL: multi-round launches on purpose)
    if (wq.n > 0 && (full || wq.n == WgradQueue::CAP)) TRY(flush_gemms());
    wq.push(g, buffer_set);
    return SSAK_OK;
  };
  auto flush_wgrads = [&]() -> int {
    TRY(flush_reductions());
    TRY(flush_gemms());
    return SSAK_OK;
  };
  SinkGuard sink_guard(&sink);
  if (dlogits) {
    // lm_head weight gradient: first in the queue, so its range is the first announced (the order every rank agrees on)
    const AT* xl = (c.final_dropout > 0.f) ? BF(p.xf) : BF(e->xin[c.num_layers]);
    TRY(wq_push(GemmX<EXACT>(V, H, M).a(dlog, V, true).b(xl, H, true).c(Gd + e->p_lm_w, H, true), -1));
    wq.ann_off[wq.n_ann] = e->p_lm_w;
    wq.ann_cnt[wq.n_ann] = (long)V * H;
    wq.ann_last[wq.n_ann++] = wq.pushed - 1;
  }
  for (int l = c.num_layers - 1; l >= 0; --l) {
    const LayerP& L = e->lp[l];
    const LayerBuf& lb = p.lb[l];
    float* stl = FP(lb.st);
    float *ln_part = nullptr, *ln_part2 = nullptr, *ffn_part = nullptr, *qkv_part = nullptr;
    TRY(red_get(ln_part_floats, &ln_part));
    const long nxt_w = (l + 1 < c.num_layers) ? e->lp[l + 1].ln1w : e->p_eln_w;
    const long nxt_b = (l + 1 < c.num_layers) ? e->lp[l + 1].ln1b : e->p_eln_b;
    if (!e->keep[l]) {
      // zeros (memset above), still part of the all-reduce; behind any kept layer whose gradients are still queued
      if (wq.n_ann > 0) {
        wq.ann_off[wq.n_ann] = L.wqkv;
        wq.ann_cnt[wq.n_ann] = layer_span;
        wq.ann_last[wq.n_ann++] = -1;
      } else {
        announce(L.wqkv, layer_span);
      }
      if (!stable) continue;  // identity layer: gradient passes through unchanged
      // x[l+1] = LN_next(r): its gradient joins the residual-stream gradient; nothing consumed x[l]
      AT* dr = free_buf(gA, gB, Gres);
      TRY(k_layernorm_bwd_t<AT>(gA, gB, BF(e->hres[l]), stl + 2 * M, stl + 3 * M, P + nxt_w, Gres, dr, nullptr, Gd + nxt_w, Gd + nxt_b,
                          ln_part, M, H, none, none, st));
      Gres = dr;
      gA = BF(p.dB);
      gB = nullptr;
      SSAK_HIP(hipMemsetAsync(gA, 0, (size_t)M * H * sizeof(AT), st));
      continue;
    }
    TRY(red_get(ln_part_floats, &ln_part2));
    const size_t ffn_part_floats = (size_t)std::max(ssak_cdiv(M, 64), 64) * I;  // (>= 64 rows: the non-fused fallback's partials)
    TRY(red_get(ffn_part_floats, &ffn_part));
    // q|k|v bias gradient: first stage inside the fused attention backward kernels when they run, else a column-sum pass over dqkv
    const size_t qkv_fused_floats = (p.fused_attn && !EXACT) ? k_attention_bwd_bias_floats(B, F, H) : 0;
    TRY(red_get(std::max((size_t)64 * 3 * H, qkv_fused_floats), &qkv_part));
    AT* dR = stable ? free_buf(gA, gB, Gres) : BF(p.dC);  // grad wrt r2
    const int set = kept % p.wgrad_sets;
    ++kept;
    if (wq.uses_set(set)) TRY(flush_gemms());  // a queued product still reads the buffers this layer is about to overwrite
    AT* dY = BF(p.dYb[set]);     // dy of the feed-forward branch (dropout mask applied): dX and dW operand
    AT* dY1 = BF(p.dY1b[set]);   // dy of the attention branch
    AT* dI = BF(p.dIb[set]);
    if (!stable) {
      // final_layer_norm backward: r2 = x1 + drop(ffn)
      TRY(k_layernorm_bwd_t<AT>(gA, gB, BF(lb.r2), stl + 2 * M, stl + 3 * M, P + L.ln2w, nullptr, dR, dY,
                          Gd + L.ln2w, Gd + L.ln2b, ln_part, M, H, DS(c.hidden_dropout, ds_hid2(l)), none, st, none, Gd + L.b2));
    } else {
      // (next LN) backward: x[l+1] = LN_next(r2), r2 = r1 + drop(ffn); the residual-stream gradient is added after it
      TRY(k_layernorm_bwd_t<AT>(gA, gB, BF(lb.r2), stl + 2 * M, stl + 3 * M, P + nxt_w, Gres, dR, dY, Gd + nxt_w,
                          Gd + nxt_b, ln_part, M, H, DS(c.hidden_dropout, ds_hid2(l)), none, st, none, Gd + L.b2));
    }
    // (the dy output is written even without hidden dropout -- a plain copy then -- so that the queued weight-gradient
    // products always read buffers of this layer's set, never the rotating residual-stream buffers)
    const AT* dy2 = dY;
    TRY(wq_push(GemmX<EXACT>(H, I, M).a(dy2, H, true).b(BF(lb.f1), I, true).c(Gd + L.w2, I, true), set));  // (b2's gradient: summed by the LN backward)
    TRY(GemmX<EXACT>(M, I, H).a(dy2, H).b_wt(EXACT ? nullptr : e->wt(l, 3), H, W + L.w2, I).c(dI, I)
            .epi(SSAK_EPI_MUL_AUX, BF(lb.f1pre))  // the forward saved the whole factor (GELU' and the dropout mask) as 8-bit codes
            .drop(c.activation_dropout, ds_act(l), seed)  // (which 1 / (1 - p) the codes decode with; no mask is drawn here)
            .colsum(Gd + L.b1).run(st, ffn_part, ffn_part_floats * sizeof(float)));  // b1's gradient = column sums of dI, taken in the epilogue
    TRY(wq_push(GemmX<EXACT>(I, H, M).a(dI, I, true).b(BF(lb.x1), H, true).c(Gd + L.w1, H, true), set));
    AT* dX = BF(p.dB);
    TRY(GemmX<EXACT>(M, H, I).a(dI, I).b_wt(EXACT ? nullptr : e->wt(l, 2), I, W + L.w1, H).bfrag(e->wt(l, 2) ? nullptr : e->frag(l, 5)).c(dX, H).run(st));
    AT* dR1;
    if (!stable) {
      // layer_norm backward: r1 = x + drop(attn_out); incoming = dR (residual of r2) + dX
      dR1 = BF(p.dA);  // gA was consumed by the final_layer_norm backward above
      TRY(k_layernorm_bwd_t<AT>(dR, dX, BF(lb.r1), stl, stl + M, P + L.ln1w, nullptr, dR1, dY1, Gd + L.ln1w,
                          Gd + L.ln1b, ln_part2, M, H, DS(c.hidden_dropout, ds_hid1(l)), none, st, none, Gd + L.bo));
    } else {
      // final_layer_norm backward: x1 = LN(r1), r1 = r + drop(attn_out); residual gradient dR is added after it
      dR1 = free_buf(dR, dX, nullptr);
      TRY(k_layernorm_bwd_t<AT>(dX, nullptr, BF(lb.r1), stl, stl + M, P + L.ln2w, dR, dR1, dY1, Gd + L.ln2w,
                          Gd + L.ln2b, ln_part2, M, H, DS(c.hidden_dropout, ds_hid1(l)), none, st, none, Gd + L.bo));
    }
    const AT* dy1 = dY1;
    TRY(wq_push(GemmX<EXACT>(H, H, M).a(dy1, H, true).b(BF(lb.ctx), H, true).c(Gd + L.wo, H, true), set));  // (bo's gradient: summed by the LN backward)
    AT* dctx = free_buf(dR1, dX, nullptr);
    TRY(GemmX<EXACT>(M, H, H).a(dy1, H).b_wt(EXACT ? nullptr : e->wt(l, 1), H, W + L.wo, H).bfrag(e->wt(l, 1) ? nullptr : e->frag(l, 4)).c(dctx, H).run(st));
    // attention backward per (utterance, head)
    AT* qkv = BF(lb.qkv);
    AT* dqkv = BF(p.dqkvb[set]);
    const long sq1 = (long)F * 3 * H, sp1 = (long)nh * F * Fp, sp2 = (long)F * Fp, sh1 = (long)F * H;
    if (p.fused_attn) {
      if constexpr (!EXACT)
        TRY(k_attention_bwd(qkv, BF(lb.ctx), FP(lb.lse), flens, dctx, FP(p.delta), dqkv, B, F, nh, H,
                            DS(c.attention_dropout, ds_attn(l)), e->attn_bwd_mode, st, qkv_fused_floats ? qkv_part : nullptr,
                            qkv_fused_floats ? Gd + L.bqkv : nullptr));
    } else {
      TRY(GemmX<EXACT>(F, hd, F).a(BF(lb.Pd), Fp, true).b(dctx, H, true).c(dqkv + 2 * H, 3 * H)
              .batch(B, nh, sp1, sp2, sh1, hd, sq1, hd).run(st));  // dV = Pd^T dctx
      TRY(GemmX<EXACT>(F, F, hd).a(dctx, H).b(qkv + 2 * H, 3 * H).c(BF(p.S), Fp)
              .batch(B, nh, sh1, hd, sq1, hd, sp1, sp2).run(st));  // dPd = dctx V^T
      TRY(k_softmax_bwd_t<AT>(BF(p.S), BF(lb.P), BF(p.dSb), B * nh * F, F, Fp, DS(c.attention_dropout, ds_attn(l)), st));
      TRY(GemmX<EXACT>(F, hd, F).a(BF(p.dSb), Fp).b(qkv + H, 3 * H, true).c(dqkv, 3 * H).alpha(scale)
              .batch(B, nh, sp1, sp2, sq1, hd, sq1, hd).run(st));  // dQ = scale dS K
      TRY(GemmX<EXACT>(F, hd, F).a(BF(p.dSb), Fp, true).b(qkv, 3 * H, true).c(dqkv + H, 3 * H).alpha(scale)
              .batch(B, nh, sp1, sp2, sq1, hd, sq1, hd).run(st));  // dK = scale dS^T Q
    }
    TRY(wq_push(GemmX<EXACT>(3 * H, H, M).a(dqkv, 3 * H, true).b(BF(e->xin[l]), H, true).c(Gd + L.wqkv, H, true), set));
    if (!qkv_fused_floats) TRY(k_colsum_t<AT>(dqkv, 3 * H, M, 3 * H, Gd + L.bqkv, st, qkv_part, (size_t)64 * 3 * H));
    TRY(GemmX<EXACT>(M, H, 3 * H).a(dqkv, 3 * H).b_wt(EXACT ? nullptr : e->wt(l, 0), 3 * H, W + L.wqkv, H).bfrag(e->wt(l, 0) ? nullptr : e->frag(l, 3)).c(dX, H).run(st));
    wq.ann_off[wq.n_ann] = L.wqkv;
    wq.ann_cnt[wq.n_ann] = layer_span;
    wq.ann_last[wq.n_ann++] = wq.pushed - 1;  // announced once the qkv product -- the layer's last -- has been launched
    ++wq.layers;
    if (wgrad_all() ? (wq.layers % p.wgrad_sets == 0) : false)
      TRY(flush_wgrads());  // (SSAK_WGRAD_ALL: one launch per p.wgrad_sets layers)
    else if ((wq.layers & 1) == 0)
      TRY(flush_reductions());  // the deferred second stages of the column reductions go out every two layers
    if (!stable) {
      // gradient w.r.t. this layer's input = dR1 (residual of r1) + dX
      gA = dR1;
      gB = dX;
    } else {
      gA = dX;  // gradient w.r.t. x[l] = LN1_l(residual stream); the stream's own gradient is dR1
      gB = nullptr;
      Gres = dR1;
    }
  }
  TRY(flush_wgrads());
  g_reduce_sink = nullptr;  // the rest of the backward launches its second stages directly
  // ---- encoder input
  AT* dh1 = free_buf(gA, gB, Gres);
  if (!stable) {
    // x0 = drop(LN(h1)), h1 = h0 + gelu(posconv(h0))
    TRY(k_layernorm_bwd_t<AT>(gA, gB, BF(p.h1), FP(p.stE), FP(p.stE) + M, P + e->p_eln_w, nullptr, dh1, nullptr, Gd + e->p_eln_w,
                        Gd + e->p_eln_b, FP(p.lnpart), M, H, none, DS(c.hidden_dropout, DS_ENCIN), st));
  } else {
    // x0 = LN1_0(r), r = drop(h0 + gelu(posconv(h0))): LN backward + residual-stream gradient, then the dropout mask
    TRY(k_layernorm_bwd_t<AT>(gA, gB, BF(p.h1), FP(p.stE), FP(p.stE) + M, P + e->lp[0].ln1w, Gres, dh1, nullptr, Gd + e->lp[0].ln1w,
                        Gd + e->lp[0].ln1b, FP(p.lnpart), M, H, none, none, st, DS(c.hidden_dropout, DS_ENCIN)));
  }
  if (whisper) {
   {
    // ---- a14 front end backward: r = dropout(gelu(conv2(gelu(conv1(mel)))) + pos); positions are fixed
    const int NM = c.num_mel_bins, Tin = p.Tin, RS1 = p.RS1, RS2 = p.RS2;
    AT* dpre2 = BF(p.dY);
    TRY(k_gelu_grad_mul_t<AT>(dh1, BF(p.wpre2), dpre2, (long)M * H, st));
    TRY(k_colsum_t<AT>(dpre2, H, M, H, Gd + e->p_c2b, st, FP(p.lnpart), cs_floats));
    TRY(k_copy_rows_padded_t<AT>(dpre2, BF(p.dpre2pad), B, F, RS2, H, st));
    // dW2[n][tap*H + c] = sum over rows kk = b*RS2 + t of dy[kk][n] * h1pad[2*kk + tap][c]   (one long-K GEMM)
    TRY(GemmX<EXACT>(H, 3 * H, B * RS2).a(BF(p.dpre2pad), H, true).b(BF(p.h1pad), 2 * H, true).c(FP(p.dwr), 3 * H, true)
            .run_wgrad(st, slab, p.slab_bytes));
    TRY(k_conv_wgrad_unrearrange(FP(p.dwr), Gd + e->p_c2w, H, H, 3, st));
    // input gradient in column form, then col2im (+ GELU' of conv1's pre-activation)
    TRY(GemmX<EXACT>(M, 3 * H, H).a(dpre2, H).b(e->conv_w[2], 3 * H, true).c(BF(p.dxcol), 3 * H).run(st));
    TRY(k_col2im_k3s2_t<AT>(BF(p.dxcol), BF(p.pre1), BF(p.dpre1pad), B, F, Tin, RS1, H, st));
    TRY(k_colsum_t<AT>(BF(p.dpre1pad), H, B * RS1, H, Gd + e->p_c1b, st, FP(p.lnpart), cs_floats));
    TRY(GemmX<EXACT>(H, 3 * NM, B * RS1).a(BF(p.dpre1pad), H, true).b(BF(p.melcl), NM, true).c(FP(p.dwr), 3 * NM, true)
            .run_wgrad(st, slab, p.slab_bytes));
    TRY(k_conv_wgrad_unrearrange(FP(p.dwr), Gd + e->p_c1w, H, NM, 3, st));
    for (int l = 0; l < c.num_layers; ++l)  // k_proj has no bias in Whisper: keep its slot out of the optimizer
      SSAK_HIP(hipMemsetAsync(Gd + e->lp[l].bqkv + H, 0, (size_t)H * sizeof(float), st));
   }
  } else {
  AT* dpre = BF(p.dY);
  TRY(k_gelu_grad_mul_t<AT>(dh1, BF(p.pc_pre), dpre, (long)M * H, st));
  TRY(k_colsum_t<AT>(dpre, H, M, H, Gd + e->p_pc_b, st, FP(p.lnpart), cs_floats));
  TRY(k_posconv_pack_t<AT>(dpre, BF(p.pgdy), B, F, H, G, K, st));
  {
    const int lead = K / 2, RS = F + K;
    // dW^T[g][tap*cg + c][n] = sum over packed rows of x[row + tap][c] * dy[row][n]: one long-K GEMM per group, with the
    // 48 output channels of a group on the N side (128x64 tiles, 75 % useful) -- on the M side they sat in 128-row tiles
    // (37 % useful) and this product was the slowest launch of the backward
    bool pcw_direct = false;
    if constexpr (!EXACT)
      pcw_direct = e->posconv_direct && e->pc_wf_frag != nullptr && k_posconv_wgrad_scratch_floats(H, G, K) * sizeof(float) <= p.slab_bytes &&
                   !SSAK_DEV_ENV("SSAK_PCW_GEMM");  // (development builds: the weight gradient alone as the Toeplitz GEMM, tools/pcw_check.py)
    if (pcw_direct) {
      // direct contraction over time: a stage of x and dy rows is written to LDS once and serves every tap (posconv.hip)
      if constexpr (!EXACT)
        TRY(k_posconv_wgrad_direct(BF(p.pgx), BF(p.pgdy), p.pg_rows, (long)B * RS, lead, FP(p.dwf), reinterpret_cast<float*>(slab), H, G, K, st));
    } else {
      TRY(GemmX<EXACT>(K * cg, cg, B * RS)
              .a(BF(p.pgx), cg, true)
              .b(BF(p.pgdy) + (long)lead * cg, cg, true)
              .c(FP(p.dwf), cg, true)
              .batch(1, G, 0, p.pg_rows * cg, 0, p.pg_rows * cg, 0, (long)cg * K * cg)
              .run(st));
    }
    TRY(k_posconv_weight_bwd(FP(p.dwf), P + e->p_pc_g, P + e->p_pc_v, e->pc_norms, Gd + e->p_pc_g, Gd + e->p_pc_v, H, G, K, st));
    // input gradient: correlation of dy with the flipped, transposed taps
    const int shift = 2 * (K / 2) - K + 1;  // 1 for even K (SamePad drops the last frame), 0 for odd
    bool pc_direct = false;
    if constexpr (!EXACT) pc_direct = e->posconv_direct && e->pc_wf_frag != nullptr;
    if (pc_direct) {
      if constexpr (!EXACT)
        TRY(k_posconv_direct(BF(p.pgdy), p.pg_rows, shift, e->pc_wb_frag, nullptr, BF(p.dB), nullptr, B, F, H, G, K, 0, st));
    } else {
      TRY(GemmX<EXACT>(F, cg, K * cg)
              .a(BF(p.pgdy) + (long)shift * cg, cg)
              .b(e->pc_wb, (long)K * cg)
              .c(BF(p.dB), H)
              .batch(B, G, (long)RS * cg, p.pg_rows * cg, 0, (long)cg * K * cg, (long)F * H, cg)
              .run(st));
    }
  }
  AT* dh0 = (dh1 == BF(p.dA)) ? BF(p.dC) : BF(p.dA);
  TRY(k_add_t<AT>(dh1, BF(p.dB), dh0, (long)M * H, st));
  TRY(k_specaug_bwd_t<AT>(dh0, e->spec_mask, flens, Gd + e->p_mse, B, F, H, st, FP(p.lnpart), cs_floats));
  // ---- feature projection
  const AT* dh0d = dh0;
  if (c.feat_proj_dropout > 0.f) {
    // replay the projection-output dropout mask on the gradient
    TRY(k_layernorm_fwd_t<AT>(dh0, nullptr, nullptr, nullptr, BF(p.scratchH), nullptr, nullptr, nullptr, M, H, 0.f,
                        DS(c.feat_proj_dropout, DS_FEATPROJ), none, st));
    dh0d = BF(p.scratchH);
  }
  TRY(GemmX<EXACT>(H, C, M).a(dh0d, H, true).b(BF(p.ln0), C, true).c(Gd + e->p_fp_w, C, true).run_wgrad(st, slab, p.slab_bytes));
  TRY(k_colsum_t<AT>(dh0d, H, M, H, Gd + e->p_fp_b, st, FP(p.lnpart), cs_floats));
  TRY(GemmX<EXACT>(M, C, H).a(dh0d, H).b(W + e->p_fp_w, C, true).c(BF(p.dln0), C).run(st));
  AT* dfeat = p.fe_train ? (((nc - 1) & 1) ? BF(p.fe_db) : BF(p.fe_da)) : BF(p.ln0);  // frozen: scratch, never read
  TRY(k_layernorm_bwd_t<AT>(BF(p.dln0), nullptr, BF(p.feat), FP(p.st0), FP(p.st0) + M, P + e->p_fpln_w, nullptr, dfeat, nullptr,
                      Gd + e->p_fpln_w, Gd + e->p_fpln_b, FP(p.lnpart), M, C, none, none, st));
  if (p.fe_train) {
    const bool ln_fe = c.feat_extract_norm == 1;
    // ---- a3 backward (--no_freeze): conv stack in reverse.  Per layer: GELU', weight gradient as per-utterance
    // K-major GEMMs on the overlapping-row operand (slabs summed in a fixed order), input gradient in column form
    // (one GEMM) + col2im; finally conv0 + GroupNorm + GELU backward by recomputation from the waveform.
    for (int i = nc - 1; i >= 1; --i) {
      const int Ci = c.conv_dim[i - 1], Co = c.conv_dim[i], k = c.conv_kernel[i], s = c.conv_stride[i];
      const int Ti = p.Tl[i], Tp = p.Tl[i - 1];
      const AT* dact = (i & 1) ? BF(p.fe_db) : BF(p.fe_da);
      AT* dprev = ((i - 1) & 1) ? BF(p.fe_db) : BF(p.fe_da);
      const AT* act_prev = BF(p.act[i - 1]);
      if (ln_fe) {
        // layer-norm variant: act = gelu(LN(conv + bias)); one kernel takes d act through GELU' and the LayerNorm
        // (recomputed from the kept pre-LN values + statistics) and sums d gamma / d beta and the conv bias gradient
        TRY(k_layernorm_bwd_t<AT>(dact, nullptr, BF(p.cpre[i]), FP(p.fe_st[i]), FP(p.fe_st[i]) + (size_t)B * Ti, P + e->p_cln_w[i], nullptr,
                            BF(p.fe_dp), BF(p.fe_dp), Gd + e->p_cln_w[i], Gd + e->p_cln_b[i], FP(p.lnpart), B * Ti, Co, none, none, st, none,
                            c.conv_bias ? Gd + e->p_conv_b[i] : nullptr, P + e->p_cln_b[i]));
      } else {
        TRY(k_gelu_grad_mul_t<AT>(dact, BF(p.cpre[i]), BF(p.fe_dp), (long)B * Ti * Co, st));
      }
      TRY(GemmX<EXACT>(Co, k * Ci, Ti).a(BF(p.fe_dp), Co, true).b(act_prev, (long)s * Ci, true).c(FP(p.fe_slab), (long)k * Ci, true)
              .batch(B, 1, (long)Ti * Co, 0, (long)Tp * Ci, 0, (long)Co * k * Ci, 0).run(st));
      TRY(k_sum_slabs(FP(p.fe_slab), B, (long)Co * k * Ci, FP(p.fe_dwr), st));
      TRY(k_conv_wgrad_unrearrange(FP(p.fe_dwr), Gd + e->p_conv_w[i], Co, Ci, k, st));
      TRY(GemmX<EXACT>(Ti, k * Ci, Co).a(BF(p.fe_dp), Co).b(e->conv_w[i], (long)k * Ci, true).c(BF(p.fe_dxcol), (long)k * Ci)
              .batch(B, 1, (long)Ti * Co, 0, 0, 0, (long)Ti * k * Ci, 0).run(st));
      TRY(k_col2im_t<AT>(BF(p.fe_dxcol), dprev, B, Tp, Ti, Ci, k, s, st));
    }
    if (ln_fe) {
      const int C0 = c.conv_dim[0], T0 = p.Tl[0];
      TRY(k_layernorm_bwd_t<AT>(BF(p.fe_da), nullptr, BF(p.cpre[0]), FP(p.fe_st[0]), FP(p.fe_st[0]) + (size_t)B * T0, P + e->p_cln_w[0], nullptr,
                          BF(p.fe_dp), BF(p.fe_dp), Gd + e->p_cln_w[0], Gd + e->p_cln_b[0], FP(p.lnpart), B * T0, C0, none, none, st, none,
                          c.conv_bias ? Gd + e->p_conv_b[0] : nullptr, P + e->p_cln_b[0]));
      TRY(k_conv0_wgrad_t<AT>(BF(p.fe_dp), e->last_input, Gd + e->p_conv_w[0], FP(p.fe_c0), B, p.T, T0, C0, c.conv_kernel[0],
                        c.conv_stride[0], st));
    } else {
      TRY(k_conv0_gn_gelu_bwd_t<AT>(e->last_input, P + e->p_conv_w[0], P + e->p_cln_w[0], P + e->p_cln_b[0], BF(p.fe_da),
                              (const double*)(ws + p.stats0), FP(p.fe_c0), Gd + e->p_conv_w[0], Gd + e->p_cln_w[0],
                              Gd + e->p_cln_b[0], B, p.T, p.Tl[0], c.conv_dim[0], st));
    }
  }
  }
  // everything else: the leading small matrices and the whole vector region (biases, LayerNorm affine)
  announce(0, e->lp[0].wqkv);
  announce(e->p_lm_w + (long)V * H, n_grad - (e->p_lm_w + (long)V * H));
  e->have_fwd = false;
  return SSAK_OK;
}

extern "C" int ssak_w2v2_backward(ssak_w2v2* e, const float* dlogits, void* workspace, size_t workspace_bytes, void* stream) {
  SSAK_REQUIRE(e && dlogits, "w2v2_backward: null pointer");
  if (e->cfg.exact) return backward_impl<float>(e, dlogits, nullptr, workspace, workspace_bytes, stream);
  return backward_impl<bf16>(e, dlogits, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int ssak_w2v2_backward_hidden(ssak_w2v2* e, const void* dhidden_bf16, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  SSAK_REQUIRE(e && dhidden_bf16, "w2v2_backward_hidden: null pointer");
  SSAK_REQUIRE(!e->cfg.exact, "w2v2_backward_hidden: not built for the fp32-exact mode");
  return backward_impl<bf16>(e, nullptr, (const bf16*)dhidden_bf16, workspace, workspace_bytes, stream);
}
