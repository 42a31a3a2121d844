"""Eager torch-CPU fp32 restatement of the Wav2Vec2-CTC module graph.  TEST INFRASTRUCTURE ONLY.

The reference (linto-ai/ssak) reaches this arithmetic through ``model(input_values,
attention_mask, labels)`` at ``ssak/train/transformers/wav2vec_train.py:387-415`` and
``ssak/infer/transformers_infer.py:235``; the arithmetic itself lives in the un-vendored,
un-pinned dependency ``transformers`` (``requirements.txt:42``; survey container: 5.15.0,
``models/wav2vec2/modeling_wav2vec2.py``).  Each function cites the lines it restates.
Checked against ``transformers.Wav2Vec2ForCTC`` by ``oracle/gen_golden.py`` (build container)
and against the committed vectors in ``tests/golden/`` by ``tests/test_oracle.py``.

Parameters are a flat ``dict[str, torch.Tensor]`` keyed by the HF ``state_dict`` names, so the
same dict loads into ``Wav2Vec2ForCTC`` for pinning and into ``ssak_amd`` for parity.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclasses.dataclass
class W2V2Config:
    """Subset of ``transformers.Wav2Vec2Config`` the hot path reads (defaults = wav2vec2-base)."""

    vocab_size: int = 32
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    conv_bias: bool = False
    feat_extract_norm: str = "group"  # "group" (base) | "layer" (XLSR)
    do_stable_layer_norm: bool = False
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
    # regularisers; values are what ssak's train script passes (wav2vec_train.py:161-165,313-325)
    attention_dropout: float = 0.1
    hidden_dropout: float = 0.05
    activation_dropout: float = 0.1
    feat_proj_dropout: float = 0.0
    final_dropout: float = 0.1
    layerdrop: float = 0.1
    mask_time_prob: float = 0.05
    mask_time_length: int = 10
    mask_time_min_masks: int = 2
    pad_token_id: int = 0
    ctc_loss_reduction: str = "mean"  # wav2vec_train.py:319
    ctc_zero_infinity: bool = True  # wav2vec_train.py:325
    initializer_range: float = 0.02

    @staticmethod
    def base(**kw) -> "W2V2Config":
        return W2V2Config(**kw)

    @staticmethod
    def xlsr_large(**kw) -> "W2V2Config":
        d = dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                 feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)
        d.update(kw)
        return W2V2Config(**d)

    @staticmethod
    def tiny(**kw) -> "W2V2Config":
        """Small config for full-tensor golden vectors (same topology as base)."""
        d = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                 conv_dim=(32,) * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
        d.update(kw)
        return W2V2Config(**d)

    def deterministic(self) -> "W2V2Config":
        """All stochastic regularisers off (the parity configuration)."""
        return dataclasses.replace(self, attention_dropout=0.0, hidden_dropout=0.0, activation_dropout=0.0,
                                   feat_proj_dropout=0.0, final_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0)

    def to_hf_kwargs(self) -> dict:
        d = dataclasses.asdict(self)
        d["conv_dim"] = list(self.conv_dim)
        d["conv_kernel"] = list(self.conv_kernel)
        d["conv_stride"] = list(self.conv_stride)
        return d


# ----------------------------------------------------------------------------- parameters
def param_shapes(cfg: W2V2Config) -> Dict[str, Tuple[int, ...]]:
    """HF state_dict names and shapes, in HF registration order (modeling_wav2vec2.py:1244-1262,1597-1625)."""
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    s: Dict[str, Tuple[int, ...]] = {"wav2vec2.masked_spec_embed": (H,)}
    fe = "wav2vec2.feature_extractor.conv_layers."
    cin = 1
    for i, (c, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):
        s[f"{fe}{i}.conv.weight"] = (c, cin, k)
        if cfg.conv_bias:
            s[f"{fe}{i}.conv.bias"] = (c,)
        if (cfg.feat_extract_norm == "group" and i == 0) or cfg.feat_extract_norm == "layer":
            s[f"{fe}{i}.layer_norm.weight"] = (c,)
            s[f"{fe}{i}.layer_norm.bias"] = (c,)
        cin = c
    C = cfg.conv_dim[-1]
    s["wav2vec2.feature_projection.layer_norm.weight"] = (C,)
    s["wav2vec2.feature_projection.layer_norm.bias"] = (C,)
    s["wav2vec2.feature_projection.projection.weight"] = (H, C)
    s["wav2vec2.feature_projection.projection.bias"] = (H,)
    pc = "wav2vec2.encoder.pos_conv_embed.conv."
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    s[pc + "bias"] = (H,)
    s[pc + "parametrizations.weight.original0"] = (1, 1, K)  # weight_g (weight_norm dim=2)
    s[pc + "parametrizations.weight.original1"] = (H, H // G, K)  # weight_v
    s["wav2vec2.encoder.layer_norm.weight"] = (H,)
    s["wav2vec2.encoder.layer_norm.bias"] = (H,)
    for l in range(cfg.num_hidden_layers):
        p = f"wav2vec2.encoder.layers.{l}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"attention.{n}.weight"] = (H, H)
            s[p + f"attention.{n}.bias"] = (H,)
        s[p + "layer_norm.weight"] = (H,)
        s[p + "layer_norm.bias"] = (H,)
        s[p + "feed_forward.intermediate_dense.weight"] = (I, H)
        s[p + "feed_forward.intermediate_dense.bias"] = (I,)
        s[p + "feed_forward.output_dense.weight"] = (H, I)
        s[p + "feed_forward.output_dense.bias"] = (H,)
        s[p + "final_layer_norm.weight"] = (H,)
        s[p + "final_layer_norm.bias"] = (H,)
    s["lm_head.weight"] = (V, H)
    s["lm_head.bias"] = (V,)
    return s


def is_feature_encoder_param(name: str) -> bool:
    """Parameters frozen by ``model.freeze_feature_encoder()`` (wav2vec_train.py:326-327)."""
    return name.startswith("wav2vec2.feature_extractor.")


def init_params(cfg: W2V2Config, seed: int = 69, generator_device: str = "cpu") -> Dict[str, torch.Tensor]:
    """Deterministic seeded initialisation (the build's own scheme, so it can be re-created on
    the GPU box without ``transformers``).  Scales follow ``Wav2Vec2PreTrainedModel._init_weights``
    (modeling_wav2vec2.py:967-995) so activations have realistic magnitudes; seed 69 is the train
    script's default (wav2vec_train.py:167)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("masked_spec_embed"):
            t = torch.rand(shape, generator=g)
        elif ".layer_norm." in name or name.endswith("layer_norm.weight") or name.endswith("layer_norm.bias"):
            # LN / GN affine: perturbed around (1, 0) so that parity tests exercise them
            base = 1.0 if name.endswith("weight") else 0.0
            t = base + 0.05 * torch.randn(shape, generator=g)
        elif name.endswith("original0"):  # weight_g := ||v|| is set below
            t = torch.zeros(shape)
        elif name.endswith("original1"):
            K = shape[2]
            cin = shape[1]
            t = torch.randn(shape, generator=g) * (2.0 * math.sqrt(1.0 / (K * cin * cfg.num_conv_pos_embedding_groups)))
        elif ".conv.weight" in name:
            fan_in = shape[1] * shape[2]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)  # kaiming_normal_
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        elif name.endswith("projection.weight"):
            k = math.sqrt(1.0 / shape[1])
            t = (torch.rand(shape, generator=g) * 2 - 1) * k
        else:  # nn.Linear
            t = torch.randn(shape, generator=g) * cfg.initializer_range
        out[name] = t.to(torch.float32).contiguous()
    v = out["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    out["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = \
        v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt().contiguous()
    return out


# ----------------------------------------------------------------------------- feature extraction (a1/a2)
def zero_mean_unit_var_norm(waves, lengths=None, padding_value: float = 0.0) -> np.ndarray:
    """``Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm`` + right zero padding to the longest
    (feature_extraction_wav2vec2.py:78-97; call sites ssak/utils/dataset.py:632,
    ssak/infer/transformers_infer.py:216).  ``waves``: list of 1-D float arrays, or [B,T] with
    ``lengths``.  Population variance, eps 1e-7, pad tail := padding_value."""
    if lengths is None:
        lengths = [len(w) for w in waves]
    T = max(len(w) for w in waves)
    out = np.full((len(waves), T), padding_value, dtype=np.float32)
    for i, (w, n) in enumerate(zip(waves, lengths)):
        x = np.asarray(w[:n], dtype=np.float32)
        out[i, :n] = (x - x.mean()) / np.sqrt(x.var() + 1e-7)
    return out


def conv_out_lengths(cfg: W2V2Config, input_lengths):
    """``_get_feat_extract_output_lengths`` (modeling_wav2vec2.py:997-1016): floor((L-k)/s)+1 chained."""
    L = np.asarray(input_lengths, dtype=np.int64)
    for k, s in zip(cfg.conv_kernel, cfg.conv_stride):
        L = np.floor_divide(L - k, s) + 1
    return L


def pad_labels(label_lists, pad_value: int = -100) -> np.ndarray:
    """``DataCollatorCTCWithPadding`` label side (wav2vec_train.py:89-100): right-pad with -100."""
    L = max((len(l) for l in label_lists), default=0)
    out = np.full((len(label_lists), L), pad_value, dtype=np.int64)
    for i, l in enumerate(label_lists):
        out[i, :len(l)] = np.asarray(l, dtype=np.int64)
    return out


# ----------------------------------------------------------------------------- model stages
def feature_encoder(p, cfg: W2V2Config, x: torch.Tensor, stages=None) -> torch.Tensor:
    """``Wav2Vec2FeatureEncoder.forward`` (modeling_wav2vec2.py:409-419): [B,T] -> [B,C,F]."""
    h = x[:, None]
    fe = "wav2vec2.feature_extractor.conv_layers."
    for i, s in enumerate(cfg.conv_stride):
        h = F.conv1d(h, p[f"{fe}{i}.conv.weight"], p.get(f"{fe}{i}.conv.bias"), stride=s)
        if cfg.feat_extract_norm == "group" and i == 0:  # :302-323, GroupNorm(C groups) = per-channel over time
            C = h.shape[1]
            h = F.group_norm(h, C, p[f"{fe}0.layer_norm.weight"], p[f"{fe}0.layer_norm.bias"], eps=1e-5)
        elif cfg.feat_extract_norm == "layer":  # :275-299
            h = F.layer_norm(h.transpose(1, 2), (h.shape[1],), p[f"{fe}{i}.layer_norm.weight"],
                             p[f"{fe}{i}.layer_norm.bias"], eps=1e-5).transpose(1, 2)
        h = F.gelu(h)
        if stages is not None:
            stages[f"conv{i}"] = h
    return h


def pos_conv_weight(p, cfg: W2V2Config) -> torch.Tensor:
    """weight_norm(dim=2) parametrisation of the positional conv (modeling_wav2vec2.py:326-358):
    w = g * v / ||v||, norm over dims (0,1) for each kernel tap."""
    pc = "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight."
    g, v = p[pc + "original0"], p[pc + "original1"]
    return g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()


def pos_conv_embed(p, cfg: W2V2Config, h: torch.Tensor) -> torch.Tensor:
    """``Wav2Vec2PositionalConvEmbedding.forward`` (:360-368) incl. SamePad (:371-379) and GELU."""
    K = cfg.num_conv_pos_embeddings
    y = F.conv1d(h.transpose(1, 2), pos_conv_weight(p, cfg), p["wav2vec2.encoder.pos_conv_embed.conv.bias"],
                 padding=K // 2, groups=cfg.num_conv_pos_embedding_groups)
    if K % 2 == 0:
        y = y[:, :, :-1]
    return F.gelu(y).transpose(1, 2)


class HashDropout:
    """The engine's counter-hash dropout masks (oracle/dropout_hash.py) for one forward: ``drop(x, p, site)`` replaces
    ``F.dropout(x, p, training=True)`` with mask(seed, site) * 1 / (1 - p) -- torch's own scale, so that a site whose kernel used
    another factor shows up.  ``attention=True``: x is [B, nh, F, F] attention probabilities (the per-row hash of attention.hip)."""

    def __init__(self, seed: int):
        self.seed = int(seed)
        self.log = []  # (site, shape, p) in call order

    def __call__(self, x: torch.Tensor, p: float, site: int, attention: bool = False) -> torch.Tensor:
        from . import dropout_hash as DH
        self.log.append((site, tuple(x.shape), p))
        if p <= 0:
            return x
        if attention:
            B, nh, Fq, Fk = x.shape
            keep = DH.attention_keep_mask(self.seed, site, B, nh, Fq, p, Fk)
        else:
            keep = DH.keep_mask(self.seed, site, tuple(x.shape), p)
        return x * (torch.from_numpy(keep).to(x.dtype) / (1.0 - p))


def _drop(x, p, train, drop, site, attention=False):
    """F.dropout with torch's generator (drop is None) or with the engine's hash masks (drop = HashDropout)."""
    if drop is None or not train:
        return F.dropout(x, p=p, training=train)
    return drop(x, p, site, attention)


def _ln(p, prefix, x, eps):
    return F.layer_norm(x, (x.shape[-1],), p[prefix + ".weight"], p[prefix + ".bias"], eps=eps)


def attention(p, cfg: W2V2Config, prefix: str, h: torch.Tensor, key_mask, train: bool, drop=None, l: int = 0) -> torch.Tensor:
    """``Wav2Vec2Attention.forward`` + ``eager_attention_forward`` (:438-463,:500-548)."""
    B, T, H = h.shape
    nh = cfg.num_attention_heads
    hd = H // nh
    q = F.linear(h, p[prefix + "q_proj.weight"], p[prefix + "q_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
    k = F.linear(h, p[prefix + "k_proj.weight"], p[prefix + "k_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
    v = F.linear(h, p[prefix + "v_proj.weight"], p[prefix + "v_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
    if key_mask is not None:  # padded KEYS get -inf; padded query rows are left alone (create_bidirectional_mask)
        s = s.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    a = F.softmax(s, dim=-1)
    a = _drop(a, cfg.attention_dropout, train, drop, 16 + 4 * l, attention=True)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, H)
    return F.linear(o, p[prefix + "out_proj.weight"], p[prefix + "out_proj.bias"])


def feed_forward(p, cfg: W2V2Config, prefix: str, h: torch.Tensor, train: bool, drop=None, l: int = 0) -> torch.Tensor:
    """``Wav2Vec2FeedForward.forward`` (:565-572)."""
    x = F.gelu(F.linear(h, p[prefix + "intermediate_dense.weight"], p[prefix + "intermediate_dense.bias"]))
    x = _drop(x, cfg.activation_dropout, train, drop, 16 + 4 * l + 2)
    x = F.linear(x, p[prefix + "output_dense.weight"], p[prefix + "output_dense.bias"])
    return _drop(x, cfg.hidden_dropout, train, drop, 16 + 4 * l + 3)


def encoder_layer(p, cfg: W2V2Config, l: int, h: torch.Tensor, key_mask, train: bool, drop=None) -> torch.Tensor:
    """post-LN ``Wav2Vec2EncoderLayer`` (:591-608) / pre-LN ``...StableLayerNorm`` (:631-654)."""
    pre = f"wav2vec2.encoder.layers.{l}."
    eps = cfg.layer_norm_eps
    if not cfg.do_stable_layer_norm:
        a = attention(p, cfg, pre + "attention.", h, key_mask, train, drop, l)
        h = h + _drop(a, cfg.hidden_dropout, train, drop, 16 + 4 * l + 1)
        h = _ln(p, pre + "layer_norm", h, eps)
        h = h + feed_forward(p, cfg, pre + "feed_forward.", h, train, drop, l)
        return _ln(p, pre + "final_layer_norm", h, eps)
    a = attention(p, cfg, pre + "attention.", _ln(p, pre + "layer_norm", h, eps), key_mask, train, drop, l)
    h = h + _drop(a, cfg.hidden_dropout, train, drop, 16 + 4 * l + 1)
    return h + feed_forward(p, cfg, pre + "feed_forward.", _ln(p, pre + "final_layer_norm", h, eps), train, drop, l)


def forward(p: Dict[str, torch.Tensor], cfg: W2V2Config, input_values: torch.Tensor,
            lengths=None, labels: Optional[torch.Tensor] = None, train: bool = False,
            mask_time_indices: Optional[torch.Tensor] = None, layer_keep=None, stages: Optional[dict] = None,
            gradient_checkpointing: bool = False, drop: Optional[HashDropout] = None):
    """``Wav2Vec2ForCTC.forward`` (:1667-1742) through ``Wav2Vec2Model.forward`` (:1319-1380) and
    ``Wav2Vec2Encoder.forward`` (:667-726).

    ``lengths`` (samples per utterance) stands for ``attention_mask`` (None = no mask, the
    group-norm/base convention).  ``mask_time_indices`` [B,F] bool is the
    SpecAugment mask (``_mask_hidden_states`` :1272-1316) supplied by the caller; ``layer_keep``
    is the per-layer LayerDrop decision (:701-712); ``drop`` (with ``train=True``) replaces torch's dropout generator
    by the engine's counter-hash masks (site ids of w2v2_engine.hip:127-131).  Returns (loss | None, logits[B,F,V])."""
    B, T = input_values.shape
    eps = cfg.layer_norm_eps
    feats = feature_encoder(p, cfg, input_values, stages).transpose(1, 2)  # [B,F,C]
    Fr = feats.shape[1]
    key_mask = None
    if lengths is not None:
        fl = torch.as_tensor(conv_out_lengths(cfg, np.asarray(lengths)))
        key_mask = torch.arange(Fr)[None, :] < fl[:, None]  # _get_feature_vector_attention_mask :1018-1036
    nf = _ln(p, "wav2vec2.feature_projection.layer_norm", feats, eps)  # :429-434
    h = F.linear(nf, p["wav2vec2.feature_projection.projection.weight"],
                 p["wav2vec2.feature_projection.projection.bias"])
    h = _drop(h, cfg.feat_proj_dropout, train, drop, 1)
    if stages is not None:
        stages["feat_proj"] = h
    if mask_time_indices is not None:
        h = torch.where(mask_time_indices[:, :, None], p["wav2vec2.masked_spec_embed"][None, None, :], h)
    if key_mask is not None:
        h = h * key_mask[:, :, None]  # :678-681 padded frames := 0
    h = h + pos_conv_embed(p, cfg, h)
    if stages is not None:
        stages["pos_conv_added"] = h
    if not cfg.do_stable_layer_norm:
        h = _ln(p, "wav2vec2.encoder.layer_norm", h, eps)
    h = _drop(h, cfg.hidden_dropout, train, drop, 2)
    if stages is not None:
        stages["encoder_in"] = h
    for l in range(cfg.num_hidden_layers):
        if layer_keep is not None and not layer_keep[l]:
            continue
        if gradient_checkpointing and torch.is_grad_enabled():
            # model.gradient_checkpointing_enable() of the train script (wav2vec_train.py:329): each encoder layer's
            # activations are recomputed in the backward (modeling_wav2vec2.py: GradientCheckpointingLayer)
            from torch.utils.checkpoint import checkpoint
            h = checkpoint(lambda t, l=l: encoder_layer(p, cfg, l, t, key_mask, train, drop), h, use_reentrant=False,
                           preserve_rng_state=train)
        else:
            h = encoder_layer(p, cfg, l, h, key_mask, train, drop)
        if stages is not None:
            stages[f"layer{l}"] = h
    if cfg.do_stable_layer_norm:
        h = _ln(p, "wav2vec2.encoder.layer_norm", h, eps)
    if stages is not None:
        stages["last_hidden"] = h  # Wav2Vec2Model(...)[0]: what the SpeechBrain recipe's wav2vec2 module returns
    h = _drop(h, cfg.final_dropout, train, drop, 3)
    logits = F.linear(h, p["lm_head.weight"], p["lm_head.bias"])
    loss = None
    if labels is not None:
        if int(labels.max()) >= cfg.vocab_size:
            raise ValueError(f"Label values must be <= vocab_size: {cfg.vocab_size}")  # :1686-1687
        in_len = torch.as_tensor(conv_out_lengths(cfg, np.full(B, T) if lengths is None else np.asarray(lengths)))
        lm = labels >= 0
        logp = F.log_softmax(logits, dim=-1, dtype=torch.float32).transpose(0, 1)
        loss = F.ctc_loss(logp, labels.masked_select(lm), in_len, lm.sum(-1), blank=cfg.pad_token_id,
                          reduction=cfg.ctc_loss_reduction, zero_infinity=cfg.ctc_zero_infinity)
    return loss, logits


def trainable_names(cfg: W2V2Config, freeze_feature_encoder: bool = True):
    return [n for n in param_shapes(cfg) if not (freeze_feature_encoder and is_feature_encoder_param(n))]


def loss_and_grads(p, cfg, input_values, lengths, labels, freeze_feature_encoder=True, **kw):
    """Forward + autograd backward (``trainer.py:2548`` ``loss.backward()``); returns loss, logits, grads."""
    names = trainable_names(cfg, freeze_feature_encoder)
    q = {n: (t.detach().clone().requires_grad_(True) if n in names else t.detach()) for n, t in p.items()}
    loss, logits = forward(q, cfg, input_values, lengths, labels, **kw)
    loss.backward()
    grads = {n: (q[n].grad if q[n].grad is not None else torch.zeros_like(q[n])) for n in names}
    return loss.detach(), logits.detach(), grads


# ----------------------------------------------------------------------------- SpecAugment indices (a5)
def compute_mask_indices(shape, mask_prob, mask_length, lengths=None, min_masks=0, rng=np.random) -> np.ndarray:
    """``_compute_mask_indices`` (modeling_wav2vec2.py:101-217) driven by an explicit numpy RNG
    (same draw order: one ``rand`` for epsilon, then one ``choice`` per utterance)."""
    B, S = shape
    if mask_length > S:
        raise ValueError("`mask_length` has to be smaller than `sequence_length`")
    eps = rng.rand(1).item()

    def nspan(L):
        n = max(int(mask_prob * L / mask_length + eps), min_masks)
        if n * mask_length > S:
            n = S // mask_length
        if L - (mask_length - 1) < n:
            n = max(L - (mask_length - 1), 0)
        return n

    in_len = [S] * B if lengths is None else list(lengths)
    mask = np.zeros((B, S), dtype=bool)
    nmax = nspan(S)
    if nmax == 0:
        return mask
    for b, L in enumerate(in_len):
        n = nspan(L)
        idx = rng.choice(np.arange(L - (mask_length - 1)), n, replace=False)
        dummy = S - 1 if len(idx) == 0 else idx[0]
        idx = np.concatenate([idx, np.ones(nmax - n, dtype=np.int32) * dummy])
        span = np.minimum(idx[:, None] + np.arange(mask_length)[None, :], S - 1).reshape(-1)
        mask[b, span.astype(np.int64)] = True
    return mask


# ----------------------------------------------------------------------------- greedy CTC decode (a12)
def greedy_ctc_ids(logits: np.ndarray, blank: int = 0):
    """argmax + collapse repeats + drop blank (transformers_infer.py:84-85 ->
    ``Wav2Vec2CTCTokenizer.batch_decode`` group_tokens=True, pad token = blank)."""
    out = []
    for ids in np.asarray(logits).argmax(-1):
        keep = np.ones(len(ids), dtype=bool)
        keep[1:] = ids[1:] != ids[:-1]
        ids = ids[keep]
        out.append([int(i) for i in ids if i != blank])
    return out


def ids_to_text(ids, vocab, word_delimiter: str = "|") -> str:
    """token ids -> string: join, word delimiter -> space, strip (tokenization_wav2vec2 ``convert_tokens_to_string``)."""
    return "".join(" " if vocab[i] == word_delimiter else vocab[i] for i in ids).strip()
