"""Eager torch-CPU fp32 restatement of the Whisper encoder + a CTC head.  TEST INFRASTRUCTURE ONLY.

BASELINE config 4 ("Whisper-small encoder + CTC head") has NO counterpart in the reference: linto-ai/ssak trains
Whisper as a seq2seq model with cross-entropy (ssak/train/transformers/whisper_train.py:432,498-507) and never puts
a CTC head on the encoder.  The composition is the build's: ``WhisperEncoder`` (transformers
modeling_whisper.py:592-642, layers :360-413) -> ``Linear(d_model, vocab)`` -> ``log_softmax`` -> ``F.ctc_loss``
("mean", zero_infinity).  Parity is against that composition, pinned to ``transformers.WhisperEncoder`` by
``oracle/gen_golden.py`` -> ``tests/golden/whisper_tiny.npz``.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclasses.dataclass
class WhisperCTCConfig:
    """whisper-small encoder dimensions by default."""
    vocab_size: int = 56
    num_mel_bins: int = 80
    d_model: int = 768
    encoder_layers: int = 12
    encoder_attention_heads: int = 12
    encoder_ffn_dim: int = 3072
    max_source_positions: int = 1500
    dropout: float = 0.0
    attention_dropout: float = 0.0
    activation_dropout: float = 0.0
    encoder_layerdrop: float = 0.0
    pad_token_id: int = 0  # CTC blank

    @staticmethod
    def tiny(**kw) -> "WhisperCTCConfig":
        d = dict(vocab_size=32, d_model=64, encoder_layers=2, encoder_attention_heads=4, encoder_ffn_dim=128,
                 max_source_positions=50)
        d.update(kw)
        return WhisperCTCConfig(**d)


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """Whisper's fixed positional table (modeling_whisper.py ``sinusoids``)."""
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length).view(-1, 1) * inv.view(1, -1)
    return torch.cat([t.sin(), t.cos()], dim=1)


def param_shapes(cfg: WhisperCTCConfig) -> Dict[str, Tuple[int, ...]]:
    H, I = cfg.d_model, cfg.encoder_ffn_dim
    s = {"encoder.conv1.weight": (H, cfg.num_mel_bins, 3), "encoder.conv1.bias": (H,),
         "encoder.conv2.weight": (H, H, 3), "encoder.conv2.bias": (H,),
         "encoder.embed_positions.weight": (cfg.max_source_positions, H)}
    for l in range(cfg.encoder_layers):
        p = f"encoder.layers.{l}."
        s[p + "self_attn.k_proj.weight"] = (H, H)  # no bias on k_proj
        for n in ("v_proj", "q_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (H, H)
            s[p + f"self_attn.{n}.bias"] = (H,)
        s[p + "self_attn_layer_norm.weight"] = (H,)
        s[p + "self_attn_layer_norm.bias"] = (H,)
        s[p + "fc1.weight"] = (I, H)
        s[p + "fc1.bias"] = (I,)
        s[p + "fc2.weight"] = (H, I)
        s[p + "fc2.bias"] = (H,)
        s[p + "final_layer_norm.weight"] = (H,)
        s[p + "final_layer_norm.bias"] = (H,)
    s["encoder.layer_norm.weight"] = (H,)
    s["encoder.layer_norm.bias"] = (H,)
    s["ctc_head.weight"] = (cfg.vocab_size, H)
    s["ctc_head.bias"] = (cfg.vocab_size,)
    return s


def init_params(cfg: WhisperCTCConfig, seed: int = 69) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out = {}
    for n, shp in param_shapes(cfg).items():
        if n.endswith("embed_positions.weight"):
            t = sinusoids(*shp)
        elif "layer_norm" in n:
            t = (1.0 if n.endswith("weight") else 0.0) + 0.05 * torch.randn(shp, generator=g)
        elif n.endswith(".bias"):
            t = 0.02 * torch.randn(shp, generator=g)
        elif "conv" in n:
            t = torch.randn(shp, generator=g) * math.sqrt(1.0 / (shp[1] * shp[2]))
        else:
            t = torch.randn(shp, generator=g) * 0.05
        out[n] = t.float().contiguous()
    return out


def trainable_names(cfg: WhisperCTCConfig):
    return [n for n in param_shapes(cfg) if not n.endswith("embed_positions.weight")]  # fixed sinusoids (requires_grad False)


def forward(p, cfg: WhisperCTCConfig, mel: torch.Tensor, labels: Optional[torch.Tensor] = None, layer_keep=None):
    """mel [B, 80, 2*max_source_positions] -> (loss | None, logits [B, max_source_positions, V])."""
    H, nh = cfg.d_model, cfg.encoder_attention_heads
    hd = H // nh
    x = F.gelu(F.conv1d(mel, p["encoder.conv1.weight"], p["encoder.conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, p["encoder.conv2.weight"], p["encoder.conv2.bias"], stride=2, padding=1))
    h = x.permute(0, 2, 1) + p["encoder.embed_positions.weight"]
    B, T, _ = h.shape
    for l in range(cfg.encoder_layers):
        if layer_keep is not None and not layer_keep[l]:
            continue
        q_ = f"encoder.layers.{l}."
        y = F.layer_norm(h, (H,), p[q_ + "self_attn_layer_norm.weight"], p[q_ + "self_attn_layer_norm.bias"], 1e-5)
        q = F.linear(y, p[q_ + "self_attn.q_proj.weight"], p[q_ + "self_attn.q_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
        k = F.linear(y, p[q_ + "self_attn.k_proj.weight"]).view(B, T, nh, hd).transpose(1, 2)
        v = F.linear(y, p[q_ + "self_attn.v_proj.weight"], p[q_ + "self_attn.v_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
        a = F.softmax(torch.matmul(q, k.transpose(2, 3)) * hd ** -0.5, dim=-1)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, H)
        h = h + F.linear(o, p[q_ + "self_attn.out_proj.weight"], p[q_ + "self_attn.out_proj.bias"])
        y = F.layer_norm(h, (H,), p[q_ + "final_layer_norm.weight"], p[q_ + "final_layer_norm.bias"], 1e-5)
        y = F.linear(F.gelu(F.linear(y, p[q_ + "fc1.weight"], p[q_ + "fc1.bias"])), p[q_ + "fc2.weight"], p[q_ + "fc2.bias"])
        h = h + y
    h = F.layer_norm(h, (H,), p["encoder.layer_norm.weight"], p["encoder.layer_norm.bias"], 1e-5)
    logits = F.linear(h, p["ctc_head.weight"], p["ctc_head.bias"])
    loss = None
    if labels is not None:
        lm = labels >= 0
        lp = F.log_softmax(logits, dim=-1, dtype=torch.float32).transpose(0, 1)
        loss = F.ctc_loss(lp, labels.masked_select(lm), torch.full((B,), T), lm.sum(-1), blank=cfg.pad_token_id,
                          reduction="mean", zero_infinity=True)
    return loss, logits


def loss_and_grads(p, cfg, mel, labels, **kw):
    names = trainable_names(cfg)
    q = {n: (t.detach().clone().requires_grad_(True) if n in names else t.detach()) for n, t in p.items()}
    loss, logits = forward(q, cfg, mel, labels, **kw)
    loss.backward()
    return loss.detach(), logits.detach(), {n: q[n].grad for n in names}
