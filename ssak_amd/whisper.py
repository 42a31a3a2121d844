"""Whisper encoder + CTC head on the HIP engine (BASELINE config 4).

The reference has no such model: it fine-tunes Whisper as a seq2seq model with cross-entropy
(ssak/train/transformers/whisper_train.py:432,498-507).  This composition -- ``WhisperEncoder`` (transformers
modeling_whisper.py:592-642) -> ``Linear(d_model, vocab)`` -> CTC -- is the build's; its log-mel input is produced on
the device by ``ssak_amd.hip.logmel_whisper`` (a13), the call site being ssak/utils/dataset.py:632-637.
Parameter names follow ``WhisperEncoder``'s state dict with an ``encoder.`` prefix, plus ``ctc_head.*``.
"""
from __future__ import annotations

import dataclasses

import torch

from . import hip
from .model import Wav2Vec2ForCTC


@dataclasses.dataclass
class WhisperCTCConfig:
    """whisper-small encoder by default (d_model 768, 12 layers, 12 heads, ffn 3072, 80 mels, 1500 positions)."""
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
    pad_token_id: int = 0
    ctc_loss_reduction: str = "mean"
    ctc_zero_infinity: bool = True
    # attributes the shared host code reads
    mask_time_prob: float = 0.0
    mask_time_length: int = 10
    mask_time_min_masks: int = 0
    feat_extract_norm: str = "group"  # no attention mask

    @property
    def num_hidden_layers(self):
        return self.encoder_layers

    @property
    def layerdrop(self):
        return self.encoder_layerdrop

    @property
    def conv_kernel(self):
        return (3, 3)

    @property
    def conv_stride(self):
        return (1, 2)


class WhisperEncoderForCTC(Wav2Vec2ForCTC):
    _HEAD = ("ctc_head.weight", "ctc_head.bias")

    """``model(input_features [B, 80, 2*frames], labels=...)`` -> ``.loss`` / ``.logits [B, frames, V]``."""

    def __init__(self, config: WhisperCTCConfig, device: str = "cuda:0", seed: int = 69, exact: bool = False):
        """``exact``: the fp32-exact verification mode of the engine (float activations, fp32 matrix products)."""
        if not torch.cuda.is_available():
            raise RuntimeError("ssak_amd needs an MI355X: there is no CPU fallback for the acoustic model")
        self.config = config
        self.device = torch.device(device)
        self.training = False
        self.freeze = True
        self.exact = bool(exact)
        c = hip.W2V2Config()
        c.arch = 1
        c.exact = int(self.exact)
        c.vocab_size = (config.vocab_size + 7) // 8 * 8
        c.hidden_size, c.num_layers = config.d_model, config.encoder_layers
        c.num_heads, c.intermediate_size = config.encoder_attention_heads, config.encoder_ffn_dim
        c.num_mel_bins, c.max_source_positions = config.num_mel_bins, config.max_source_positions
        c.layer_norm_eps = 1e-5
        c.do_stable_layer_norm = 1
        c.attention_dropout, c.hidden_dropout = config.attention_dropout, config.dropout
        c.activation_dropout, c.feat_proj_dropout, c.final_dropout = config.activation_dropout, 0.0, 0.0
        self._finish_init(c, seed)

    def load_state_dict(self, sd, strict: bool = True):
        sd = dict(sd)
        for name in self.layout:  # the synthetic zero bias slots of k_proj are not part of a Whisper checkpoint
            if name.endswith("k_proj.bias") and name not in sd:
                sd[name] = torch.zeros(self.layout[name][2])
        return super().load_state_dict(sd, strict)

    def state_dict(self):
        return {n: t for n, t in super().state_dict().items() if not n.endswith("k_proj.bias")}
