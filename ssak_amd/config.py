"""Model configuration of the acoustic model (the subset of ``transformers.Wav2Vec2Config`` the path reads).

Defaults are wav2vec2-base with the regularisers the reference's train script passes
(ssak/train/transformers/wav2vec_train.py:161-165,313-325).
"""
from __future__ import annotations

import dataclasses
import json
from typing import Tuple


@dataclasses.dataclass
class Wav2Vec2Config:
    vocab_size: int = 32
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    conv_bias: bool = False
    feat_extract_norm: str = "group"
    do_stable_layer_norm: bool = False
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
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
    ctc_loss_reduction: str = "mean"
    ctc_zero_infinity: bool = True

    def deterministic(self) -> "Wav2Vec2Config":
        return dataclasses.replace(self, attention_dropout=0.0, hidden_dropout=0.0, activation_dropout=0.0,
                                   feat_proj_dropout=0.0, final_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0)

    @classmethod
    def from_hf_dict(cls, d: dict) -> "Wav2Vec2Config":
        names = {f.name for f in dataclasses.fields(cls)}
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items() if k in names}
        return cls(**kw)

    @classmethod
    def from_json_file(cls, path: str) -> "Wav2Vec2Config":
        with open(path) as f:
            return cls.from_hf_dict(json.load(f))

    def to_dict(self) -> dict:
        d = dataclasses.asdict(self)
        for k in ("conv_dim", "conv_kernel", "conv_stride"):
            d[k] = list(d[k])
        d["model_type"] = "wav2vec2"
        d["architectures"] = ["Wav2Vec2ForCTC"]
        return d
