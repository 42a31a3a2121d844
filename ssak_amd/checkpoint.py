"""Model folders in the HuggingFace layout the reference reads and writes (``config.json``, ``vocab.json``,
``model.safetensors`` / ``pytorch_model.bin``; ssak/train/transformers/wav2vec_train.py:419-420,
ssak/infer/transformers_infer.py:140-169)."""
from __future__ import annotations

import json
import os
from typing import Dict

import torch

from .config import Wav2Vec2Config
from .data import CharTokenizer

_LEGACY = {"wav2vec2.encoder.pos_conv_embed.conv.weight_g": "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0",
           "wav2vec2.encoder.pos_conv_embed.conv.weight_v": "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1"}


def load_state_dict_file(folder: str) -> Dict[str, torch.Tensor]:
    st = os.path.join(folder, "model.safetensors")
    if os.path.isfile(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(folder, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    return {_LEGACY.get(k, k): v for k, v in sd.items()}


def save_pretrained(model, tokenizer: CharTokenizer, folder: str):
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "config.json"), "w") as f:
        json.dump(model.config.to_dict(), f, indent=1)
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in model.state_dict().items()}, os.path.join(folder, "model.safetensors"))
    if tokenizer is not None:
        tokenizer.save(folder)
    with open(os.path.join(folder, "preprocessor_config.json"), "w") as f:
        json.dump({"do_normalize": True, "feature_size": 1, "padding_value": 0.0, "sampling_rate": 16000,
                   "return_attention_mask": model.config.feat_extract_norm == "layer"}, f, indent=1)


def load_pretrained(folder: str, device: str = "cuda:0", freeze_feature_encoder: bool = True, **config_overrides):
    """-> (model, tokenizer).  Parameters absent from the checkpoint (e.g. a resized lm_head) keep their init."""
    from .model import Wav2Vec2ForCTC
    import dataclasses
    cfg = Wav2Vec2Config.from_json_file(os.path.join(folder, "config.json"))
    cfg = dataclasses.replace(cfg, **config_overrides)
    tok = CharTokenizer.from_vocab_json(os.path.join(folder, "vocab.json"))
    model = Wav2Vec2ForCTC(cfg, device=device, freeze_feature_encoder=freeze_feature_encoder)
    sd = load_state_dict_file(folder)
    sd = {k: v for k, v in sd.items() if k in model.layout}
    model.load_state_dict(sd, strict=False)
    return model, tok
