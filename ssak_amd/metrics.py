"""Evaluation metric of the train script on the device: the counterpart of ``compute_metrics``
(ssak/train/transformers/wav2vec_train.py:107-125).  Logits stay in HBM; greedy decode (``ssak_ctc_greedy_decode``) and the
word-level edit counts (``ssak_ctc_wer``) run as kernels and two integers per utterance come back."""
from __future__ import annotations

from typing import Sequence

import torch

from . import hip


def token_classes(vocab: Sequence[str], delimiter: str = "|") -> torch.Tensor:
    """uint8 [V] table for ``ssak_ctc_wer``: 1 = word separator, 2 = "<...>" token (deleted from the text by
    remove_special_words), 3 = apostrophe (ends its word under glue_apostrophe=False), 0 = letter."""
    cls = []
    for t in vocab:
        if t == delimiter or t == " ":
            cls.append(1)
        elif len(t) >= 2 and t.startswith("<") and t.endswith(">"):
            cls.append(2)
        elif t == "'":
            cls.append(3)
        else:
            cls.append(0)
    return torch.tensor(cls, dtype=torch.uint8)


class WerAccumulator:
    """Running sums over an evaluation set; ``add`` issues kernels only, ``compute`` does the single host read."""

    def __init__(self, vocab: Sequence[str], pad_id: int, device="cuda:0", delimiter: str = "|"):
        self.cls = token_classes(vocab, delimiter).to(device)
        self.pad_id = pad_id
        self.sums = torch.zeros(2, dtype=torch.int64, device=device)

    def add(self, logits: torch.Tensor, labels: torch.Tensor, frame_lens=None):
        """logits [B, F, V] fp32 on the device, labels [B, L] with -100 padding."""
        ids, n = hip.ctc_greedy_decode(logits.contiguous(), frame_lens, self.pad_id)
        edits, nref = hip.ctc_wer(ids, n, labels, self.cls)
        self.sums += torch.stack([edits.sum(), nref.sum()]).to(torch.int64)
        return edits, nref

    def compute(self) -> dict:
        e, r = (int(v) for v in self.sums.cpu())
        return {"wer": e / max(1, r)}


def compute_metrics(logits: torch.Tensor, labels: torch.Tensor, vocab: Sequence[str], pad_id: int, frame_lens=None) -> dict:
    acc = WerAccumulator(vocab, pad_id, logits.device)
    acc.add(logits, labels, frame_lens)
    return acc.compute()
