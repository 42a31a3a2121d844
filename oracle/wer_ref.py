"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the evaluation metric of the reference's train
script, ssak/train/transformers/wav2vec_train.py:107-125: greedy decode -> ``processor.batch_decode`` (predictions
grouped, labels with ``group_tokens=False``) -> ``remove_special_words(x, glue_apostrophe=False)``
(ssak/utils/text_basic.py:91-110) -> ``datasets.load_metric("wer")``, i.e. jiwer's word-level Levenshtein measures
summed over the utterances: wer = sum(S + D + I) / sum(S + D + H).  ``datasets.load_metric`` no longer exists in the
installed datasets 5.x and jiwer is not installed, so the metric is pinned by known-answer cases (tests/test_oracle.py),
not by running the third-party code."""
from __future__ import annotations

import re
from typing import List, Sequence

import numpy as np


def ids_to_text(ids: Sequence[int], vocab: Sequence[str], pad_id: int, group_tokens: bool, delimiter: str = "|") -> str:
    """Wav2Vec2CTCTokenizer.decode semantics: optional grouping of repeats, the pad token (= CTC blank) dropped, the word
    delimiter shown as a space; other special tokens stay in the text as "<...>"."""
    out, prev = [], None
    for i in ids:
        i = int(i)
        if i < 0:
            continue
        if group_tokens and i == prev:
            continue
        prev = i
        if i == pad_id:
            continue
        out.append(" " if vocab[i] == delimiter else vocab[i])
    return re.sub(r"\s+", " ", "".join(out)).strip()


def format_words_for_wer(text: str) -> str:
    """remove_special_words(text, glue_apostrophe=False), text_basic.py:99-110."""
    if not text:
        return ""
    text = re.sub(r"<.*?>", "", text)
    text = re.sub(r"'", "' ", text).strip()
    return re.sub(r"\s+", " ", text).strip()


def word_edits(ref_words: List[str], hyp_words: List[str]) -> int:
    """Levenshtein distance over words (substitution, deletion, insertion all cost 1)."""
    d = np.arange(len(hyp_words) + 1)
    for i, rw in enumerate(ref_words, 1):
        prev, d[0] = d[0], i
        for j, hw in enumerate(hyp_words, 1):
            cur = min(d[j] + 1, d[j - 1] + 1, prev + (rw != hw))
            prev, d[j] = d[j], cur
    return int(d[len(hyp_words)])


def compute_metrics(pred_ids, label_ids, vocab: Sequence[str], pad_id: int):
    """pred_ids [B, F] = argmax of the logits (NOT yet collapsed), label_ids [B, L] with -100 padding ->
    (per-utterance edits, per-utterance reference words, wer)."""
    edits, nref = [], []
    for p, l in zip(pred_ids, label_ids):
        l = [pad_id if int(x) == -100 else int(x) for x in l]                          # :114
        hyp = format_words_for_wer(ids_to_text(p, vocab, pad_id, True)).split()        # :116,120
        ref = format_words_for_wer(ids_to_text(l, vocab, pad_id, False)).split()       # :118,121
        edits.append(word_edits(ref, hyp))
        nref.append(len(ref))
    return np.array(edits), np.array(nref), float(sum(edits)) / max(1, sum(nref))
