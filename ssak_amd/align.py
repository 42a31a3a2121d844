"""CTC forced alignment on the device: the counterpart of ``ssak/utils/align_transcriptions.py``.

Same function names, arguments and results as the reference (``get_trellis`` :27-70, ``backtrack`` :79-123,
``merge_repeats`` :140-156, ``merge_words`` :158-172, ``compute_alignment`` :294-402, ``loose_get_char_index`` :406-426);
the Viterbi trellis and the backtrack run in one HIP launch (``ssak_ctc_forced_align``) instead of a Python loop of torch
ops per frame.  Plotting (``plot=``) is outside this path.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import hip


@dataclasses.dataclass
class Point:
    token_index: int
    time_index: int
    score: float


@dataclasses.dataclass
class Segment:
    label: str
    start: int
    end: int
    score: float

    def __repr__(self):
        return f"{self.label}\t({self.score:4.2f}): [{self.start:5d}, {self.end:5d})"

    @property
    def length(self):
        return self.end - self.start


_last = {}  # trellis data_ptr -> path of the launch that produced it (get_trellis and backtrack are one kernel)


def forced_align(emission: torch.Tensor, tokens: Sequence[int], blank_id: int = 0, first_as_garbage: bool = False):
    """emission [F, V] float32 log-probabilities (any device) -> (trellis [F+1, L+1] on the emission's device,
    path: list of Point or None when the reference would raise "Failed to align")."""
    if len(tokens) == 0:
        raise IndexError("forced alignment needs a non-empty transcript")  # the reference fails on tokens[j - 1] too
    src_dev = emission.device
    dev = src_dev if emission.is_cuda else torch.device("cuda:0")
    em = emission.to(device=dev, dtype=torch.float32).contiguous()
    F, V = em.shape
    tok_host = torch.as_tensor(list(tokens), dtype=torch.int32)
    if int(tok_host.min()) < 0 or int(tok_host.max()) >= V:
        raise IndexError(f"token id outside the emission's {V} classes")
    tok = tok_host.to(dev)
    L = tok.numel()
    col0 = None
    if first_as_garbage:
        # column 0 of the garbage variant (:38) is F transcendental evaluations: done with the same torch ops as the
        # reference, then handed to the kernel, which applies the borders of :42-43
        c = torch.zeros(F + 1, dtype=torch.float32)
        c[1:] = (1 - emission[:, int(tok_host[0])].detach().float().cpu().exp()).log()
        col0 = c.to(dev)
    trellis = torch.empty((F + 1, L + 1), dtype=torch.float32, device=dev)
    path_token = torch.empty(F, dtype=torch.int32, device=dev)
    path_logp = torch.empty(F, dtype=torch.float32, device=dev)
    info = torch.empty(2, dtype=torch.int32, device=dev)
    ws = torch.empty(hip.lib.ssak_ctc_align_workspace_bytes(F, L), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        hip.check(hip.lib.ssak_ctc_forced_align(hip.ptr(em), hip.ptr(tok), F, V, L, int(blank_id), hip.ptr(col0), hip.ptr(trellis),
                                                hip.ptr(path_token), hip.ptr(path_logp), hip.ptr(info), hip.ptr(ws), ws.numel(),
                                                hip.stream()))
    n, first = (int(v) for v in info.cpu())
    path = None
    if n >= 0:
        tk = path_token[first:first + n].cpu().numpy()
        sc = np.exp(path_logp[first:first + n].cpu().numpy())  # Point.score = emission[...].exp()
        path = [Point(int(tk[i]), first + i, float(sc[i])) for i in range(n)]
    return trellis.to(src_dev), path


def get_trellis(emission, tokens, blank_id=0, first_as_garbage=False):
    trellis, path = forced_align(emission, tokens, blank_id, first_as_garbage)
    _last.clear()
    _last[(trellis.data_ptr(), tuple(tokens), int(blank_id))] = path
    return trellis


def backtrack(trellis, emission, tokens, blank_id=0):
    key = (trellis.data_ptr(), tuple(tokens), int(blank_id))
    if key in _last:
        path = _last[key]
    else:  # a trellis that did not come from get_trellis: one more launch gives the same walk
        _, path = forced_align(emission, tokens, blank_id, first_as_garbage=False)
    if path is None:
        raise RuntimeError("Failed to align (not enough tokens for the duration?)")
    return path


def merge_repeats(transcript, path: List[Point]) -> List[Segment]:
    i1, i2 = 0, 0
    segments = []
    while i1 < len(path):
        while i2 < len(path) and path[i1].token_index == path[i2].token_index:
            i2 += 1
        score = sum(path[k].score for k in range(i1, i2)) / (i2 - i1)
        segments.append(Segment(transcript[path[i1].token_index], path[i1].time_index, path[i2 - 1].time_index + 1, score))
        i1 = i2
    return segments


def merge_words(segments: List[Segment], separator: str = " ") -> List[Segment]:
    words = []
    i1, i2 = 0, 0
    while i1 < len(segments):
        if i2 >= len(segments) or segments[i2].label == separator:
            if i1 != i2:
                segs = segments[i1:i2]
                word = "".join(seg.label for seg in segs)
                score = sum(seg.score * seg.length for seg in segs) / sum(seg.length for seg in segs)
                words.append(Segment(word, segments[i1].start, segments[i2 - 1].end, score))
            i1 = i2 + 1
            i2 = i1
        else:
            i2 += 1
    return words


# ssak/utils/text_basic.py:15-16: string.punctuation + the listed extra marks, minus "-" and "'"
_PUNCTUATION = "".join(c for c in __import__("string").punctuation + "。，！？：”、…" + "؟،؛" + "—" + "«°»×‹›•“–‘″‘" if c not in "-'")


def get_model_vocab(model_and_processor):
    """(labels with "|" shown as " ", blank id = index of <pad> / [PAD]) -- ssak/infer/general.py:133-141."""
    tok = model_and_processor[1]
    vocab = tok.get_vocab() if hasattr(tok, "get_vocab") else {c: i for i, c in enumerate(tok.vocab)}
    inv = {v: k for k, v in vocab.items()}
    labels = [inv[i] for i in range(len(inv))]
    labels = [l if l != "|" else " " for l in labels]
    blank_id = labels.index("<pad>") if "<pad>" in labels else labels.index("[PAD]") if "[PAD]" in labels else -1
    if blank_id == -1:
        raise ValueError("Neither <pad> nor [PAD] found in labels")
    return labels, blank_id


def loose_get_char_index(dictionary, c, default):
    """Label id of character ``c``, trying its case variants, else ``default`` (None drops it) -- :406-426 without the
    transliteration table, which lives in the reference's text-normalisation module (out of scope)."""
    i = dictionary.get(c)
    if i is None:
        for c2 in (c.lower(), c.upper()):
            i = dictionary.get(c2)
            if i is not None:
                break
    return default if i is None else i


def compute_alignment(audio, transcript, model, add_before_after=None, first_as_garbage=False, plot=False, verbose=False):
    """(labels, emission, trellis, char_segments, word_segments) as align_transcriptions.py:294-402."""
    from .infer import compute_log_probas
    if plot:
        raise NotImplementedError("plotting is outside the device path")
    emission = compute_log_probas(model, audio)
    labels, blank_id = get_model_vocab(model)
    if transcript is None:
        ids, n = hip.ctc_greedy_decode(emission.to(model[0].device)[None].contiguous(), None, blank_id)
        transcript = "".join(labels[i] for i in ids[0, :int(n[0])].cpu().tolist())
    if isinstance(transcript, str):
        transcript_characters, transcript_words = transcript, None
    else:
        assert isinstance(transcript, list), f"Got unexpected transcript (of type {type(transcript)})"
        for w in transcript:
            assert isinstance(w, str), f"Got unexpected type {type(w)} (not a string)"
        transcript_characters, transcript_words = " ".join(transcript), transcript
    space_id = labels.index(" ") if " " in labels else blank_id
    if add_before_after:
        assert len(add_before_after) == 1 and add_before_after in labels
        transcript_characters = add_before_after + transcript_characters + add_before_after
    labels = labels[:emission.shape[1]]
    dictionary = {c: i for i, c in enumerate(labels)}
    tokens = [loose_get_char_index(dictionary, c, space_id) for c in transcript_characters]
    tokens = [i for i in tokens if i is not None]
    trellis = get_trellis(emission, tokens, blank_id=blank_id, first_as_garbage=first_as_garbage)
    path = backtrack(trellis, emission, tokens, blank_id=blank_id)
    char_segments = merge_repeats(transcript_characters, path)
    if add_before_after:
        assert char_segments[0].label == add_before_after and char_segments[-1].label == add_before_after
        char_segments = char_segments[1:-1]
        trellis = trellis[:, [0] + list(range(2, trellis.shape[1] - 1))]
        transcript_characters = transcript_characters[1:-1]
    if transcript_words is None:
        word_segments = merge_words(char_segments)
    else:
        word_segments = []
        i2 = -1
        for word in transcript_words:
            i1 = i2 + 1
            i2 = i1 + len(word)
            segs1 = char_segments[i1:i2]
            assert "".join(seg.label for seg in segs1) == word
            segs2 = [s for s in segs1 if s.label not in " " + _PUNCTUATION]
            segs = segs2 if len(segs2) != 0 else segs1
            score = sum(seg.score * seg.length for seg in segs) / sum(seg.length for seg in segs)
            word_segments.append(Segment(word, segs[0].start, segs[-1].end, score))
    return labels, emission, trellis, char_segments, word_segments
