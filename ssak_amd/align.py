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

    def __repr__(self):  # same text as the reference prints (:147)
        return "%s\t(%4.2f): [%5d, %5d)" % (self.label, self.score, self.start, self.end)

    @property
    def length(self):
        return self.end - self.start


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
    """The Viterbi trellis [F+1, L+1] (:27-70).  The kernel walks the best path in the same launch; it travels WITH the
    returned tensor (attribute ``ssak_alignment``: the token sequence, the blank id and the path) so that :func:`backtrack` on
    this very object needs no second launch.  Tensors derived from it (slices, copies) do not carry the attribute."""
    trellis, path = forced_align(emission, tokens, blank_id, first_as_garbage)
    trellis.ssak_alignment = (tuple(int(t) for t in tokens), int(blank_id), path)
    return trellis


def backtrack(trellis, emission, tokens, blank_id=0):
    """Best path as a list of Point (:79-123); RuntimeError when the transcript does not fit the frames."""
    carried = getattr(trellis, "ssak_alignment", None)
    if carried is not None and carried[0] == tuple(int(t) for t in tokens) and carried[1] == int(blank_id):
        path = carried[2]
    else:  # a trellis that did not come from get_trellis (or other tokens): one more launch gives the walk
        _, path = forced_align(emission, tokens, blank_id, first_as_garbage=False)
    if path is None:
        raise RuntimeError("Failed to align (not enough tokens for the duration?)")
    return path


def _run_starts(values: np.ndarray) -> np.ndarray:
    """Indices where a new run of equal values begins."""
    if len(values) == 0:
        return np.zeros(0, dtype=np.int64)
    return np.flatnonzero(np.concatenate(([True], values[1:] != values[:-1])))


def merge_repeats(transcript, path: List[Point]) -> List[Segment]:
    """Consecutive path points on the same transcript position -> one Segment per position: frames [first, last + 1), score =
    mean of the points' scores (:140-156).  Run-length grouping over the path arrays."""
    if not path:
        return []
    pos = np.fromiter((p.token_index for p in path), dtype=np.int64, count=len(path))
    frame = np.fromiter((p.time_index for p in path), dtype=np.int64, count=len(path))
    score = np.fromiter((p.score for p in path), dtype=np.float64, count=len(path))
    first = _run_starts(pos)
    last = np.concatenate((first[1:], [len(path)])) - 1
    mean = np.add.reduceat(score, first) / (last - first + 1)
    return [Segment(transcript[int(pos[a])], int(frame[a]), int(frame[b]) + 1, float(m)) for a, b, m in zip(first, last, mean)]


def _pooled_score(segments: Sequence[Segment]) -> float:
    """Duration-weighted mean score of a group of segments."""
    dur = np.array([g.length for g in segments], dtype=np.float64)
    return float(np.dot([g.score for g in segments], dur) / dur.sum())


def merge_words(segments: List[Segment], separator: str = " ") -> List[Segment]:
    """Character segments -> word segments: maximal runs without a separator label (:158-172)."""
    is_sep = np.array([g.label == separator for g in segments], dtype=bool)
    kept = np.flatnonzero(~is_sep)
    if len(kept) == 0:
        return []
    breaks = np.flatnonzero(np.diff(kept) > 1) + 1  # a gap in the kept indices = at least one separator in between
    words = []
    for run in np.split(kept, breaks):
        group = segments[int(run[0]):int(run[-1]) + 1]
        words.append(Segment("".join(g.label for g in group), group[0].start, group[-1].end, _pooled_score(group)))
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
        # the caller's own word list: word k owns the character segments [offset_k, offset_k + len(word_k)), the single
        # separator between words is skipped; timing and score come from its letters when it has any
        # (spaces / punctuation only count for a word made of nothing else)
        word_segments = []
        offset = 0
        for word in transcript_words:
            chars = char_segments[offset:offset + len(word)]
            offset += len(word) + 1
            assert "".join(c.label for c in chars) == word
            letters = [c for c in chars if c.label not in " " + _PUNCTUATION] or chars
            word_segments.append(Segment(word, letters[0].start, letters[-1].end, _pooled_score(letters)))
    return labels, emission, trellis, char_segments, word_segments
