"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's CTC forced alignment.

Follows ssak/utils/align_transcriptions.py with its module constants USE_MAX = True, USE_CHAR_REPEATED = True (:24-25):
``get_trellis`` (:27-70), ``backtrack`` (:79-123), ``merge_repeats`` (:140-156), ``merge_words`` (:158-172).  The
reference module itself cannot be imported here (ModuleNotFoundError: num2words, an ordinary missing dependency), so the
restatement is pinned by (a) the same third-party torch ops the reference calls (``torch.cumsum`` / ``torch.maximum``,
tests/golden/align.npz written by oracle/gen_golden_align.py), (b) exhaustive enumeration of all alignments on tiny
cases (tests/test_oracle.py) -- parity with the reference's own end-to-end goldens
(tests/expected/align_audio_transcript/*) needs pretrained weights and is unpinned.

numpy float32 throughout; column 0 is accumulated in float64 and rounded per element, which is what ``torch.cumsum``
does for a float32 CPU tensor (accumulation type double).
"""
from __future__ import annotations

import dataclasses
from typing import List, Sequence

import numpy as np


def get_trellis(emission: np.ndarray, tokens: Sequence[int], blank_id: int = 0, first_as_garbage: bool = False) -> np.ndarray:
    """emission [F, V] float32 log-probabilities -> trellis [F+1, L+1] float32 (align_transcriptions.py:27-70)."""
    emission = np.asarray(emission, dtype=np.float32)
    tokens = np.asarray(tokens, dtype=np.int64)
    F, L = emission.shape[0], len(tokens)
    trellis = np.empty((F + 1, L + 1), dtype=np.float32)
    trellis[0, 0] = 0
    if first_as_garbage:
        import torch  # the reference's own float32 exp / log (numpy's differ in the last ulp)
        trellis[1:, 0] = (1 - torch.from_numpy(np.ascontiguousarray(emission[:, tokens[0]])).exp()).log().numpy()  # :38
    else:
        trellis[1:, 0] = np.cumsum(emission[:, blank_id].astype(np.float64)).astype(np.float32)  # :40
    if L > 0:
        trellis[0, -L:] = -np.inf                                                              # :42
        trellis[-L:, 0] = np.inf                                                               # :43
    e_tok = emission[:, tokens]  # [F, L]
    for t in range(F):                                                                         # :45-53
        stay_blank = trellis[t, 1:] + emission[t, blank_id]
        stay_tok = trellis[t, 1:] + e_tok[t]
        change = trellis[t, :-1] + e_tok[t]
        trellis[t + 1, 1:] = np.maximum(stay_blank, np.maximum(stay_tok, change))
    return trellis


@dataclasses.dataclass
class Point:
    token_index: int
    time_index: int
    score: float


def backtrack_raw(trellis: np.ndarray, emission: np.ndarray, tokens: Sequence[int], blank_id: int = 0):
    """The path of align_transcriptions.py:79-123 as (token_index, time_index, log-probability) triples, oldest first;
    the reference's ``Point.score`` is exp() of the third.  Raises RuntimeError as the reference does."""
    emission = np.asarray(emission, dtype=np.float32)
    F = emission.shape[0]
    j = trellis.shape[1] - 1
    t_start = int(np.argmax(trellis[:, j]))                                                    # :88
    out = []
    done = False
    for t in range(t_start, 0, -1):
        tok = tokens[j - 1]
        stayed = np.maximum(trellis[t - 1, j] + emission[t - 1, blank_id], trellis[t - 1, j] + emission[t - 1, tok])  # :96-99
        changed = trellis[t - 1, j - 1] + emission[t - 1, tok]                                 # :103
        if changed < stayed and t < F:                                                         # :106-108
            logp = np.maximum(emission[t - 1, 0], emission[t, tok])
        else:
            logp = emission[t - 1, tok if changed > stayed else 0]                             # :112
        out.append((j - 1, t - 1, np.float32(logp)))
        if changed > stayed:                                                                   # :117-120
            j -= 1
            if j == 0:
                done = True
                break
    if not done:
        raise RuntimeError("Failed to align (not enough tokens for the duration?)")            # :122
    return out[::-1]


def backtrack(trellis, emission, tokens, blank_id: int = 0) -> List[Point]:
    return [Point(j, t, float(np.exp(np.float32(lp)))) for j, t, lp in backtrack_raw(trellis, emission, tokens, blank_id)]


@dataclasses.dataclass
class Segment:
    label: str
    start: int
    end: int
    score: float

    @property
    def length(self):
        return self.end - self.start


def merge_repeats(transcript, path: List[Point]) -> List[Segment]:
    """align_transcriptions.py:140-156."""
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
    """align_transcriptions.py:158-172."""
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


def best_score_by_enumeration(emission: np.ndarray, tokens: Sequence[int], blank_id: int = 0) -> float:
    """max over every frame labelling the trellis recursion admits, of the summed log-probabilities, for the LAST
    column at time F (tiny cases only).  A path spends t0 >= 0 frames on blank before the first token (column 0 is the
    running sum of blank), then each frame either enters the next token (emits it), or stays on the current token emitting
    it again or emitting blank.  Float64, so compare with a tolerance."""
    e = np.asarray(emission, dtype=np.float64)
    F, L = e.shape[0], len(tokens)
    best = -np.inf

    def rec(t, j, score):
        nonlocal best
        if t == F:
            if j == L:
                best = max(best, score)
            return
        if j == 0:
            rec(t + 1, 0, score + e[t, blank_id])
        else:
            rec(t + 1, j, score + max(e[t, blank_id], e[t, tokens[j - 1]]))
        if j < L:
            rec(t + 1, j + 1, score + e[t, tokens[j]])

    rec(0, 0, 0.0)
    return best
