"""CPU restatement of the per-utterance loop of the reference's ``tools/align_audio_transcript.py``.  TEST INFRASTRUCTURE ONLY.

``split_long_audio_kaldifolder`` (:121-443) walks the utterances of a Kaldi folder one by one; an utterance is copied
(:296-302), dropped, or aligned (``compute_alignment``, ssak/utils/align_transcriptions.py:294-402) and cut at word boundaries
(:348-437).  This file restates that walk for ONE already-planned utterance at a time, the way the reference does it -- one
alignment per call, the trellis / backtrack / merge of ``oracle/align_ref.py`` (bit-pinned to the reference's torch ops by
tests/golden/align.npz) -- so that the batched device tool (``ssak_amd/tools/align_audio_transcript.py``: ONE kernel launch per
batch of utterances) can be checked for bit-exact cut points on the same emissions.  Parity with the reference's own
end-to-end goldens (tests/expected/align_audio_transcript/*) needs downloaded weights: unpinned.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from . import align_ref as AR

PUNCTUATION = "".join(c for c in __import__("string").punctuation + "。，！？：”、…" + "؟،؛" + "—" + "«°»×‹›•“–‘″‘" if c not in "-'")


def word_segments_from_words(emission: np.ndarray, spoken: Sequence[str], labels: Sequence[str], blank_id: int, first_as_garbage: bool):
    """compute_alignment (:318-402) for a transcript given as a LIST of words: characters = words joined by " ", unknown
    characters -> the space label (loose_get_char_index without the transliteration table), trellis + backtrack +
    merge_repeats, then one word segment per given word from its letters (punctuation / spaces only count for a word made of
    nothing else), score = duration-weighted mean of the letters' scores."""
    chars = " ".join(spoken)
    labels = list(labels[:emission.shape[1]])
    table = {c: i for i, c in enumerate(labels)}
    space_id = labels.index(" ") if " " in labels else blank_id

    def index(c):
        for v in (c, c.lower(), c.upper()):
            if v in table:
                return table[v]
        return space_id

    tokens = [index(c) for c in chars]
    trellis = AR.get_trellis(emission, tokens, blank_id, first_as_garbage)
    path = AR.backtrack(trellis, emission, tokens, blank_id)
    char_segments = AR.merge_repeats(chars, path)
    words, off = [], 0
    for w in spoken:
        cs = char_segments[off:off + len(w)]
        off += len(w) + 1
        assert "".join(c.label for c in cs) == w
        letters = [c for c in cs if c.label not in " " + PUNCTUATION] or cs
        dur = np.array([c.end - c.start for c in letters], dtype=np.float64)
        score = float(np.dot([c.score for c in letters], dur) / dur.sum())
        words.append(AR.Segment(w, letters[0].start, letters[-1].end, score))
    return emission.shape[0], char_segments, words


def cut_lines(uid: str, wavid: str, spk: str, start: float, words: Sequence[str], word_segments, num_frames: int, audio_len: int,
              sample_rate: int, max_duration: float, refine_timestamps, skip_warnings: bool = False):
    """The cutting loop (:395-437 with add_segment :348-393) -> lines for (text, utt2spk, utt2dur, segments), formatted as the
    reference writes them."""
    out = {"text": [], "utt2spk": [], "utt2dur": [], "segments": []}
    ratio = audio_len / (num_frames * sample_rate)
    state = {"index": 1, "first": 0.0, "last": 0.0, "text": ""}

    def add_segment():
        new_id = f"{uid}_cut{state['index']:02}"
        state["index"] += 1
        new_start, new_end = start + state["first"], start + state["last"]
        long = new_end - new_start > max_duration
        if state["last"] <= state["first"]:
            pass  # skipped: null or negative duration
        elif long and skip_warnings:
            pass  # skipped: too long
        else:
            out["text"].append(f"{new_id} {state['text']}\n")
            out["utt2spk"].append(f"{new_id} {spk}\n")
            out["utt2dur"].append(f"{new_id} {new_end - new_start:.3f}\n")
            out["segments"].append(f"{new_id} {wavid} {new_start:.3f} {new_end:.3f}\n")
        state["first"] = state["last"]
        state["text"] = ""

    segs = [AR.Segment(s.label, s.start, s.end, s.score) for s in word_segments]
    assert len(segs) == len(words)
    if len(segs) and not refine_timestamps:
        segs[0].start = 0
        segs[-1].end = num_frames
    for i, (segment, word) in enumerate(zip(segs, words)):
        if word.strip() in PUNCTUATION:
            segment.end = segment.start
        if refine_timestamps and i == 0:
            state["first"] = state["last"] = segment.start * ratio
        end = segment.end * ratio
        if end - state["first"] > max_duration and state["text"]:
            add_segment()
        state["last"] = end
        if state["text"]:
            state["text"] += " "
        state["text"] += word
    if state["text"]:
        state["last"] = segs[-1].end * ratio
        add_segment()
    return out
