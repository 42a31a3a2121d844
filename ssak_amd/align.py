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


def forced_align_batch(emissions: Sequence[torch.Tensor], tokens_list: Sequence[Sequence[int]], blank_id: int = 0,
                       first_as_garbage: bool = False, want_trellis: bool = False, device=None):
    """Many utterances in ONE launch (``ssak_ctc_forced_align_batch``: one workgroup per utterance).  ``emissions[i]`` is
    [F_i, V] float32 log-probabilities, ``tokens_list[i]`` its token ids.  Returns a list of (trellis [F_i+1, L_i+1] or None,
    path or None) -- every entry identical to what :func:`forced_align` gives for that utterance alone."""
    n = len(emissions)
    assert n == len(tokens_list) and n > 0
    dev = torch.device(device) if device is not None else (emissions[0].device if emissions[0].is_cuda else torch.device("cuda:0"))
    V = int(emissions[0].shape[1])
    Fs = [int(e.shape[0]) for e in emissions]
    Ls = [len(t) for t in tokens_list]
    if min(Ls) == 0:
        raise IndexError("forced alignment needs a non-empty transcript")
    Fmax, Lmax = max(Fs), max(Ls)
    em = torch.zeros((n, Fmax, V), dtype=torch.float32)
    tok = torch.zeros((n, Lmax), dtype=torch.int32)
    col0 = torch.zeros((n, Fmax + 1), dtype=torch.float32) if first_as_garbage else None
    for i, (e, t) in enumerate(zip(emissions, tokens_list)):
        e = e.detach().to(device="cpu", dtype=torch.float32)
        assert e.shape[1] == V
        em[i, :Fs[i]] = e
        th = torch.as_tensor(list(t), dtype=torch.int32)
        if int(th.min()) < 0 or int(th.max()) >= V:
            raise IndexError(f"token id outside the emission's {V} classes")
        tok[i, :Ls[i]] = th
        if first_as_garbage:  # column 0 of the garbage variant (:38), with the reference's torch ops
            col0[i, 1:Fs[i] + 1] = (1 - e[:, int(th[0])].exp()).log()
    em_d, tok_d = em.to(dev), tok.to(dev)
    fl = torch.tensor(Fs, dtype=torch.int32, device=dev)
    tl = torch.tensor(Ls, dtype=torch.int32, device=dev)
    col0_d = col0.to(dev) if col0 is not None else None
    trellis = torch.empty((n, Fmax + 1, Lmax + 1), dtype=torch.float32, device=dev) if want_trellis else None
    path_token = torch.empty((n, Fmax), dtype=torch.int32, device=dev)
    path_logp = torch.empty((n, Fmax), dtype=torch.float32, device=dev)
    info = torch.empty((n, 2), dtype=torch.int32, device=dev)
    ws = torch.empty(hip.lib.ssak_ctc_align_batch_workspace_bytes(n, Fmax, Lmax), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        hip.check(hip.lib.ssak_ctc_forced_align_batch(hip.ptr(em_d), hip.ptr(fl), hip.ptr(tok_d), hip.ptr(tl), n, Fmax, V, Lmax,
                                                      int(blank_id), hip.ptr(col0_d), hip.ptr(trellis), hip.ptr(path_token),
                                                      hip.ptr(path_logp), hip.ptr(info), hip.ptr(ws), ws.numel(), hip.stream()))
    info_h = info.cpu().numpy()
    pt_h, pl_h = path_token.cpu().numpy(), path_logp.cpu().numpy()
    out = []
    for i in range(n):
        cnt, first = int(info_h[i, 0]), int(info_h[i, 1])
        path = None
        if cnt >= 0:
            sc = np.exp(pl_h[i, first:first + cnt])
            path = [Point(int(pt_h[i, first + k]), first + k, float(sc[k])) for k in range(cnt)]
        out.append((trellis[i, :Fs[i] + 1, :Ls[i] + 1] if want_trellis else None, path))
    return out


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


def _transcript_tokens(transcript, labels, blank_id, add_before_after=None):
    """(characters, words or None, token ids, labels cut to the emission's classes) of compute_alignment (:318-356)."""
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
    dictionary = {c: i for i, c in enumerate(labels)}
    tokens = [loose_get_char_index(dictionary, c, space_id) for c in transcript_characters]
    tokens = [i for i in tokens if i is not None]
    return transcript_characters, transcript_words, tokens


def _segments_from_path(path, transcript_characters, transcript_words, add_before_after=None):
    """(char_segments, word_segments) from the best path (:364-402)."""
    char_segments = merge_repeats(transcript_characters, path)
    if add_before_after:
        assert char_segments[0].label == add_before_after and char_segments[-1].label == add_before_after
        char_segments = char_segments[1:-1]
    if transcript_words is None:
        return char_segments, merge_words(char_segments)
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
    return char_segments, word_segments


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
    labels = labels[:emission.shape[1]]
    transcript_characters, transcript_words, tokens = _transcript_tokens(transcript, labels, blank_id, add_before_after)
    trellis = get_trellis(emission, tokens, blank_id=blank_id, first_as_garbage=first_as_garbage)
    path = backtrack(trellis, emission, tokens, blank_id=blank_id)
    char_segments, word_segments = _segments_from_path(path, transcript_characters, transcript_words, add_before_after)
    if add_before_after:
        trellis = trellis[:, [0] + list(range(2, trellis.shape[1] - 1))]
    return labels, emission, trellis, char_segments, word_segments


def compute_alignment_batch(audios, transcripts, model, first_as_garbage=False, emissions=None):
    """compute_alignment for many utterances with ONE alignment launch.  Returns, per utterance, either
    (num_frames, char_segments, word_segments) or the exception compute_alignment would have raised for it (the caller of the
    reference's loop catches per utterance, tools/align_audio_transcript.py:337-346).

    Emissions are computed per utterance for group-norm ("base") models -- those run without attention mask, so padding an
    utterance inside a batch would change its own logits -- and may be supplied by the caller (``emissions``)."""
    from .infer import compute_log_probas
    labels, blank_id = get_model_vocab(model)
    if emissions is None:
        emissions = [compute_log_probas(model, a) for a in audios]
    results = [None] * len(transcripts)
    jobs = []
    for i, (em, tr) in enumerate(zip(emissions, transcripts)):
        try:
            lab = labels[:em.shape[1]]
            chars, words, tokens = _transcript_tokens(tr, lab, blank_id)
            if not tokens:
                raise IndexError("forced alignment needs a non-empty transcript")
            jobs.append((i, chars, words, tokens))
        except Exception as err:  # noqa: BLE001 -- reported per utterance, like the reference's loop
            results[i] = err
    if jobs:
        aligned = forced_align_batch([emissions[i] for i, *_ in jobs], [t for *_, t in jobs], blank_id, first_as_garbage)
        for (i, chars, words, _), (_, path) in zip(jobs, aligned):
            try:
                if path is None:
                    raise RuntimeError("Failed to align (not enough tokens for the duration?)")
                cs, wsg = _segments_from_path(path, chars, words)
                results[i] = (int(emissions[i].shape[0]), cs, wsg)
            except Exception as err:  # noqa: BLE001
                results[i] = err
    return results
