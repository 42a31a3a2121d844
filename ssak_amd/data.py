"""Kaldi-folder ingest for the acoustic-model path (the callers' side of the boundary).

Re-states the on-disk API of the reference -- ``text`` + ``wav.scp`` mandatory, ``segments`` optional else
``utt2dur`` mandatory, list-files of folders with optional trailing weights, comma-separated folders, ``$VAR``
expansion, the wav.scp forms (plain path, ``sox PATH ... |``, ``flac ... PATH |``, quoted path with spaces)
(ssak/utils/dataset.py:165-360, ssak/utils/kaldi.py:8-37) -- and the collator conventions
(right zero padding, labels padded with -100: ssak/train/transformers/wav2vec_train.py:79-100).
Audio decode is limited to PCM WAV (the reference shells out to sox / torchaudio, absent here on purpose):
anything else raises ``RuntimeError`` like ``ssak/utils/audio.py:49-55``.
"""
from __future__ import annotations

import json
import os
import re
import wave
from dataclasses import dataclass
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np


# ----------------------------------------------------------------------------- wav.scp / folders
_QUOTED = re.compile(r"'([^']*)'")


def _wavscp_audio_path(entry: str) -> str:
    """The audio file a wav.scp entry (everything after the recording id) reads: a bare path, or a shell pipe whose reader
    is sox (input file = first argument) or flac (input file = last argument); a single-quoted path may hold spaces."""
    quoted = _QUOTED.search(entry)
    if quoted:
        return quoted.group(1)
    words = [w for w in entry.split() if w != "|"]
    if len(words) == 1:
        return words[0]
    reader = os.path.basename(words[0])
    if reader == "sox":
        return words[1]
    if reader == "flac":
        return words[-1]
    raise RuntimeError(f"Unknown wav.scp format with {words[0]}")


def parse_kaldi_wavscp(path: str) -> Dict[str, str]:
    """wav.scp -> {recording id: audio path}, environment variables expanded (semantics of ssak/utils/kaldi.py:8-37: fields
    are separated by ANY whitespace, tabs included; a line with a quote takes what stands between its first two quotes;
    pinned by the reference's own test folders, tests/golden/host_strings.json)."""
    table: Dict[str, str] = {}
    with open(path) as f:
        for raw in f:
            parts = raw.split(None, 1)
            if not parts:
                continue
            rec_id, entry = parts[0], (parts[1].strip() if len(parts) > 1 else "")
            if not entry:
                raise RuntimeError(f"wav.scp line without an audio entry: {raw.strip()!r} in {path}")
            audio = _wavscp_audio_path(entry)
            table[rec_id] = os.path.expandvars(audio) if "$" in audio else audio
    return table


@dataclass
class Utterance:
    id: str
    path: str
    start: float
    end: float
    text: str

    @property
    def duration(self) -> float:
        return self.end - self.start


def expand_kaldi_paths(kaldi_path, weight: float = 1.0) -> List[Tuple[str, float]]:
    """A folder, a comma-separated list of folders, or a list-file ``folder [weight] ...`` (dataset.py:165-198)."""
    if isinstance(kaldi_path, (list, tuple)):
        out = []
        for p in kaldi_path:
            out += expand_kaldi_paths(p, weight)
        return out
    if os.path.isfile(kaldi_path):
        out: List[Tuple[str, float]] = []
        with open(kaldi_path) as f:
            for line in f:
                for w in line.strip().split():
                    if "$" in w:
                        w = os.path.expandvars(w)
                    if os.path.isdir(w):
                        out.append((w, weight))
                    else:
                        try:
                            x = float(w)
                        except ValueError:
                            raise RuntimeError("Could not find folder %s" % w)
                        if not out:
                            raise AssertionError("File cannot start with a weight (first a folder name, then a weight)")
                        out[-1] = (out[-1][0], out[-1][1] * x)
        return out
    if not os.path.isdir(kaldi_path):
        if "," in kaldi_path:
            return expand_kaldi_paths(kaldi_path.split(","), weight)
        raise RuntimeError("Could not find folder %s" % kaldi_path)
    return [(kaldi_path, weight)]


def load_kaldi_folder(folder: str, min_duration: Optional[float] = None, max_duration: Optional[float] = None,
                      sort_by_len: int = 0) -> List[Utterance]:
    """One Kaldi folder -> utterances, with the duration filter / ordering of dataset.py:255-352."""
    for fname in ("text", "wav.scp"):
        if not os.path.isfile(os.path.join(folder, fname)):
            raise RuntimeError("Could not find file %s in folder %s" % (fname, folder))
    ids, texts = [], []
    with open(os.path.join(folder, "text"), encoding="utf8") as f:
        for line in f:
            r = line.strip().split(" ", 1)
            if not r or not r[0]:
                continue
            ids.append(r[0])
            texts.append(r[1] if len(r) > 1 else "")
    wav = parse_kaldi_wavscp(os.path.join(folder, "wav.scp"))
    utts: List[Utterance] = []
    seg_path = os.path.join(folder, "segments")
    if os.path.isfile(seg_path):
        segs = {}
        with open(seg_path) as f:
            for line in f:
                fl = line.strip().split()
                if len(fl) < 4:
                    continue
                st, en = float(fl[2]), float(fl[3])
                assert en - st > 0, f"Error in {folder}/segments:\nDuration of utterance {fl[0]} is negative: {en - st}"
                segs[fl[0]] = (fl[1], st, en)
        for i, t in zip(ids, texts):
            if i not in segs:
                continue
            wid, st, en = segs[i]
            d = en - st
            if (max_duration and d > max_duration) or (min_duration and d < min_duration):
                continue
            utts.append(Utterance(i, wav[wid], st, en, t))
    else:
        dur_path = os.path.join(folder, "utt2dur")
        if not os.path.isfile(dur_path):
            raise RuntimeError("Could not find file %s in folder %s" % ("utt2dur", folder))
        durs = {}
        with open(dur_path) as f:
            for line in f:
                if line.strip():
                    k, v = line.strip().split()[:2]
                    durs[k] = float(v)
        for i, t in zip(ids, texts):
            d = durs[i]
            if (max_duration and d > max_duration) or (min_duration and d < min_duration):
                continue
            utts.append(Utterance(i, wav[i], 0.0, d, t))
    if sort_by_len or min_duration or max_duration:
        utts.sort(key=lambda u: (u.duration, len(u.text)))
    if sort_by_len and sort_by_len < 0:
        utts.reverse()
    return utts


def load_kaldi(kaldi_path, min_duration=None, max_duration=None, sort_by_len: int = 0) -> List[Utterance]:
    """Folder(s) / list-file -> utterances; integer weights duplicate a folder's data (dataset.py:100-160)."""
    out: List[Utterance] = []
    for folder, w in expand_kaldi_paths(kaldi_path):
        u = load_kaldi_folder(folder, min_duration, max_duration, sort_by_len)
        reps = max(1, int(round(w)))
        out += u * reps
    return out


# ----------------------------------------------------------------------------- audio (PCM WAV only)
def load_audio(path: str, start: Optional[float] = None, end: Optional[float] = None, sample_rate: int = 16000) -> np.ndarray:
    """PCM WAV -> mono float32 in [-1,1); segment cut at ``int(start*sr)`` (ssak/utils/audio.py:85-92).
    Resampling / codecs are out of scope: any other rate or format raises RuntimeError."""
    if not os.path.isfile(path):
        raise RuntimeError(f"File {path} does not exist")
    try:
        with wave.open(path, "rb") as f:
            sr, nch, sw, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
            if f.getcomptype() != "NONE":
                raise RuntimeError(f"{path}: compressed WAV is not supported (PCM only)")
            s0 = int(start * sr) if start else 0
            s1 = min(n, int(end * sr)) if end else n
            f.setpos(min(s0, n))
            raw = f.readframes(max(0, s1 - s0))
    except (wave.Error, EOFError) as err:
        raise RuntimeError(f"Could not read {path} as PCM WAV (sox/ffmpeg decoding is not built): {err}") from err
    if sr != sample_rate:
        raise RuntimeError(f"{path}: sample rate {sr} != {sample_rate}; resampling is outside this path")
    if sw == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif sw == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise RuntimeError(f"{path}: unsupported sample width {sw}")
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)
    return x


def write_wav(path: str, x: np.ndarray, sample_rate: int = 16000):
    pcm = np.round(np.clip(x, -1, 1) * 32767.0).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sample_rate)
        f.writeframes(pcm.tobytes())


# ----------------------------------------------------------------------------- text <-> ids
_SPACES = re.compile(r"\s+")


def remove_special_words(text: str, glue_apostrophe: Optional[bool] = True) -> str:
    """Label clean-up applied before tokenisation (ssak/utils/text_basic.py:91-125): ``<...>`` words dropped, apostrophes
    glued to their neighbours (True), followed by a space (False) or left alone (None), whitespace collapsed."""
    if not text:
        return ""
    text = re.sub(r"<.*?>", "", text)
    if glue_apostrophe is True:
        text = re.sub(r"\s+'\s+", "'", text)
    elif glue_apostrophe is False:
        text = text.replace("'", "' ")
    return _SPACES.sub(" ", text).strip()


class CharTokenizer:
    """Character CTC tokenizer with the conventions of ``Wav2Vec2CTCTokenizer``: ``|`` is the word delimiter,
    pad token = CTC blank, decoding merges repeats, drops pad, maps the delimiter to a space."""

    def __init__(self, vocab: Sequence[str], pad_token="<pad>", unk_token="<unk>", word_delimiter_token="|"):
        self.vocab = list(vocab)
        self.index = {t: i for i, t in enumerate(self.vocab)}
        self.pad_token_id = self.index[pad_token]
        self.unk_token_id = self.index.get(unk_token, self.pad_token_id)
        self.delim = word_delimiter_token
        self.special = {t for t in self.vocab if t.startswith("<") and t.endswith(">")}

    @classmethod
    def from_vocab_json(cls, path: str, **kw) -> "CharTokenizer":
        with open(path) as f:
            d = json.load(f)
        return cls([t for t, _ in sorted(d.items(), key=lambda kv: kv[1])], **kw)

    def save(self, folder: str):
        with open(os.path.join(folder, "vocab.json"), "w") as f:
            json.dump(self.index, f, ensure_ascii=False)

    def __len__(self):
        return len(self.vocab)

    def encode(self, text: str) -> List[int]:
        return [self.index.get(self.delim if c == " " else c, self.unk_token_id) for c in text]

    def decode(self, ids: Iterable[int], group_tokens: bool = True) -> str:
        toks, prev = [], None
        for i in ids:
            i = int(i)
            if i < 0:
                continue
            if group_tokens and i == prev:
                continue
            prev = i
            if i == self.pad_token_id or self.vocab[i] in self.special:
                continue
            toks.append(" " if self.vocab[i] == self.delim else self.vocab[i])
        return _SPACES.sub(" ", "".join(toks)).strip()


# ----------------------------------------------------------------------------- batching / collation
def pad_waves(waves: Sequence[np.ndarray], target_len: int | None = None) -> Tuple[np.ndarray, np.ndarray]:
    """Right zero padding to the longest (processor.pad, wav2vec_train.py:79-85) -> ([B,T] float32, lengths).  ``target_len``:
    pad to this length instead (a rank's shard of a global batch is padded like the WHOLE batch: the reference's collator pads the
    global batch before DataParallel scatters it, and an unmasked group-norm model's logits depend on the padding)."""
    lens = np.array([len(w) for w in waves], dtype=np.int32)
    out = np.zeros((len(waves), max(int(lens.max()), int(target_len or 0))), dtype=np.float32)
    for i, w in enumerate(waves):
        out[i, :len(w)] = w
    return out, lens


def pad_labels(label_lists: Sequence[Sequence[int]], pad_value: int = -100) -> np.ndarray:
    """Label padding replaced by -100 (wav2vec_train.py:89-100)."""
    L = max((len(l) for l in label_lists), default=0)
    out = np.full((len(label_lists), max(L, 1)), pad_value, dtype=np.int64)
    for i, l in enumerate(label_lists):
        out[i, :len(l)] = np.asarray(l, dtype=np.int64)
    return out


def length_grouped_batches(lengths: Sequence[float], batch_size: int, rng: np.random.RandomState,
                           mega_factor: int = 50, frame_budget: Optional[float] = None, max_batch: Optional[int] = None) -> List[List[int]]:
    """``group_by_length`` batching (wav2vec_train.py:355 -> HF LengthGroupedSampler, trainer.py:758-775):
    shuffle, cut into mega-batches of ``mega_factor * batch_size``, sort each by length (longest first), slice.

    ``frame_budget`` (an extension; None = the reference's constant-COUNT slices): slice each sorted mega-batch so that every batch
    holds about the same PADDED length instead -- count x longest utterance <= ``frame_budget`` (in the unit of ``lengths``), at
    least one and at most ``max_batch`` (default 8 x batch_size) utterances -- i.e. many short utterances or few long ones per
    step.  With mixed 1-15 s audio a constant count leaves the short batches with a fraction of the work of the long ones
    (16 x 1.5 s = 1 200 frames: 19 output tiles for 256 CUs); a constant frame budget keeps every step's products the same size.
    The same shuffle / mega-batch / sort as the count mode, so one epoch still visits every utterance exactly once."""
    idx = rng.permutation(len(lengths))
    mega = mega_factor * batch_size
    out = []
    cap = max_batch or 8 * batch_size
    for i in range(0, len(idx), mega):
        chunk = sorted(idx[i:i + mega].tolist(), key=lambda j: -lengths[j])
        if frame_budget is None:
            out += [chunk[k:k + batch_size] for k in range(0, len(chunk), batch_size)]
            continue
        k = 0
        while k < len(chunk):
            n = int(max(1, min(cap, frame_budget // max(lengths[chunk[k]], 1))))  # (sorted: the first one is the longest)
            out.append(chunk[k:k + n])
            k += n
    return out


def shard_batch(indices: Sequence[int], rank: int, world: int) -> List[int]:
    """Contiguous shard of a (length-sorted) global batch for one data-parallel rank
    (``per_device_train_batch_size = batch_size // num_devices``, wav2vec_train.py:349,356).  Nothing is dropped: HF's default
    ``dataloader_drop_last=False`` (docker/transformers_modified/trainer.py:834) trains on the short last batch of an epoch,
    so when the batch does not divide by the world size the first ``len % world`` ranks take one utterance more and the
    trainer weights each rank's gradient by its utterance count; a rank may get an empty shard."""
    n = len(indices)
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return list(indices[start:start + base + (1 if rank < extra else 0)])


def to_audio_batches(inputs, batch_size: int = 1, sort_by_len: bool = False, output_ids: bool = False,
                     sample_rate: int = 16000) -> Iterator[list]:
    """Audio file(s) / Kaldi folder(s) -> batches of float32 arrays (optionally with ids), as
    ssak/utils/dataset.py:647-752 feeds the inference loop."""
    if isinstance(inputs, str):
        inputs = [inputs]
    items: List[Tuple[np.ndarray, str]] = []
    for inp in inputs:
        if os.path.isdir(inp) or (os.path.isfile(inp) and not inp.lower().endswith((".wav",))):
            for u in load_kaldi(inp, sort_by_len=-1 if sort_by_len else 0):
                items.append((load_audio(u.path, u.start, u.end, sample_rate), u.id))
        else:
            items.append((load_audio(inp, sample_rate=sample_rate), os.path.basename(inp)))
    if sort_by_len:
        items.sort(key=lambda it: -len(it[0]))
    for i in range(0, len(items), batch_size):
        chunk = items[i:i + batch_size]
        yield [(a, k) for a, k in chunk] if output_ids else [a for a, _ in chunk]
