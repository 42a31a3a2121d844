"""Synthetic benchmark inputs of BASELINE.md / SURVEY.md section 8d: 16 kHz utterances of Gaussian noise
(sigma 0.1) plus 220/440/880 Hz sinusoids (amp 0.05), clipped to [-1,1] and quantised to PCM16; transcripts are
seeded random strings over a-z + space with length uniform in [60,120]; 32-symbol vocabulary, <pad> = 0 = blank."""
from __future__ import annotations

import numpy as np

VOCAB = ["<pad>", "<s>", "</s>", "<unk>", "|"] + [chr(ord("a") + i) for i in range(26)] + ["'"]
CHAR2ID = {c: i for i, c in enumerate(VOCAB)}


def synth_wave(rng: np.random.Generator, n: int) -> np.ndarray:
    t = np.arange(n) / 16000.0
    x = rng.standard_normal(n) * 0.1
    for f in (220.0, 440.0, 880.0):
        x = x + 0.05 * np.sin(2 * np.pi * f * t + rng.uniform(0, 2 * np.pi))
    pcm = np.round(np.clip(x, -1, 1) * 32767.0).astype(np.int16)
    return pcm.astype(np.float32) / 32768.0


def synth_text(rng: np.random.Generator, lo: int = 60, hi: int = 120) -> str:
    n = int(rng.integers(lo, hi + 1))
    alphabet = [chr(ord("a") + i) for i in range(26)] + [" "]
    return "".join(alphabet[i] for i in rng.integers(0, 27, n))


def text_to_ids(text: str):
    return [CHAR2ID["|"] if c == " " else CHAR2ID.get(c, CHAR2ID["<unk>"]) for c in text]


def synth_batch(B: int, n_samples: int = 160000, seed: int = 1234):
    """(waves [B,T] float32 raw, labels [B,Lmax] int64 padded with -100)."""
    rng = np.random.default_rng(seed)
    waves = np.stack([synth_wave(rng, n_samples) for _ in range(B)])
    ids = [text_to_ids(synth_text(rng)) for _ in range(B)]
    L = max(len(i) for i in ids)
    labels = np.full((B, L), -100, dtype=np.int64)
    for b, i in enumerate(ids):
        labels[b, :len(i)] = i
    return waves, labels


def write_kaldi_folder(folder: str, n_utts: int, seconds: float = 10.0, seed: int = 1234, threads: int = 8):
    """SURVEY.md section 8d's synthetic input as files: a Kaldi folder of ``n_utts`` PCM16 mono 16 kHz WAV files of exactly
    ``seconds`` (wav.scp, text, utt2dur; no segments).  Same signal model as :func:`synth_wave` (Gaussian noise sigma 0.1 + 220 /
    440 / 880 Hz sinusoids of amplitude 0.05 with a random phase, clipped, PCM16), generated per file from its own seeded stream;
    the sinusoids come from six shared tables by the angle-addition formula so that 4 096 files take seconds, not minutes."""
    import os
    import wave
    from concurrent.futures import ThreadPoolExecutor
    n = int(round(seconds * 16000))
    t = np.arange(n) / 16000.0
    tabs = [(np.sin(2 * np.pi * f * t).astype(np.float32), np.cos(2 * np.pi * f * t).astype(np.float32)) for f in (220.0, 440.0, 880.0)]
    os.makedirs(os.path.join(folder, "audio"), exist_ok=True)

    def one(i):
        rng = np.random.default_rng([seed, i])
        x = rng.standard_normal(n, dtype=np.float32) * np.float32(0.1)
        for sn, cs in tabs:
            ph = rng.uniform(0, 2 * np.pi)
            x += np.float32(0.05 * np.cos(ph)) * sn + np.float32(0.05 * np.sin(ph)) * cs
        pcm = np.round(np.clip(x, -1, 1) * 32767.0).astype("<i2")
        path = os.path.join(folder, "audio", f"utt{i:05d}.wav")
        with wave.open(path, "wb") as f:
            f.setnchannels(1)
            f.setsampwidth(2)
            f.setframerate(16000)
            f.writeframes(pcm.tobytes())
        return path, synth_text(rng)

    with ThreadPoolExecutor(max(1, threads)) as pool:
        rows = list(pool.map(one, range(n_utts)))
    with open(os.path.join(folder, "wav.scp"), "w") as fw, open(os.path.join(folder, "text"), "w") as ft, open(os.path.join(folder, "utt2dur"), "w") as fd:
        for i, (path, text) in enumerate(rows):
            fw.write(f"utt{i:05d} {path}\n")
            ft.write(f"utt{i:05d} {text}\n")
            fd.write(f"utt{i:05d} {seconds:.3f}\n")
    return rows
