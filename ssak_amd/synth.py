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
