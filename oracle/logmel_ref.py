"""float64 numpy restatement of the Whisper log-mel front end.  TEST INFRASTRUCTURE ONLY.

Reference call site: ``ssak/utils/dataset.py:632-637`` with a Whisper processor (as set up by
``ssak/train/transformers/whisper_train.py:356-367``) -> ``WhisperFeatureExtractor``
(un-vendored ``transformers``, feature_extraction_whisper.py:95-133 numpy path, :135-168 torch
path; filters from ``audio_utils.mel_filter_bank`` with Slaney scale + Slaney norm).
Pinned against ``transformers.WhisperFeatureExtractor`` by ``oracle/gen_golden.py`` ->
``tests/golden/logmel.npz``.
"""
from __future__ import annotations

import numpy as np

SR, N_FFT, HOP, N_MELS, N_SAMPLES = 16000, 400, 160, 80, 480000


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    lin = 3.0 * f / 200.0
    logstep = 27.0 / np.log(6.4)
    with np.errstate(divide="ignore", invalid="ignore"):
        lg = 15.0 + np.log(f / 1000.0) * logstep
    return np.where(f >= 1000.0, lg, lin)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    lin = 200.0 * m / 3.0
    logstep = np.log(6.4) / 27.0
    lg = 1000.0 * np.exp(logstep * (m - 15.0))
    return np.where(m >= 15.0, lg, lin)


def mel_filters(n_mels: int = N_MELS, n_fft: int = N_FFT, sr: int = SR, fmin=0.0, fmax=8000.0) -> np.ndarray:
    """[n_fft//2+1, n_mels] triangular Slaney filters, Slaney (area) normalised."""
    nb = n_fft // 2 + 1
    fft_freqs = np.linspace(0, sr // 2, nb)
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    f = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(f)
    slopes = f[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    fb *= (2.0 / (f[2:n_mels + 2] - f[:n_mels]))[None, :]
    return fb


def hann_periodic(n: int = N_FFT) -> np.ndarray:
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def log_mel(wave: np.ndarray, n_samples: int = N_SAMPLES) -> np.ndarray:
    """One utterance -> [80, n_samples//160] float32: pad/trim to ``n_samples``, centred
    reflect-padded STFT (n_fft 400, hop 160, periodic Hann), |.|^2, drop last frame, mel,
    log10(max(.,1e-10)), max(x, max(x)-8), (x+4)/4."""
    x = np.zeros(n_samples, dtype=np.float64)
    n = min(len(wave), n_samples)
    x[:n] = np.asarray(wave[:n], dtype=np.float64)
    xp = np.pad(x, (N_FFT // 2, N_FFT // 2), mode="reflect")
    nfr = 1 + (len(xp) - N_FFT) // HOP
    idx = np.arange(N_FFT)[None, :] + HOP * np.arange(nfr)[:, None]
    spec = np.fft.rfft(xp[idx] * hann_periodic()[None, :], axis=1)
    power = (spec.real ** 2 + spec.imag ** 2)[:-1]  # drop last frame
    mel = np.maximum(power @ mel_filters(), 1e-10)
    lg = np.log10(mel).T
    lg = np.maximum(lg, lg.max() - 8.0)
    return ((lg + 4.0) / 4.0).astype(np.float32)
