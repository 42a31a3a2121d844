"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the audio conforming of the reference,
ssak/utils/audio.py:101-154 (``conform_audio``): channel average (librosa.to_mono = mean over channels, :118) and
``torchaudio.transforms.Resample(sr, sample_rate)`` (:134) with torchaudio's defaults -- resampling_method
"sinc_interp_hann", lowpass_filter_width 6, rolloff 0.99.

torchaudio is a third-party dependency of the reference (requirements.txt:40, unpinned) that is NOT installed here, so
its published algorithm (torchaudio/functional/functional.py, ``_get_sinc_resample_kernel`` and
``_apply_sinc_resample_kernel``, v2.x) is restated with the same torch ops (float64 index grid, clamp, cos^2 window,
sin(t)/t, ``conv1d`` with stride = reduced source rate).  **Parity unpinned** against torchaudio itself; what is checked
(tests/test_oracle.py) are the algorithm's defining properties: identity for equal rates, the output length
ceil(new * n / orig), unit DC gain, and a sinusoid below the new Nyquist keeping its amplitude, phase and frequency."""
from __future__ import annotations

import math

import numpy as np
import torch


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1)[:, None, None] / new + idx
    t *= base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0).to(t), t.sin() / t)
    kernels *= window * scale
    return kernels.to(torch.float32), width, orig, new


def resample(waveform: np.ndarray, orig_freq: int, new_freq: int) -> np.ndarray:
    """1-D float32 waveform -> resampled float32 (torchaudio.functional.resample semantics)."""
    x = torch.as_tensor(np.asarray(waveform, dtype=np.float32))
    if int(orig_freq) == int(new_freq):
        return x.numpy()
    kernel, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
    length = x.shape[-1]
    xp = torch.nn.functional.pad(x[None, None], (width, width + orig))
    y = torch.nn.functional.conv1d(xp, kernel, stride=orig)  # [1, new, frames]
    y = y.transpose(1, 2).reshape(-1)
    return y[:math.ceil(new * length / orig)].numpy()


def to_mono(frames: np.ndarray) -> np.ndarray:
    """[n, channels] -> [n] mean over channels (librosa.to_mono on the transposed array, audio.py:117-118)."""
    frames = np.asarray(frames, dtype=np.float32)
    return frames if frames.ndim == 1 else frames.mean(axis=1, dtype=np.float32)


def pcm16_to_float(raw: bytes, channels: int) -> np.ndarray:
    x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    return x.reshape(-1, channels) if channels > 1 else x
