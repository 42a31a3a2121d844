"""float32 numpy restatement of the ALGORITHM of the device log-mel kernel (ssak_amd/csrc/logmel.hip, round 5): the 400-point
real STFT as a 200-point complex mixed-radix Stockham FFT (passes 5, 5, 8) plus the real-input split.  TEST INFRASTRUCTURE ONLY.

It exists to bound the fp32 error of that route against the float64 oracle (`logmel_ref.log_mel`, itself pinned against
``transformers.WhisperFeatureExtractor`` by tests/golden/logmel.npz) on the CPU: tests/test_oracle.py.  The kernel is checked against
the same oracle on the GPU (tests/test_gpu_ops.py).  Reference call site: ssak/utils/dataset.py:632-637 with a Whisper processor.
"""
from __future__ import annotations

import numpy as np

from . import logmel_ref as R

F32, C64 = np.float32, np.complex64
NZ = R.N_FFT // 2
RADICES = (5, 5, 8)


def stockham_fft200(z: np.ndarray) -> np.ndarray:
    """z [frames, 200] complex64 -> its DFT, every operation rounded to float32.  Pass with radix r and Ns = product of the earlier
    radices: butterfly j < 200 / r reads z[j + q * 200 / r] * exp(-2 pi i q (j mod Ns) / (Ns r)), q < r, and writes its r-point DFT to
    (j div Ns) Ns r + (j mod Ns) + p Ns."""
    a = z.astype(C64)
    ns = 1
    for r in RADICES:
        nb = NZ // r
        j = np.arange(nb)
        k = j % ns
        v = [a[:, j + q * nb] for q in range(r)]
        for q in range(1, r):
            ang = -2.0 * np.pi * q * k / (ns * r)
            v[q] = (v[q] * (np.cos(ang) + 1j * np.sin(ang)).astype(C64)[None, :]).astype(C64)
        w = np.exp(-2j * np.pi * np.outer(np.arange(r), np.arange(r)) / r).astype(C64)
        out = np.zeros_like(a)
        base = (j // ns) * ns * r + k
        for p in range(r):
            acc = np.zeros_like(v[0])
            for q in range(r):
                acc = (acc + v[q] * w[p, q]).astype(C64)
            out[:, base + p * ns] = acc
        a = out
        ns *= r
    return a


def log_mel_fp32_fft(wave: np.ndarray, n_samples: int = R.N_SAMPLES) -> np.ndarray:
    """The same function as ``logmel_ref.log_mel`` by the kernel's route, in float32."""
    x = np.zeros(n_samples, F32)
    n = min(len(wave), n_samples)
    x[:n] = wave[:n]
    xp = np.pad(x, (R.N_FFT // 2, R.N_FFT // 2), mode="reflect")
    nfr = 1 + (len(xp) - R.N_FFT) // R.HOP
    idx = np.arange(R.N_FFT)[None, :] + R.HOP * np.arange(nfr)[:, None]
    y = (xp[idx] * R.hann_periodic().astype(F32)[None, :]).astype(F32)
    zf = stockham_fft200((y[:, 0::2] + 1j * y[:, 1::2]).astype(C64))
    k = np.arange(NZ + 1)
    zk, zm = zf[:, k % NZ], np.conj(zf[:, (NZ - k) % NZ])
    e = ((zk + zm) * F32(0.5)).astype(C64)
    o = ((zk - zm) * C64(-0.5j)).astype(C64)
    xk = (e + np.exp(-2j * np.pi * k / R.N_FFT).astype(C64)[None, :] * o).astype(C64)
    power = (xk.real.astype(F32) ** 2 + xk.imag.astype(F32) ** 2).astype(F32)[:-1]
    mel = np.maximum((power @ R.mel_filters().astype(F32)).astype(F32), F32(1e-10))
    lg = np.log10(mel).T
    lg = np.maximum(lg, lg.max() - 8.0)
    return ((lg + 4.0) / 4.0).astype(F32)
