"""GPU: audio ingest -- PCM decode / mono mix / sinc resampling / normalisation on the device against the
CPU restatement (oracle/resample_ref.py), through real WAV files."""
import os
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write(path, x, sr, nch=1):
    pcm = np.round(np.clip(x, -1, 1) * 32767.0).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(nch)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes(pcm.tobytes())
    return pcm


def _signal(rng, n, sr):
    t = np.arange(n) / sr
    return (0.3 * np.sin(2 * np.pi * 313.0 * t) + 0.2 * np.sin(2 * np.pi * 1900.0 * t + 1.0) + 0.05 * rng.standard_normal(n)).astype(np.float32)


def test_ingest_matches_oracle(tmp_path):
    """Mixed batch: 16 kHz mono (no rate change), 44.1 kHz stereo with a segment cut, 8 kHz mono (upsampled), 48 kHz mono."""
    from oracle import resample_ref as R
    from oracle.w2v2_ref import zero_mean_unit_var_norm
    from ssak_amd.ingest import DeviceIngest
    rng = np.random.default_rng(0)
    specs = [(16000, 1, 23456, None, None), (44100, 2, 61234, 0.25, 1.1), (8000, 1, 9001, None, 0.9), (48000, 1, 30011, 0.1, None),
             (44100, 2, 5000, None, None)]
    items, refs = [], []
    for i, (sr, nch, n, s, e) in enumerate(specs):
        x = _signal(rng, n * nch, sr)
        p = str(tmp_path / f"u{i}.wav")
        pcm = _write(p, x, sr, nch)
        frames = pcm.astype(np.float32) / 32768.0
        frames = frames.reshape(-1, nch) if nch > 1 else frames
        s0 = int(s * sr) if s else 0                                   # audio.py:85-91
        cnt = int((e - (s or 0)) * sr) if e else len(frames) - s0
        mono = R.to_mono(frames[s0:s0 + cnt])
        refs.append(R.resample(mono, sr, 16000))
        items.append((p, s, e))
    ing = DeviceIngest(16000, normalize=False)
    waves, lens = ing.load_batch(items)
    assert lens.cpu().tolist() == [len(r) for r in refs]
    w = waves.cpu().numpy()
    for b, r in enumerate(refs):
        assert np.abs(w[b, :len(r)] - r).max() < 2e-6 * max(1.0, np.abs(r).max()) + 2e-6, b
        assert (w[b, len(r):] == 0).all()
    # with normalisation: the same as the a1 oracle applied to the oracle's waveforms
    wn, ln = DeviceIngest(16000).load_batch(items)
    want = zero_mean_unit_var_norm(refs)
    assert np.abs(wn.cpu().numpy()[:, :want.shape[1]] - want).max() < 5e-5


def test_ingest_prefetcher_and_errors(tmp_path):
    from ssak_amd.ingest import BatchPrefetcher, DeviceIngest
    rng = np.random.default_rng(1)
    paths = []
    for i in range(6):
        p = str(tmp_path / f"v{i}.wav")
        _write(p, _signal(rng, 16000 + 500 * i, 16000), 16000)
        paths.append((p, None, None))
    ing = DeviceIngest(16000)
    got = [(w.shape, l.cpu().tolist()) for w, l in BatchPrefetcher(ing, [paths[:3], paths[3:]], depth=2)]
    assert got[0][1] == [16000, 16500, 17000] and got[1][1] == [17500, 18000, 18500]
    with pytest.raises(RuntimeError):
        list(BatchPrefetcher(ing, [[(str(tmp_path / "missing.wav"), None, None)]]))


def test_ingest_prefetcher_carries_labels_and_runs_one_batch_ahead(tmp_path):
    """The prefetcher's device part runs on the ingest's own stream one batch ahead of the consumer (H2D copy + decode + normalise
    beside the consumer's step, an event hand-off) and the label matrix of a batch rides in the same pinned buffer and H2D copy:
    every batch's waves equal `load_batch` on the same items bit for bit, the labels come back unchanged, segments (start / end)
    and files of different lengths included, while the consumer stream is kept busy."""
    from ssak_amd.ingest import BatchPrefetcher, DeviceIngest
    rng = np.random.default_rng(2)
    items = []
    for i in range(10):
        p = str(tmp_path / f"w{i}.wav")
        _write(p, _signal(rng, 20000 + 777 * i, 16000), 16000)
        items.append((p, None, None) if i % 3 else (p, 0.25, 1.0))
    batches = [items[:4], items[4:7], items[7:]]
    labels = [rng.integers(-100, 31, (len(b), 5 + k)).astype(np.int64) for k, b in enumerate(batches)]
    ing = DeviceIngest(16000)
    want = [DeviceIngest(16000).load_batch(b) for b in batches]
    busy = torch.randn(2048, 2048, device="cuda")
    got = []
    for w, l, y in BatchPrefetcher(ing, batches, depth=2, labels=labels):
        for _ in range(20):
            busy = busy @ busy * 1e-3  # (the consumer's stream has work queued while the next batch's device part is issued)
        got.append((w.clone(), l.clone(), y.clone()))
    torch.cuda.synchronize()
    assert len(got) == 3
    for (w, l, y), (ww, wl), lab in zip(got, want, labels):
        assert torch.equal(w, ww) and torch.equal(l, wl)
        assert y.dtype == torch.int64 and np.array_equal(y.cpu().numpy(), lab)


def test_ingest_prefetcher_hands_out_the_batch_in_flight_before_a_later_error(tmp_path):
    """A bad file in batch k + 1 must not swallow batch k (whose H2D copy is already issued): the consumer gets batch k, runs its
    step, and sees the error at the next fetch -- under data parallelism every rank then stops at the same collective."""
    from ssak_amd.ingest import BatchPrefetcher, DeviceIngest
    rng = np.random.default_rng(3)
    paths = []
    for i in range(4):
        p = str(tmp_path / f"x{i}.wav")
        _write(p, _signal(rng, 16000 + 100 * i, 16000), 16000)
        paths.append((p, None, None))
    batches = [paths[:2], paths[2:], [(str(tmp_path / "missing.wav"), None, None)], paths[:2]]
    seen = []
    with pytest.raises(RuntimeError):
        for w, l in BatchPrefetcher(DeviceIngest(16000), batches, depth=2):
            seen.append(l.cpu().tolist())
    assert seen == [[16000, 16100], [16200, 16300]]


def test_wav_header_cache_follows_a_rewritten_file(tmp_path):
    """The header cache is keyed on (mtime, size): a file rewritten between epochs is parsed again."""
    from ssak_amd.ingest import DeviceIngest, clear_wav_cache, wav_info
    rng = np.random.default_rng(4)
    p = str(tmp_path / "r.wav")
    _write(p, _signal(rng, 16000, 16000), 16000)
    assert wav_info(p).frames == 16000
    _write(p, _signal(rng, 12345, 16000), 16000)
    os.utime(p, ns=(1, 1))  # (even with a coarse clock the size differs; this also covers an mtime going backwards)
    assert wav_info(p).frames == 12345
    w, l = DeviceIngest(16000, normalize=False).load_batch([(p, None, None)])
    assert l.cpu().tolist() == [12345]
    clear_wav_cache()
    assert wav_info(p).frames == 12345
