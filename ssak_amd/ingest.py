"""Kaldi-folder audio -> device batches: the counterpart of the reference's on-the-fly loading
(``ssak/utils/dataset.py:630-645`` -> ``ssak/utils/audio.py:24-154``) with everything after the file read on the GPU.

The host only finds the PCM byte range of each segment (``offset = int(start * sr)``, audio.py:85-92) and copies it into
one pinned staging buffer; one asynchronous H2D copy later the device converts to mono fp32
(``ssak_pcm_to_mono_f32``), converts the sample rate (``ssak_resample_sinc`` = torchaudio's windowed-sinc resampler) and
normalises (``ssak_wave_normalize``, a1).  ``BatchPrefetcher`` reads the next batches on a background thread while the
current step runs, which is what the reference's 6 dataloader workers are for (wav2vec_train.py:360).
PCM WAV only: sox / ffmpeg decoding is outside this path (``RuntimeError``, as the reference's loader raises).
"""
from __future__ import annotations

import ctypes as C
import os
import queue
import threading
import wave
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip


def read_pcm_segment(path: str, start: Optional[float] = None, end: Optional[float] = None) -> Tuple[bytes, int, int, int, int]:
    """(raw interleaved PCM bytes of the segment, sample_rate, channels, bytes per sample, frames)."""
    if not os.path.isfile(path):
        raise RuntimeError(f"File not found: {path}")  # audio.py:49-51
    try:
        with wave.open(path, "rb") as f:
            sr, nch, sw, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
            if f.getcomptype() != "NONE":
                raise RuntimeError(f"{path}: compressed WAV is not supported (PCM only)")
            s0 = min(n, int(float(start) * sr)) if start else 0                      # audio.py:85-87
            cnt = min(n - s0, int((float(end) - float(start or 0)) * sr)) if end else n - s0  # audio.py:89-91
            f.setpos(s0)
            raw = f.readframes(max(0, cnt))
    except (wave.Error, EOFError) as err:
        raise RuntimeError(f"Could not read {path} as PCM WAV (sox/ffmpeg decoding is not built): {err}") from err
    if sw not in (1, 2, 4):
        raise RuntimeError(f"{path}: unsupported sample width {sw}")
    return raw, sr, nch, sw, len(raw) // (nch * sw)


class DeviceIngest:
    """Turns lists of (path, start, end) into normalised fp32 batches on the device."""

    def __init__(self, sample_rate: int = 16000, device="cuda:0", normalize: bool = True):
        self.sample_rate, self.device, self.normalize = sample_rate, torch.device(device), normalize
        self._tables = {}
        # ring of reusable pinned staging buffers: pinning a fresh buffer per batch costs milliseconds (and the caching host
        # allocator cannot recycle one whose copy is still pending); [tensor, event of the last H2D copy out of it]
        self._ring = [[None, None] for _ in range(6)]
        self._next = 0

    def _staging(self, nbytes: int):
        slot = self._ring[self._next % len(self._ring)]
        self._next += 1
        if slot[1] is not None:
            slot[1].synchronize()  # the copy that last read this buffer has completed (normally long ago)
        if slot[0] is None or slot[0].numel() < nbytes:
            slot[0] = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8).pin_memory()
        return slot

    def _table(self, sr: int):
        if sr not in self._tables:
            o, n, w, taps = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            hip.check(hip.lib.ssak_resample_plan(sr, self.sample_rate, C.byref(o), C.byref(n), C.byref(w), C.byref(taps)))
            host = torch.empty(n.value * taps.value, dtype=torch.float32)
            hip.check(hip.lib.ssak_resample_table(sr, self.sample_rate, C.c_void_p(host.data_ptr())))
            self._tables[sr] = (host.to(self.device), o.value, n.value)
        return self._tables[sr]

    def stage(self, items: Sequence[Tuple[str, Optional[float], Optional[float]]]):
        """Host part (thread-safe, no GPU work): read the byte ranges into ONE pinned buffer."""
        segs = [read_pcm_segment(p, s, e) for p, s, e in items]
        total = sum(len(r) for r, *_ in segs)
        slot = self._staging(total)
        pinned = slot[0]
        view = pinned.numpy()
        offs, pos = [], 0
        for raw, *_ in segs:
            view[pos:pos + len(raw)] = np.frombuffer(raw, dtype=np.uint8)
            offs.append(pos)
            pos += len(raw)
        meta = [(sr, nch, sw, n) for _, sr, nch, sw, n in segs]
        return (pinned[:max(total, 1)], slot), offs, meta

    def to_device(self, staged):
        """Device part: H2D copy, PCM -> mono fp32 -> target rate -> zero-mean / unit-variance.  Returns (waves [B, T] fp32,
        lens [B] int32), both on the device; no host synchronisation."""
        (pinned, slot), offs, meta = staged
        B = len(meta)
        dev = self.device
        with torch.cuda.device(dev):
            raw = pinned.to(dev, non_blocking=True)
            if slot[1] is None:
                slot[1] = torch.cuda.Event()
            slot[1].record()
            out_len = [n if sr == self.sample_rate else -(-(self._table(sr)[2] * n) // self._table(sr)[1]) for sr, _, _, n in meta]
            T = max(max(out_len), 1)
            T = (T + 7) // 8 * 8
            waves = torch.zeros((B, T), dtype=torch.float32, device=dev)
            lens = torch.tensor(out_len, dtype=torch.int32).to(dev, non_blocking=True)
            groups = {}
            for i, (sr, nch, sw, n) in enumerate(meta):
                groups.setdefault((sr, nch, sw), []).append(i)
            for (sr, nch, sw), idx in groups.items():
                nb = len(idx)
                tin = max(max(meta[i][3] for i in idx), 1)
                off_d = torch.tensor([offs[i] for i in idx], dtype=torch.int64).to(dev, non_blocking=True)
                nfr_d = torch.tensor([meta[i][3] for i in idx], dtype=torch.int32).to(dev, non_blocking=True)
                mono = waves if (sr == self.sample_rate and nb == B) else torch.empty((nb, tin), dtype=torch.float32, device=dev)
                tmax = mono.shape[1]
                hip.check(hip.lib.ssak_pcm_to_mono_f32(hip.ptr(raw), hip.ptr(off_d), hip.ptr(nfr_d), nb, nch, sw, tmax, hip.ptr(mono),
                                                       hip.stream()))
                if sr != self.sample_rate:
                    table, _, _ = self._table(sr)
                    res = waves if nb == B else torch.empty((nb, T), dtype=torch.float32, device=dev)
                    hip.check(hip.lib.ssak_resample_sinc(hip.ptr(mono), hip.ptr(nfr_d), nb, tmax, sr, self.sample_rate, hip.ptr(table),
                                                         hip.ptr(res), T, None, hip.stream()))
                    mono = res
                if mono is not waves:
                    waves[torch.tensor(idx, device=dev), :mono.shape[1]] = mono
            if self.normalize:
                waves = hip.wave_normalize(waves, lens)
        return waves, lens

    def load_batch(self, items):
        return self.to_device(self.stage(items))


class BatchPrefetcher:
    """Iterates over batches of items; file reads + pinned staging of the next ``depth`` batches run on a background thread."""

    def __init__(self, ingest: DeviceIngest, batches: Sequence[Sequence[Tuple[str, Optional[float], Optional[float]]]], depth: int = 2):
        self.ingest, self.batches = ingest, list(batches)
        self.q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def _work(self):
        try:
            for b in self.batches:
                self.q.put(self.ingest.stage(b))
        except BaseException as err:  # surfaced on the consumer side
            self.q.put(err)
        self.q.put(None)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield self.ingest.to_device(item)
