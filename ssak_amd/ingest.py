"""Kaldi-folder audio -> device batches: the counterpart of the reference's on-the-fly loading
(``ssak/utils/dataset.py:630-645`` -> ``ssak/utils/audio.py:24-154``) with everything after the file read on the GPU.

The host only finds the PCM byte range of each segment (``offset = int(start * sr)``, audio.py:85-92; the WAV header is parsed
once per file) and native reader threads (``ssak_read_ranges``) ``pread`` it straight into one pinned staging buffer; one asynchronous H2D copy later the device converts to mono fp32
(``ssak_pcm_to_mono_f32``), converts the sample rate (``ssak_resample_sinc`` = torchaudio's windowed-sinc resampler) and
normalises (``ssak_wave_normalize``, a1).  ``BatchPrefetcher`` reads the next batches on a background thread while the
current step runs, which is what the reference's 6 dataloader workers are for (wav2vec_train.py:360).
PCM WAV only: sox / ffmpeg decoding is outside this path (``RuntimeError``, as the reference's loader raises).
"""
from __future__ import annotations

import ctypes as C
import os
import queue
import struct
import threading
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip


class WavInfo:
    """Where the PCM samples of a RIFF / WAVE file sit: parsed once per path (RIFF chunk walk), cached."""
    __slots__ = ("sample_rate", "channels", "sample_width", "data_offset", "frames", "sig")

    def __init__(self, sample_rate, channels, sample_width, data_offset, frames):
        self.sample_rate, self.channels, self.sample_width, self.data_offset, self.frames = sample_rate, channels, sample_width, data_offset, frames
        self.sig = None  # (st_mtime_ns, st_size) of the file the header was read from: a rewritten file is parsed again


_WAV_CACHE: dict = {}
_WAV_LOCK = threading.Lock()


def wav_info(path: str) -> WavInfo:
    """Header of a PCM WAV file (format tag 1, or WAVE_FORMAT_EXTENSIBLE with the PCM sub-format), as the reference's loader needs
    it (sample rate, channels, frames: ssak/utils/audio.py:64-92).  RuntimeError for anything else, as audio.py:49-55 raises."""
    info = _WAV_CACHE.get(path)
    if info is not None:
        try:  # one stat (~1 us) per visit: a file rewritten between epochs must not keep its old frame count / data offset
            st0 = os.stat(path)
            if info.sig == (st0.st_mtime_ns, st0.st_size):
                return info
        except OSError:
            pass  # gone or unreadable: the open below raises the reference's error
    # one open, one 4 KiB read (the fmt and data chunk headers of nearly every file), one fstat: the header walk costs as much as
    # reading a cached 320 KB file when it takes a buffered open, three seeks and four reads
    try:
        fd = os.open(path, os.O_RDONLY)
    except (FileNotFoundError, IsADirectoryError, NotADirectoryError):
        raise RuntimeError(f"File not found: {path}") from None  # audio.py:49-51
    except OSError as err:
        raise RuntimeError(f"Could not read {path} as PCM WAV (sox/ffmpeg decoding is not built): {err}") from err
    try:
        st = os.fstat(fd)
        import stat as _stat
        if not _stat.S_ISREG(st.st_mode):
            raise RuntimeError(f"File not found: {path}")
        buf = os.pread(fd, 4096, 0)
        base = 0  # file offset of buf[0]

        def at(pos, n):  # n bytes at file offset pos (from the first block, else a read of its own)
            nonlocal buf, base
            if pos < base or pos + n > base + len(buf):
                buf, base = os.pread(fd, max(n, 4096), pos), pos
            return buf[pos - base:pos - base + n]

        head = at(0, 12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise RuntimeError(f"Could not read {path} as PCM WAV (sox/ffmpeg decoding is not built): not a RIFF / WAVE file")
        fmt = None
        pos = 12
        while True:
            ck = at(pos, 8)
            if len(ck) < 8:
                raise RuntimeError(f"Could not read {path} as PCM WAV (sox/ffmpeg decoding is not built): no data chunk")
            cid, size = ck[:4], struct.unpack("<I", ck[4:])[0]
            if cid == b"fmt ":
                body = at(pos + 8, min(size, 40))
                tag, nch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
                if tag == 0xFFFE and len(body) >= 26:
                    tag = struct.unpack("<H", body[24:26])[0]
                fmt = (tag, nch, sr, bits)
            elif cid == b"data":
                if fmt is None:
                    raise RuntimeError(f"Could not read {path} as PCM WAV (sox/ffmpeg decoding is not built): data before fmt")
                tag, nch, sr, bits = fmt
                if tag != 1:
                    raise RuntimeError(f"{path}: compressed WAV is not supported (PCM only)")
                sw = (bits + 7) // 8
                if sw not in (1, 2, 4) or nch < 1:
                    raise RuntimeError(f"{path}: unsupported sample width {sw}")
                avail = max(0, st.st_size - (pos + 8))  # (a truncated file: what is really there)
                info = WavInfo(sr, nch, sw, pos + 8, min(size, avail) // (nch * sw))
                break
            pos += 8 + size + (size & 1)
    except (OSError, struct.error) as err:
        raise RuntimeError(f"Could not read {path} as PCM WAV (sox/ffmpeg decoding is not built): {err}") from err
    finally:
        os.close(fd)
    info.sig = (st.st_mtime_ns, st.st_size)
    with _WAV_LOCK:
        _WAV_CACHE[path] = info
    return info


def clear_wav_cache() -> None:
    """Forget every parsed header (tests; a caller that knows its folder changed)."""
    with _WAV_LOCK:
        _WAV_CACHE.clear()


def segment_range(info: WavInfo, start: Optional[float], end: Optional[float]) -> Tuple[int, int]:
    """(file offset, frames) of the segment: offset = int(start * sr) frames in, int((end - start) * sr) frames long, clipped to the
    file (audio.py:85-92)."""
    n, sr = info.frames, info.sample_rate
    s0 = min(n, int(float(start) * sr)) if start else 0
    cnt = min(n - s0, int((float(end) - float(start or 0)) * sr)) if end else n - s0
    return info.data_offset + s0 * info.channels * info.sample_width, max(0, cnt)


def read_pcm_segment(path: str, start: Optional[float] = None, end: Optional[float] = None) -> Tuple[bytes, int, int, int, int]:
    """(raw interleaved PCM bytes of the segment, sample_rate, channels, bytes per sample, frames)."""
    info = wav_info(path)
    off, cnt = segment_range(info, start, end)
    fd = os.open(path, os.O_RDONLY)
    try:
        raw = os.pread(fd, cnt * info.channels * info.sample_width, off)
    finally:
        os.close(fd)
    return raw, info.sample_rate, info.channels, info.sample_width, len(raw) // (info.channels * info.sample_width)


class DeviceIngest:
    """Turns lists of (path, start, end) into normalised fp32 batches on the device."""

    def __init__(self, sample_rate: int = 16000, device="cuda:0", normalize: bool = True, readers: Optional[int] = None):
        """``readers``: file-reader threads (default: the CPUs this process may use, at most 8; the reference runs 6 dataloader
        worker processes, wav2vec_train.py:360)."""
        self.sample_rate, self.device, self.normalize = sample_rate, torch.device(device), normalize
        n_cpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        self.readers = max(1, min(8, n_cpu) if readers is None else int(readers))
        self._tables = {}
        self._stream = None  # the device part's own stream (to_device_async)
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

    def stage(self, items: Sequence[Tuple[str, Optional[float], Optional[float]]], labels: Optional[np.ndarray] = None):
        """Host part (no GPU work): the byte ranges of the segments (headers parsed once per file, in parallel) into ONE pinned
        buffer, read by the reader threads straight into their slices of it.  ``labels`` (int64 [B, L], -100 padding): appended
        to the same buffer, so that the batch's ONE H2D copy carries them too (a ``labels.to(device)`` from pageable memory on
        the compute stream blocks the host until the previous step has drained)."""
        paths = [p for p, _, _ in items]
        infos = [wav_info(p) for p in paths]  # (first visit: ~10 us per header, later ones one stat: cheaper here than a hand-off)
        ranges = [segment_range(i, s, e) for i, (_, s, e) in zip(infos, items)]
        sizes = [cnt * i.channels * i.sample_width for i, (_, cnt) in zip(infos, ranges)]
        audio = sum(sizes)
        lab_off = (audio + 7) // 8 * 8
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int64)
        total = audio if lab is None else lab_off + lab.nbytes
        slot = self._staging(total)
        pinned = slot[0]
        if lab is not None:
            pinned.numpy()[lab_off:lab_off + lab.nbytes] = lab.reshape(-1).view(np.uint8)
        offs, pos = [], 0
        for nbytes in sizes:
            offs.append(pos)
            pos += nbytes
        # the reads: native threads pread the ranges straight into their slices of the pinned buffer (ssak_read_ranges; the
        # interpreter lock is released for the call)
        n, base = len(items), pinned.data_ptr()
        c_paths = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
        c_foff = (C.c_int64 * n)(*[foff for foff, _ in ranges])
        c_size = (C.c_int64 * n)(*sizes)
        c_dst = (C.c_void_p * n)(*[base + o for o in offs])
        try:
            hip.check(hip.lib.ssak_read_ranges(c_paths, c_foff, c_size, c_dst, n, self.readers))
        except ValueError as err:  # an unreadable or truncated file: what the reference's loader raises (audio.py:49-55)
            raise RuntimeError(str(err)) from None
        meta = [(i.sample_rate, i.channels, i.sample_width, cnt) for i, (_, cnt) in zip(infos, ranges)]
        return (pinned[:max(total, 1)], slot), offs, meta, (None if lab is None else (lab_off, lab.shape))

    def to_device_async(self, staged):
        """``to_device`` on the ingest's OWN stream: the H2D copy runs on a copy engine and the small decode / normalise kernels beside
        the caller's work instead of in front of it.  Returns ((waves, lens[, labels]), event); the consumer's stream waits for the event
        (``BatchPrefetcher`` does, one batch ahead)."""
        with torch.cuda.device(self.device):
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            with torch.cuda.stream(self._stream):
                out = self.to_device(staged)
                ev = torch.cuda.Event()
                ev.record(self._stream)
        return out, ev

    def to_device(self, staged):
        """Device part: H2D copy, PCM -> mono fp32 -> target rate -> zero-mean / unit-variance.  Returns (waves [B, T] fp32,
        lens [B] int32), both on the device; no host synchronisation."""
        (pinned, slot), offs, meta, lab_info = staged
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
            if lab_info is not None:
                lo, shape = lab_info
                return waves, lens, raw[lo:lo + 8 * int(np.prod(shape))].view(torch.int64).view(*shape)
        return waves, lens

    def load_batch(self, items):
        return self.to_device(self.stage(items))


class BatchPrefetcher:
    """Iterates over batches of items; file reads + pinned staging of the next ``depth`` batches run on a background thread."""

    def __init__(self, ingest: DeviceIngest, batches: Sequence[Sequence[Tuple[str, Optional[float], Optional[float]]]], depth: int = 2,
                 labels: Optional[Sequence[np.ndarray]] = None):
        """``labels``: one int64 [B, L] array per batch (-100 padding); the iterator then yields (waves, lens, labels) with the labels
        on the device too, carried by the batch's one H2D copy."""
        self.ingest, self.batches = ingest, list(batches)
        self.labels = None if labels is None else list(labels)
        self.q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def _work(self):
        try:
            for k, b in enumerate(self.batches):
                self.q.put(self.ingest.stage(b, None if self.labels is None else self.labels[k]))
        except BaseException as err:  # surfaced on the consumer side
            self.q.put(err)
        self.q.put(None)

    def __iter__(self):
        # the device part of batch k + 1 (H2D copy + decode + normalise on the ingest's stream) is issued before batch k is handed
        # out: it overlaps the consumer's step instead of sitting in front of it on the consumer's stream
        ahead = None
        while True:
            item = self.q.get()
            err = item if isinstance(item, BaseException) else None
            nxt = None
            if err is None and item is not None:
                try:
                    nxt = self.ingest.to_device_async(item)
                except BaseException as e:  # noqa: BLE001 -- re-raised below, after the batch already in flight was handed out
                    err = e
            if ahead is not None:
                out, ev = ahead
                cur = torch.cuda.current_stream(self.ingest.device)
                cur.wait_event(ev)
                for t in out:
                    t.record_stream(cur)  # (allocated on the ingest's stream, consumed on the caller's)
                yield out  # (batch k is complete whatever happened to batch k + 1: the consumer runs its step before the error)
            if err is not None:
                raise err
            if item is None:
                return
            ahead = nxt
