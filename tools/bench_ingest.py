"""Audio ingest throughput: 10 s utterances read from WAV files into normalised device batches of 32.
usage: PYTHONPATH=. python tools/bench_ingest.py [sr=16000] [channels=1] [N=256]"""
import os
import sys
import tempfile
import time
import wave

import numpy as np
import torch

from ssak_amd.data import load_audio, pad_waves
from ssak_amd.ingest import BatchPrefetcher, DeviceIngest
from ssak_amd import hip

sr = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
B = 32
d = tempfile.mkdtemp()
rng = np.random.default_rng(0)
items = []
for i in range(N):
    p = os.path.join(d, f"u{i}.wav")
    pcm = (rng.standard_normal(10 * sr * nch) * 3000).astype("<i2")
    with wave.open(p, "wb") as f:
        f.setnchannels(nch)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes(pcm.tobytes())
    items.append((p, None, None))
batches = [items[i:i + B] for i in range(0, N, B)]
ing = DeviceIngest(16000)
for w, l in BatchPrefetcher(ing, batches[:2]):
    pass
torch.cuda.synchronize()
t0 = time.perf_counter()
for w, l in BatchPrefetcher(ing, batches, depth=3):
    pass
torch.cuda.synchronize()
dt = time.perf_counter() - t0
line = f"{sr} Hz x {nch} ch: device ingest {N / dt:8.1f} utt/s ({N * 10 / dt:9.0f} audio-s/s)"
if sr == 16000:  # the host path the trainer uses without --online: numpy decode, pad, H2D, device normalise
    t0 = time.perf_counter()
    for b in batches:
        x, lens = pad_waves([load_audio(p) for p, _, _ in b])
        hip.wave_normalize(torch.from_numpy(x).cuda(), torch.from_numpy(lens).cuda())
    torch.cuda.synchronize()
    dh = time.perf_counter() - t0
    line += f"   host decode + H2D {N / dh:8.1f} utt/s"
print(line)
