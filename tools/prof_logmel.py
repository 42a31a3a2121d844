"""Per-kernel times of the Whisper log-mel front end (a13): run under rocprofv3 --kernel-trace --stats.
usage: PYTHONPATH=. python tools/prof_logmel.py [B=8]"""
import sys
import numpy as np
import torch
import ssak_amd.hip as hip
from ssak_amd.synth import synth_wave

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(0)
wav = torch.tensor(np.stack([synth_wave(rng, 480000) for _ in range(B)])).cuda()
for _ in range(3):
    hip.logmel_whisper(wav)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    hip.logmel_whisper(wav)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"log-mel B={B}: {us:.1f} us per call = {us / B:.2f} us per 30 s window; 2.88 MB/window -> {2.88e6 * B / us / 1e3:.1f} GB/s of 8000")
