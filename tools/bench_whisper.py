"""BASELINE config 4: Whisper-small encoder + CTC head train step on one MI355X (secondary measurement, not the
headline bench line).  Step = log-mel (a13) -> encoder forward -> CTC -> backward -> clip + AdamW on synthetic 30 s windows."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from ssak_amd import hip
from ssak_amd.synth import synth_wave
from ssak_amd.trainer import AdamW
from ssak_amd.whisper import WhisperCTCConfig, WhisperEncoderForCTC

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = WhisperCTCConfig()
model = WhisperEncoderForCTC(cfg).train()
g = torch.Generator().manual_seed(0)
sd = {}
for n, (off, cnt, shape) in model.layout.items():
    if n.endswith("embed_positions.weight"):
        length, channels = shape  # Whisper's fixed sinusoidal table
        inv = torch.exp(-(np.log(10000.0) / (channels // 2 - 1)) * torch.arange(channels // 2))
        t = torch.arange(length).view(-1, 1) * inv.view(1, -1)
        sd[n] = torch.cat([t.sin(), t.cos()], dim=1)
    elif "layer_norm.weight" in n:
        sd[n] = torch.ones(shape)
    elif n.endswith(".bias"):
        sd[n] = torch.zeros(shape)
    else:
        sd[n] = torch.randn(shape, generator=g) * 0.02
model.load_state_dict(sd)
opt = AdamW(model, lr=1e-4, warmup_steps=500)
rng = np.random.default_rng(0)
wav = torch.tensor(np.stack([synth_wave(rng, 480000) for _ in range(B)])).cuda()
labels = torch.randint(1, cfg.vocab_size, (B, 200)).cuda()


def step():
    mel = hip.logmel_whisper(wav)
    out = model(mel, labels=labels)
    model.backward()
    opt.step()
    return out.loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
gf = 1032.0  # ~3 x 344.16 GF per 30 s window, nothing frozen
print(json.dumps({"workload": "Whisper-small encoder + CTC head train step, bf16, 30 s windows (BASELINE configs[3])",
                  "windows_per_sec": round(B / dt, 2), "audio_sec_per_sec": round(30 * B / dt, 1), "ms_per_step": round(dt * 1e3, 2),
                  "batch": B, "whole_step_tflops": round(gf * B / dt / 1e3, 1), "loss": round(float(loss.item()), 4)}))
