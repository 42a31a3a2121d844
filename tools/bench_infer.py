"""Inference throughput (the ssak/infer path, BASELINE configs[0] on the GPU): waveform normalise -> Wav2Vec2-base forward ->
greedy CTC decode for batches of 10 s utterances.  usage: PYTHONPATH=. python tools/bench_infer.py [B=32] [steps=20]"""
import json
import sys
import time

import torch

from ssak_amd import hip
from ssak_amd.config import Wav2Vec2Config
from ssak_amd.model import Wav2Vec2ForCTC
from ssak_amd.synth import synth_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = Wav2Vec2Config()
model = Wav2Vec2ForCTC(cfg, seed=69).eval()
g = torch.Generator().manual_seed(69)
sd = {}
for name, (off, n, shape) in model.layout.items():
    if name.endswith("layer_norm.weight"):
        sd[name] = torch.ones(shape)
    elif name.endswith(".bias"):
        sd[name] = torch.zeros(shape)
    elif ".conv.weight" in name or name.endswith("original1"):
        sd[name] = torch.randn(shape, generator=g) * (2.0 / (shape[1] * shape[2])) ** 0.5
    else:
        sd[name] = torch.randn(shape, generator=g) * 0.02
v = sd["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
sd["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
model.load_state_dict(sd)
waves = torch.tensor(synth_batch(B, 160000, seed=1)[0]).cuda()


def step():
    x = hip.wave_normalize(waves, None)
    out = model(x)
    return hip.ctc_greedy_decode(out.logits.contiguous(), None, cfg.pad_token_id)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    ids, n = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"workload": "Wav2Vec2-base forward + greedy CTC decode, bf16, 10 s utterances", "batch": B,
                  "utterances_per_sec": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 2),
                  "forward_tflops": round(148.16 * B / dt / 1e3, 1)}))
