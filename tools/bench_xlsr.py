"""BASELINE config 5: Wav2Vec2-large-XLSR CTC train step on mixed-length, length-grouped batches, one MI355X
(secondary measurement, not the headline bench line; SURVEY.md section 8d "Config 5").

Model: Wav2Vec2Config(hidden 1024, 24 layers, 16 heads, FFN 4096, feat_extract_norm="layer", conv_bias, stable layer norm),
feature encoder frozen, regularisers at the train script's defaults.  Data: N synthetic utterances with durations drawn
log-uniformly in [1 s, 15 s] (the script's --min/--max_duration, wav2vec_train.py:149-150), batched as HF's
LengthGroupedSampler does (mega-batches of 50 x B sorted by length, docker/transformers_modified/trainer.py:758-775),
right-padded to the longest of the batch, with lengths (the attention mask).  Reports utterances/s and audio-seconds/s
over one pass of all batches, and algorithmic TFLOP/s counted on the real (unpadded) lengths.

usage: python tools/bench_xlsr.py [B=16] [N=320] [--dp-rank R --dp-world W]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from ssak_amd.config import Wav2Vec2Config
from ssak_amd.data import length_grouped_batches
from ssak_amd.model import Wav2Vec2ForCTC, conv_out_lengths
from ssak_amd.synth import synth_text, synth_wave, text_to_ids
from ssak_amd.trainer import AdamW, Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 320

cfg = Wav2Vec2Config(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                     feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)


def train_gflop(T: int) -> float:
    """Algorithmic GFLOP of one train step on one utterance of T samples, frozen feature encoder."""
    L, cin, fe = T, 1, 0.0
    for c, k, s in zip(cfg.conv_dim, cfg.conv_kernel, cfg.conv_stride):
        L = (L - k) // s + 1
        fe += 2.0 * L * c * cin * k
        cin = c
    H, I, F = cfg.hidden_size, cfg.intermediate_size, L
    rest = 2.0 * F * cin * H + 2.0 * F * H * (H // cfg.num_conv_pos_embedding_groups) * cfg.num_conv_pos_embeddings
    rest += cfg.num_hidden_layers * (2.0 * F * (4 * H * H + 2 * H * I) + 4.0 * F * F * H) + 2.0 * F * H * cfg.vocab_size
    return (fe + 3.0 * rest) / 1e9


assert abs(train_gflop(160000) - 1053.50) < 1.0, train_gflop(160000)  # XLSR-large @10 s

model = Wav2Vec2ForCTC(cfg, freeze_feature_encoder=True, seed=69).train()
g = torch.Generator().manual_seed(69)
sd = {}
for name, (off, n, shape) in model.layout.items():
    if name.endswith("layer_norm.weight"):
        t = torch.ones(shape)
    elif name.endswith(".bias"):
        t = torch.zeros(shape)
    elif name.endswith("masked_spec_embed"):
        t = torch.rand(shape, generator=g)
    elif ".conv.weight" in name or name.endswith("original1"):
        t = torch.randn(shape, generator=g) * (2.0 / (shape[1] * shape[2])) ** 0.5
    else:
        t = torch.randn(shape, generator=g) * 0.02
    sd[name] = t
v = sd["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
sd["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
model.load_state_dict(sd)
opt = AdamW(model, lr=1e-4, warmup_steps=500)
trainer = Trainer(model, opt)

rng = np.random.default_rng(1234)
durs = np.exp(rng.uniform(np.log(1.0), np.log(15.0), N))
nsamp = (durs * 16000).astype(np.int64)
batches = length_grouped_batches(nsamp.tolist(), B, np.random.RandomState(0))
batches = [b for b in batches if len(b) == B]
dev_batches = []
for idx in batches:
    T = int(max(nsamp[i] for i in idx))
    T = (T + 7) // 8 * 8
    wav = np.zeros((B, T), np.float32)
    ids = []
    for r, i in enumerate(idx):
        wav[r, :nsamp[i]] = synth_wave(rng, int(nsamp[i]))
        # transcripts scaled to the duration (about 8 characters per second), always feasible for CTC
        fl = int(conv_out_lengths(cfg, np.array([nsamp[i]]))[0])
        n = max(1, min(int(durs[i] * 8), (fl - 1) // 2))
        ids.append(text_to_ids(synth_text(rng, n, n)))
    Lm = max(len(x) for x in ids)
    lab = np.full((B, Lm), -100, np.int64)
    for r, x in enumerate(ids):
        lab[r, :len(x)] = x
    dev_batches.append((torch.tensor(wav).cuda(), torch.tensor(nsamp[idx].astype(np.int32)).cuda(), torch.tensor(lab).cuda()))

# warm-up on the longest batch (sizes the workspace once) and one short one
order = sorted(range(len(dev_batches)), key=lambda j: -dev_batches[j][0].shape[1])
for j in (order[0], order[-1]):
    trainer.train_step(*dev_batches[j])
torch.cuda.synchronize()
t0 = time.perf_counter()
for w, l, y in dev_batches:
    loss = trainer.train_step(w, l, y)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
utts = B * len(dev_batches)
used = [i for b in batches for i in b]
audio = float(sum(durs[i] for i in used))
padded = float(sum(dev_batches[j][0].shape[1] * B for j in range(len(dev_batches)))) / 16000.0
gf = sum(train_gflop(int(nsamp[i])) for i in used)
print(json.dumps({"workload": "Wav2Vec2-large-XLSR CTC train step, bf16, durations log-uniform 1-15 s, length-grouped batches "
                              "(BASELINE configs[4], single GPU)",
                  "utterances_per_sec": round(utts / dt, 2), "audio_sec_per_sec": round(audio / dt, 1),
                  "batch": B, "batches": len(dev_batches), "ms_per_step": round(dt / len(dev_batches) * 1e3, 2),
                  "padding_overhead": round(padded / audio - 1.0, 4), "algorithmic_tflops": round(gf / dt / 1e3, 1),
                  "loss": round(float(loss.item()), 4)}))
