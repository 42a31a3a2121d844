import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from ssak_amd.config import Wav2Vec2Config
from ssak_amd.model import Wav2Vec2ForCTC
from ssak_amd.trainer import AdamW, Trainer
from ssak_amd.synth import synth_batch
cfg = Wav2Vec2Config(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                     feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)
model = Wav2Vec2ForCTC(cfg, freeze_feature_encoder=False, seed=1).train()
g = torch.Generator().manual_seed(1)
sd = {}
for name, (off, n, shape) in model.layout.items():
    if name.endswith("layer_norm.weight"): t = torch.ones(shape)
    elif name.endswith(".bias"): t = torch.zeros(shape)
    elif name.endswith("masked_spec_embed"): t = torch.rand(shape, generator=g)
    elif ".conv.weight" in name or name.endswith("original1"): t = torch.randn(shape, generator=g) * (2.0 / (shape[1] * shape[2])) ** 0.5
    else: t = torch.randn(shape, generator=g) * 0.02
    sd[name] = t
v = sd["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
sd["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
model.load_state_dict(sd)
tr = Trainer(model, AdamW(model, lr=1e-4, warmup_steps=10))
B = 8
w, lab = synth_batch(B, 160000, seed=3)
w, lab = torch.tensor(w).cuda(), torch.tensor(lab).cuda()
lens = torch.full((B,), 160000, dtype=torch.int32).cuda()
for _ in range(2): l = tr.train_step(w, lens, lab)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): l = tr.train_step(w, lens, lab)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("XLSR-large --no_freeze B=8 x 10 s:", round(B / dt, 1), "utt/s loss", float(l))
