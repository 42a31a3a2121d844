"""Development check (library built with `make DEV=1`): the positional convolution weight gradient of the direct kernel against the
Toeplitz-GEMM form on identical inputs (SSAK_PCW_GEMM=1 selects the GEMM for this product alone): run twice, compare /tmp/pcw_*.npy."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import w2v2_ref as R
from ssak_amd.config import Wav2Vec2Config
from ssak_amd.model import Wav2Vec2ForCTC
import dataclasses
oc = R.W2V2Config.base(num_hidden_layers=1).deterministic()
d = dataclasses.asdict(oc); d.pop("initializer_range")
p = R.init_params(oc, 41)
rng = np.random.default_rng(9)
x = R.zero_mean_unit_var_norm([rng.standard_normal(160000).astype(np.float32) for _ in range(2)])
labels = R.pad_labels([list(rng.integers(1, 32, 6)), list(rng.integers(1, 32, 4))])
names = ["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0", "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
model = Wav2Vec2ForCTC(Wav2Vec2Config(**d)).train(); model.load_state_dict(p)
out = model(torch.tensor(x), labels=torch.tensor(labels)); model.backward()
g = {n: model.grad(n).double().cpu().numpy().copy() for n in names}
loss, logits, grads = R.loss_and_grads(p, oc, torch.tensor(x), None, torch.tensor(labels))
for n in names:
    r = grads[n].double().numpy()
    print(os.environ.get("SSAK_PCW_GEMM", "direct"), n[-9:], "rel l2 vs oracle", np.linalg.norm(g[n] - r) / np.linalg.norm(r), "norm", np.linalg.norm(g[n]))
np.save(f"/tmp/pcw_{os.environ.get('SSAK_PCW_GEMM','direct')}.npy", np.concatenate([g[n].ravel() for n in names]))
