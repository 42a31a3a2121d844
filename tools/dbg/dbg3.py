import os, sys
sys.path.insert(0, os.getcwd())
import dataclasses, numpy as np, torch
from oracle import w2v2_ref as R
from oracle.gen_golden_full import CURVE, curve_inputs
from ssak_amd.config import Wav2Vec2Config
from ssak_amd.model import Wav2Vec2ForCTC
from ssak_amd.trainer import AdamW, Trainer
z = np.load("tests/golden/w2v2_base_curve.npz")
oc = R.W2V2Config.base().deterministic()
d = dataclasses.asdict(oc); d.pop("initializer_range")
model = Wav2Vec2ForCTC(Wav2Vec2Config(**d)).train()
model.load_state_dict(R.init_params(oc, 69))
steps = int(z["steps"])
opt = AdamW(model, lr=float(z["base_lr"]), warmup_steps=int(z["warmup"]), total_steps=steps, max_grad_norm=1.0)
tr = Trainer(model, opt)
batches = [(torch.tensor(x).cuda(), torch.tensor(l).cuda()) for x, l in curve_inputs()]
for s in range(steps):
    x, l = batches[s % 4]
    loss = float(tr.train_step(x, None, l, raw=False).item())
    gn = opt.grad_norm()
    print(s, f"{loss:.5f} {z['loss'][s]:.5f} rel {abs(loss - z['loss'][s]) / z['loss'][s]:.2e}  gnorm {gn:.3f} {z['grad_norm'][s]:.3f}")
