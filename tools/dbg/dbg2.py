import os, sys
sys.path.insert(0, os.getcwd())
import dataclasses, numpy as np, torch
import torch.distributed as dist
from oracle import w2v2_ref as R
from ssak_amd.config import Wav2Vec2Config
from ssak_amd.model import Wav2Vec2ForCTC
from ssak_amd.trainer import AdamW, Trainer
oc = R.W2V2Config.tiny().deterministic()
d = dataclasses.asdict(oc); d.pop("initializer_range")
p = R.init_params(oc, 3)
rng = np.random.default_rng(1)
x = torch.tensor(R.zero_mean_unit_var_norm([rng.standard_normal(8000).astype(np.float32) for _ in range(2)])).cuda()
labels = torch.tensor(R.pad_labels([[3, 4, 5], [6, 7]])).cuda()
def run(distributed, side):
    model = Wav2Vec2ForCTC(Wav2Vec2Config(**d), seed=11).train()
    model.load_state_dict(p)
    if distributed:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    tr = Trainer(model, AdamW(model, warmup_steps=0, lr=1e-3), optimizer_stream=side)
    out = []
    for _ in range(2):
        loss = tr.train_step(x, None, labels, raw=False)
        torch.cuda.synchronize()
        out.append((float(loss.item()), tr.opt.grad_norm(), model.params.clone(), model.grads.clone()))
    if distributed:
        dist.destroy_process_group()
    return out
a = run(False, False); b = run(False, True); c = run(True, True); e = run(True, False)
for nm, r in (("plain", a), ("side", b), ("dist+side", c), ("dist", e)):
    print(nm, [(l, n) for l, n, _, _ in r])
for nm, r in (("side", b), ("dist+side", c), ("dist", e)):
    for s in range(2):
        print(nm, s, "dparams", float((r[s][2] - a[s][2]).abs().max()), "dgrads", float((r[s][3] - a[s][3]).abs().max()))
