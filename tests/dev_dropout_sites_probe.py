"""Which dropout site disagrees with the oracle?  One site at a time (all other probabilities 0), engine (fp32-exact mode and
bf16) against oracle/w2v2_ref.py with the engine's hash masks.  Development probe for tests/test_gpu_dropout.py."""
import dataclasses
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import w2v2_ref as R  # noqa: E402
from oracle.gen_golden_dropout import tiny_case  # noqa: E402
from ssak_amd.config import Wav2Vec2Config  # noqa: E402
from ssak_amd.model import Wav2Vec2ForCTC  # noqa: E402


def main():
    for xlsr in (False, True):
        oc0, params, x, lens, labels, mask, keep, seed = tiny_case(xlsr)
        sites = ["none", "attention_dropout", "hidden_dropout", "activation_dropout", "feat_proj_dropout", "final_dropout", "layerdrop", "specaug", "all"]
        for s in sites:
            kw = dict(attention_dropout=0.0, hidden_dropout=0.0, activation_dropout=0.0, feat_proj_dropout=0.0, final_dropout=0.0)
            if s in kw:
                kw[s] = 0.25
            if s == "all":
                kw = {k: 0.25 for k in kw}
            lk = keep if s in ("layerdrop", "all") else None
            mk = mask if s in ("specaug", "all") else None
            if lk is None:
                kw["layerdrop"] = 0.0  # (the engine draws its own decisions when none are supplied)
            if mk is None:
                kw["mask_time_prob"] = 0.0
            oc = dataclasses.replace(oc0, **kw)
            loss, logits, grads = R.loss_and_grads(params, oc, torch.tensor(x), lens, torch.tensor(labels), train=True,
                                                   mask_time_indices=None if mk is None else torch.tensor(mk), layer_keep=lk,
                                                   drop=R.HashDropout(seed))
            for exact in (True, False):
                d = dataclasses.asdict(oc)
                d.pop("initializer_range")
                model = Wav2Vec2ForCTC(Wav2Vec2Config(**d), exact=exact).train()
                model.load_state_dict(params)
                out = model(torch.tensor(x), labels=torch.tensor(labels), mask_time_indices=mk, layer_keep=lk, dropout_seed=seed,
                            lengths=None if lens is None else torch.tensor(lens))
                model.backward()
                lg = out.logits.cpu()
                if lens is not None:
                    fl = R.conv_out_lengths(oc, lens)
                    e = max(float((lg[b, :f] - logits[b, :f]).abs().max()) for b, f in enumerate(fl))
                else:
                    e = float((lg - logits).abs().max())
                worst = ("", 0.0)
                gm = max(float(g.abs().max()) for g in grads.values())
                for n, g in grads.items():
                    got = model.grad(n).cpu()
                    if n in model._HEAD:
                        got = got[:model.config.vocab_size]
                    ee = float((got - g).abs().max()) / max(float(g.abs().max()), 1e-3 * gm)
                    if ee > worst[1]:
                        worst = (n, ee)
                print(f"xlsr={int(xlsr)} site={s:20s} exact={int(exact)} logits max abs err {e:.3e} loss {out.loss.item():.5f} vs {loss.item():.5f} "
                      f"worst grad {worst[1]:.3e} {worst[0]}", flush=True)


if __name__ == "__main__":
    main()
