"""Regularisers-ON golden vectors (round 3).  TEST INFRASTRUCTURE ONLY.

Run in the BUILD container (``python -m oracle.gen_golden_dropout [tiny] [tiny_xlsr] [base]``).  The timed configuration of
the benchmark is the train script's default (ssak/train/transformers/wav2vec_train.py:161-165,313-325: attention dropout
0.1, hidden dropout 0.05, LayerDrop 0.1, SpecAugment 0.05; transformers defaults activation 0.1, final 0.1), i.e. every
stochastic regulariser ON.  torch's dropout bits cannot be reproduced on the device, so the comparison runs the other way
round: ``transformers.Wav2Vec2ForCTC`` in ``train()`` mode is made to consume the ENGINE's counter-hash masks
(``oracle/dropout_hash.py``, pinned bit for bit against the device functions by tests/test_gpu_dropout.py):

* every ``nn.Dropout`` module of the HF model is replaced, BY MODULE NAME, with one that multiplies by the hash mask of the
  engine site it corresponds to and by torch's own 1 / (1 - p) (modeling_wav2vec2.py:427-433 feature projection, :554-571
  feed-forward, :586-596 / :621-642 layer residual, :663-692 / :735-765 encoder input, :1608-1698 head);
* the ``nn.functional.dropout`` call on the attention probabilities (``eager_attention_forward``, :458) is patched while a
  forward pre-hook on each ``Wav2Vec2Attention`` says which layer is running;
* LayerDrop: ``torch.rand([])`` (:701 / :774) is patched to return 0.0 (< layerdrop: skip) or 1.0 (keep) per layer from an
  explicit keep list;
* SpecAugment: ``_compute_mask_indices`` (:1294) is patched to return an explicit mask (the span sampler itself is pinned
  separately by tests/golden/specaug.npz).

So WHERE each site sits, what it scales by, and what a skipped layer does in both topologies is decided by the third-party
model; a dropout on the wrong side of a residual, a missing 1 / (1 - p), or activation dropout before instead of after GELU
moves the goldens.  The CPU restatement (``w2v2_ref.forward(train=True, drop=HashDropout(seed))``) is asserted against the
patched HF run here (logits 2e-4, gradients 5e-3 of each tensor's largest element), so the GPU box can compare full tensors.
"""
from __future__ import annotations

import dataclasses
import os
import sys
from unittest import mock

import numpy as np
import torch

from . import dropout_hash as DH
from . import w2v2_ref as R
from .gen_golden import GOLD, base_inputs, synth_wave
from .gen_golden_full import grad_summary


class _HashDropoutModule(torch.nn.Module):
    def __init__(self, p: float, site: int, drop: R.HashDropout):
        super().__init__()
        self.p, self.site, self.drop = p, site, drop

    def forward(self, x):
        if not self.training:
            return x
        return self.drop(x, self.p, self.site)


def _site_of(name: str):
    """HF module name of an nn.Dropout -> engine site id (ssak_amd/csrc/w2v2_engine.hip:127-131)."""
    if name == "wav2vec2.feature_projection.dropout":
        return DH.DS_FEATPROJ
    if name == "wav2vec2.encoder.dropout":
        return DH.DS_ENCIN
    if name == "dropout":
        return DH.DS_FINAL
    parts = name.split(".")
    if parts[:3] == ["wav2vec2", "encoder", "layers"]:
        l = int(parts[3])
        tail = ".".join(parts[4:])
        return {"dropout": DH.ds_hid1(l), "feed_forward.intermediate_dropout": DH.ds_act(l),
                "feed_forward.output_dropout": DH.ds_hid2(l)}[tail]
    raise KeyError(name)


def run_hf_train(cfg: R.W2V2Config, params, x, lengths, labels, seed: int, mask, layer_keep):
    """One training-mode forward + backward of transformers.Wav2Vec2ForCTC with the engine's masks.  Returns loss, logits,
    grads, and the HashDropout log (site, shape, p) in HF's call order."""
    import transformers
    import transformers.models.wav2vec2.modeling_wav2vec2 as MW
    kw = cfg.to_hf_kwargs()
    kw["mask_time_prob"] = max(kw["mask_time_prob"], 0.05)
    hc = transformers.Wav2Vec2Config(**kw)
    hc._attn_implementation = "eager"  # the sdpa path draws its dropout inside the fused op
    m = transformers.Wav2Vec2ForCTC(hc)
    m.load_state_dict(params, strict=True)
    m.freeze_feature_encoder()
    m.train()
    drop = R.HashDropout(seed)
    replaced = []
    for name, mod in list(m.named_modules()):
        if isinstance(mod, torch.nn.Dropout):
            site = _site_of(name)
            parent = m.get_submodule(name.rsplit(".", 1)[0]) if "." in name else m
            setattr(parent, name.rsplit(".", 1)[-1], _HashDropoutModule(mod.p, site, drop))
            replaced.append(name)
    assert len(replaced) == 3 + 3 * cfg.num_hidden_layers, replaced
    state = {"layer": None}
    for l, layer in enumerate(m.wav2vec2.encoder.layers):
        layer.attention.register_forward_pre_hook(lambda mod, args, kwargs=None, l=l: state.__setitem__("layer", l), with_kwargs=False)
        assert abs(layer.attention.dropout - cfg.attention_dropout) < 1e-12

    real_dropout = torch.nn.functional.dropout

    def attn_dropout(inp, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return real_dropout(inp, p, training, inplace)
        assert inp.dim() == 4 and state["layer"] is not None, "an unexpected F.dropout call: every nn.Dropout was replaced"
        return drop(inp, p, DH.ds_attn(state["layer"]), attention=True)

    keep_q = list(layer_keep) if layer_keep is not None else [1] * cfg.num_hidden_layers
    real_rand = torch.rand

    def fake_rand(*size, **kw_):
        if len(size) == 1 and isinstance(size[0], (list, tuple)) and len(size[0]) == 0 and not kw_:
            return torch.tensor(1.0 if keep_q.pop(0) else 0.0)  # LayerDrop draw: skip iff value < layerdrop
        return real_rand(*size, **kw_)

    am = None
    if lengths is not None:
        am = (torch.arange(x.shape[1])[None, :] < torch.tensor(lengths)[:, None]).long()
    layerdrop = 0.5 if layer_keep is not None else 0.0
    m.config.layerdrop = layerdrop
    mask_np = np.asarray(mask, dtype=bool) if mask is not None else np.zeros((x.shape[0], int(R.conv_out_lengths(cfg, x.shape[1]))), bool)
    with mock.patch.object(torch.nn.functional, "dropout", attn_dropout), mock.patch.object(torch, "rand", fake_rand), \
            mock.patch.object(MW, "_compute_mask_indices", lambda *a, **k: mask_np.copy()):
        out = m(torch.tensor(x), attention_mask=am, labels=torch.tensor(labels))
        out.loss.backward()
    assert not keep_q, "LayerDrop draws left over"
    grads = {n: p.grad.detach().numpy() for n, p in m.named_parameters() if p.grad is not None}
    # a skipped layer leaves its parameters without gradient: store zeros so that every trainable name is present
    for n, p in m.named_parameters():
        if p.requires_grad and n not in grads:
            grads[n] = np.zeros(tuple(p.shape), np.float32)
    return out.loss.item(), out.logits.detach().numpy(), grads, drop.log


def check_oracle(cfg, params, x, lengths, labels, seed, mask, layer_keep, hf, tol=2e-4):
    """The CPU restatement with the same masks against the patched HF run (this is what pins it with the regularisers on)."""
    drop = R.HashDropout(seed)
    loss, logits, grads = R.loss_and_grads(params, cfg, torch.tensor(x), lengths, torch.tensor(labels), train=True,
                                           mask_time_indices=None if mask is None else torch.tensor(mask),
                                           layer_keep=layer_keep, drop=drop)
    hl, hlog, hg, hlog_sites = hf
    assert sorted(drop.log) == sorted(hlog_sites), "the restatement and HF disagree on the set of dropout calls"
    assert abs(loss.item() - hl) < tol * max(1, abs(hl)), (loss.item(), hl)
    e = np.abs(logits.numpy() - hlog).max()
    assert e < tol, e
    floor = 1e-3 * max(np.abs(g).max() for g in hg.values())
    worst = 0.0
    for n, g in hg.items():
        e = np.abs(grads[n].numpy() - g).max() / max(np.abs(g).max(), floor)
        worst = max(worst, e)
        assert e < 5e-3, (n, e)
    return worst


TINY_P = 0.25


def tiny_case(xlsr: bool):
    """3-layer tiny config, every dropout at 0.25, middle layer dropped, SpecAugment mask; XLSR: ragged + attention mask."""
    kw = dict(num_hidden_layers=3, attention_dropout=TINY_P, hidden_dropout=TINY_P, activation_dropout=TINY_P,
              feat_proj_dropout=TINY_P, final_dropout=TINY_P, layerdrop=0.5, mask_time_prob=0.2, mask_time_length=3)
    if xlsr:
        kw.update(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)
    cfg = R.W2V2Config.tiny(**kw)
    rng = np.random.default_rng(2025 + int(xlsr))
    lens = [8000, 5000, 6500] if xlsr else None
    x = R.zero_mean_unit_var_norm([synth_wave(rng, n) for n in (lens or [8000] * 3)])
    labels = R.pad_labels([list(rng.integers(1, 32, n)) for n in (9, 4, 6)])
    Fr = int(R.conv_out_lengths(cfg, 8000))
    fl = None if lens is None else R.conv_out_lengths(cfg, lens)
    mask = R.compute_mask_indices((3, Fr), 0.2, 3, fl, 2, rng=np.random.RandomState(3 + int(xlsr)))
    return cfg, R.init_params(cfg, seed=75 + int(xlsr)), x, lens, labels, mask, [1, 0, 1], 0xD1CE5EED0000 + int(xlsr)


def _gen_tiny(xlsr: bool, fname: str):
    cfg, params, x, lens, labels, mask, keep, seed = tiny_case(xlsr)
    hf = run_hf_train(cfg, params, x, lens, labels, seed, mask, keep)
    w = check_oracle(cfg, params, x, lens, labels, seed, mask, keep, hf)
    print(fname, "oracle vs patched HF: worst rel grad err", w, "loss", hf[0], "dropout calls", len(hf[3]))
    # sanity: the masks matter (the same run without them is far away)
    det = R.loss_and_grads(params, dataclasses.replace(cfg).deterministic(), torch.tensor(x), lens, torch.tensor(labels))
    assert np.abs(det[1].numpy() - hf[1]).max() > 0.05
    d = dict(x=x, labels=labels, mask=mask, layer_keep=np.array(keep), seed=np.uint64(seed), loss=np.float32(hf[0]), logits=hf[1],
             p=np.float32(TINY_P))
    if lens is not None:
        d["lens"] = np.array(lens)
    for n, g in hf[2].items():
        d["grad/" + n] = g.astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, fname), **d)


def gen_tiny():
    _gen_tiny(False, "w2v2_tiny_dropout.npz")


def gen_tiny_xlsr():
    _gen_tiny(True, "w2v2_tiny_xlsr_dropout.npz")


BASE_KEEP = [1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1]  # two of twelve layers dropped (LayerDrop 0.1 draws ~1.2 on average)
BASE_SEED = 0x5EED0BA5E


def base_case():
    cfg = R.W2V2Config.base()  # the train script's defaults: every regulariser on
    x, labels = base_inputs()
    Fr = int(R.conv_out_lengths(cfg, x.shape[1]))
    mask = R.compute_mask_indices((x.shape[0], Fr), cfg.mask_time_prob, cfg.mask_time_length, None, cfg.mask_time_min_masks,
                                  rng=np.random.RandomState(77))
    return cfg, x, labels, mask


def gen_base():
    cfg, x, labels, mask = base_case()
    params = R.init_params(cfg, seed=69)
    hf = run_hf_train(cfg, params, x, None, labels, BASE_SEED, mask, BASE_KEEP)
    w = check_oracle(cfg, params, x, None, labels, BASE_SEED, mask, BASE_KEEP, hf)
    print("base, regularisers on: oracle vs patched HF worst rel grad err", w, "loss", hf[0])
    d = dict(labels=labels, mask=mask, layer_keep=np.array(BASE_KEEP), seed=np.uint64(BASE_SEED), loss=np.float32(hf[0]),
             logits=hf[1].astype(np.float32), x_head=x[:, :64])
    d.update(grad_summary({n: hf[2][n] for n in R.trainable_names(cfg)}))
    np.savez_compressed(os.path.join(GOLD, "w2v2_base_dropout.npz"), **d)
    print("w2v2_base_dropout ok")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    for w in sys.argv[1:] or ["tiny", "tiny_xlsr", "base"]:
        globals()["gen_" + w]()


if __name__ == "__main__":
    main()
