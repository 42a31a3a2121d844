"""Full-size golden vectors (round 2).  TEST INFRASTRUCTURE ONLY.

Run in the BUILD container (``python -m oracle.gen_golden_full [base_proj] [base_curve] [xlsr_large] [whisper_small]``).
Like ``gen_golden.py`` it imports the third-party classes the reference calls on its hot path
(``transformers.Wav2Vec2ForCTC`` reached from ssak/train/transformers/wav2vec_train.py:311-329,387-415;
``torch.optim.AdamW`` + ``get_linear_schedule_with_warmup`` + ``clip_grad_norm_`` as HF ``Trainer`` drives them,
docker/transformers_modified/trainer.py:1754-1855) and writes SUMMARY fixtures of the full-size configurations:

* ``w2v2_base.npz``       base config, B=2 x 10 s: logits, loss, per-parameter gradient norms AND three seeded random
                          projections <g, r_k> per parameter (a sign flip / permutation inside a matrix moves them).
* ``w2v2_base_curve.npz`` base config, B=2, 50 optimizer steps, regularisers off: loss, grad-norm and lr per step.
* ``w2v2_xlsr_large.npz`` XLSR-large (24 x 1024, layer-norm feature encoder, stable LN), 2 ragged utterances + mask.
* ``whisper_small.npz``   Whisper-small encoder + CTC head (the build-defined composition), one 30 s window.

Inputs and parameters are regenerated from seeds on the GPU box (``*_inputs()`` below, ``init_params``); the files
hold outputs only.  The projection directions come from ``proj_dirs`` (seeded per parameter index).
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

from . import adamw_ref, logmel_ref
from . import w2v2_ref as R
from .gen_golden import GOLD, base_inputs, check_ref, hf_model, run_hf, synth_wave

NPROJ = 3


def proj_dirs(index: int, numel: int) -> torch.Tensor:
    """[NPROJ, numel] fp32 standard-normal directions of parameter number ``index`` (registration order)."""
    g = torch.Generator().manual_seed(90_000 + index)
    return torch.randn((NPROJ, numel), generator=g)


def grad_summary(grads: dict):
    """names, norms (fp64), heads (first 16), projections <g, r_k> (fp64) of a name -> ndarray gradient dict."""
    names, norms, heads, projs = [], [], [], []
    for i, (n, g) in enumerate(grads.items()):
        g = np.asarray(g)
        flat = torch.tensor(g.reshape(-1), dtype=torch.float64)
        names.append(n)
        norms.append(float(flat.norm()))
        h = np.zeros(16, np.float32)
        h[:min(16, flat.numel())] = g.reshape(-1)[:16]
        heads.append(h)
        projs.append((proj_dirs(i, flat.numel()).double() @ flat).numpy())
    return dict(grad_names=np.array(names), grad_norms=np.array(norms), grad_heads=np.stack(heads),
                grad_projs=np.stack(projs))


# ----------------------------------------------------------------------------- base: one step, projections
def gen_base_proj():
    cfg = R.W2V2Config.base().deterministic()
    params = R.init_params(cfg, seed=69)
    x, labels = base_inputs()
    hf = run_hf(cfg, params, x, None, labels)
    w = check_ref(cfg, params, x, None, labels, hf)
    print("base: oracle vs HF worst rel grad err", w, "loss", hf[0])
    d = dict(labels=labels, loss=np.float32(hf[0]), logits=hf[1].astype(np.float32), wave_seed=np.int64(1234))
    d["x_checksum"] = np.array([x.astype(np.float64).sum(), np.abs(x).astype(np.float64).sum()])
    d["x_head"] = x[:, :64]
    d.update(grad_summary(hf[2]))
    np.savez_compressed(os.path.join(GOLD, "w2v2_base.npz"), **d)
    print("w2v2_base ok")


# ----------------------------------------------------------------------------- base: 50-step matched-loss curve
# the train script's own schedule (wav2vec_train.py:353-384: lr 1e-4, warmup_steps=500, linear decay over the run, clip 1.0)
CURVE = dict(steps=50, lr=1e-4, warmup=500, total=20000, n_batches=4, B=2, T=160000, weight_decay=0.0, max_grad_norm=1.0)
# a 100x faster ramp: the loss falls from 15 to the blank plateau (3.3) within 7 steps through gradient-norm spikes of 166
CURVE_FAST = dict(CURVE, warmup=5, total=50)


def curve_inputs():
    """The 4 x (B=2) batches the curve cycles through: BASELINE.md synthetic utterances, labels of 60..120 symbols."""
    rng = np.random.default_rng(4321)
    batches = []
    for _ in range(CURVE["n_batches"]):
        x = R.zero_mean_unit_var_norm([synth_wave(rng, CURVE["T"]) for _ in range(CURVE["B"])])
        labels = R.pad_labels([list(rng.integers(5, 32, int(rng.integers(60, 121)))) for _ in range(CURVE["B"])])
        batches.append((x, labels))
    return batches


def gen_base_curve_fast():
    gen_base_curve(CURVE_FAST, "w2v2_base_curve_fast.npz")


def gen_base_curve(CURVE=CURVE, fname="w2v2_base_curve.npz"):
    """HF Trainer's inner loop (trainer.py:1754-1855: training_step -> clip_grad_norm_ -> optimizer.step ->
    lr_scheduler.step -> zero_grad) on transformers.Wav2Vec2ForCTC, feature encoder frozen, model.eval() so that the
    dropouts / LayerDrop / SpecAugment are inactive (SURVEY.md section 8d "Matched loss")."""
    import transformers
    cfg = R.W2V2Config.base().deterministic()
    params = R.init_params(cfg, seed=69)
    m = hf_model(cfg, params)
    m.freeze_feature_encoder()
    m.eval()
    train = [p for p in m.parameters() if p.requires_grad]
    # HF parameter groups (trainer.py:1013-1024); identical updates at weight_decay 0
    opt = torch.optim.AdamW(train, lr=CURVE["lr"], betas=(0.9, 0.999), eps=1e-8, weight_decay=CURVE["weight_decay"])
    sch = transformers.get_linear_schedule_with_warmup(opt, CURVE["warmup"], CURVE["total"])
    batches = curve_inputs()
    losses, norms, lrs = [], [], []
    t0 = time.time()
    for step in range(CURVE["steps"]):
        x, labels = batches[step % len(batches)]
        lrs.append(sch.get_last_lr()[0])
        assert abs(lrs[-1] - adamw_ref.linear_warmup_lr(CURVE["lr"], step, CURVE["warmup"], CURVE["total"])) < 1e-12
        out = m(torch.tensor(x), labels=torch.tensor(labels))
        out.loss.backward()
        nrm = torch.nn.utils.clip_grad_norm_(train, CURVE["max_grad_norm"])
        opt.step()
        sch.step()
        opt.zero_grad()
        losses.append(out.loss.item())
        norms.append(nrm.item())
        print(f"step {step} loss {losses[-1]:.5f} gnorm {norms[-1]:.4f} lr {lrs[-1]:.3e} ({time.time() - t0:.0f}s)", flush=True)
    sd = m.state_dict()
    names = [n for n in R.trainable_names(cfg)]
    final = {n: sd[n].numpy() for n in names}
    s = grad_summary(final)  # same summary scheme, applied to the final parameters
    np.savez_compressed(os.path.join(GOLD, fname), loss=np.array(losses), grad_norm=np.array(norms),
                        lr=np.array(lrs), param_names=s["grad_names"], param_norms=s["grad_norms"],
                        param_projs=s["grad_projs"], **{("base_lr" if k == "lr" else k): np.array(v) for k, v in CURVE.items()})
    print(fname, "ok")


# ----------------------------------------------------------------------------- XLSR-large, ragged
XLSR_LENS = (52000, 33333)  # 3.25 s and 2.08 s -> 162 / 103 frames


def xlsr_inputs():
    rng = np.random.default_rng(555)
    x = R.zero_mean_unit_var_norm([synth_wave(rng, n) for n in XLSR_LENS])
    labels = R.pad_labels([list(rng.integers(5, 32, n)) for n in (40, 22)])
    return x, list(XLSR_LENS), labels


def gen_xlsr_large():
    cfg = R.W2V2Config.xlsr_large().deterministic()
    params = R.init_params(cfg, seed=71)
    x, lens, labels = xlsr_inputs()
    hf = run_hf(cfg, params, x, lens, labels)
    w = check_ref(cfg, params, x, lens, labels, hf)
    print("xlsr-large: oracle vs HF worst rel grad err", w, "loss", hf[0])
    fl = R.conv_out_lengths(cfg, lens)
    logits = hf[1].astype(np.float32)
    for b, f in enumerate(fl):  # only valid frames are defined
        logits[b, f:] = 0
    d = dict(lens=np.array(lens), frame_lens=fl, labels=labels, loss=np.float32(hf[0]), logits=logits, x_head=x[:, :64])
    d.update(grad_summary(hf[2]))
    np.savez_compressed(os.path.join(GOLD, "w2v2_xlsr_large.npz"), **d)
    print("w2v2_xlsr_large ok")


# ----------------------------------------------------------------------------- Whisper-small encoder + CTC, one window
def whisper_inputs():
    rng = np.random.default_rng(777)
    wav = synth_wave(rng, 480000)[None]
    labels = R.pad_labels([list(rng.integers(1, 56, 180))])
    return wav, labels


def gen_whisper_small():
    import transformers
    from transformers.models.whisper.modeling_whisper import WhisperEncoder
    from . import whisper_ref as WR
    cfg = WR.WhisperCTCConfig()
    p = WR.init_params(cfg, 73)
    hc = transformers.WhisperConfig(num_mel_bins=cfg.num_mel_bins, d_model=cfg.d_model, encoder_layers=cfg.encoder_layers,
                                    encoder_attention_heads=cfg.encoder_attention_heads, encoder_ffn_dim=cfg.encoder_ffn_dim,
                                    max_source_positions=cfg.max_source_positions, dropout=0.0, attention_dropout=0.0,
                                    activation_dropout=0.0, encoder_layerdrop=0.0, decoder_layers=1, decoder_attention_heads=4,
                                    decoder_ffn_dim=128)
    enc = WhisperEncoder(hc)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in p.items() if k.startswith("encoder.")}, strict=True)
    head = torch.nn.Linear(cfg.d_model, cfg.vocab_size)
    head.weight.data.copy_(p["ctc_head.weight"])
    head.bias.data.copy_(p["ctc_head.bias"])
    wav, labels = whisper_inputs()
    fe = transformers.WhisperFeatureExtractor()
    mel = fe(wav[0], sampling_rate=16000, return_tensors="np").input_features  # [1, 80, 3000]
    assert np.abs(mel[0] - logmel_ref.log_mel(wav[0])).max() < 1e-4
    enc.train(False)
    hs = enc(torch.tensor(mel)).last_hidden_state
    logits = head(hs)
    lm = torch.tensor(labels) >= 0
    lp = torch.nn.functional.log_softmax(logits, dim=-1, dtype=torch.float32).transpose(0, 1)
    loss = torch.nn.functional.ctc_loss(lp, torch.tensor(labels).masked_select(lm), torch.full((1,), hs.shape[1]), lm.sum(-1),
                                        blank=0, reduction="mean", zero_infinity=True)
    loss.backward()
    o_loss, o_logits, o_grads = WR.loss_and_grads(p, cfg, torch.tensor(mel), torch.tensor(labels))
    assert abs(o_loss.item() - loss.item()) < 1e-4 * abs(loss.item())
    assert np.abs(o_logits.numpy() - logits.detach().numpy()).max() < 5e-4, np.abs(o_logits.numpy() - logits.detach().numpy()).max()
    hg = {}
    for n in WR.trainable_names(cfg):
        if n.startswith("encoder."):
            hg[n] = dict(enc.named_parameters())[n[len("encoder."):]].grad
    hg["ctc_head.weight"], hg["ctc_head.bias"] = head.weight.grad, head.bias.grad
    floor = 1e-3 * max(float(g.abs().max()) for g in hg.values())
    worst = 0.0
    for n, g in hg.items():
        e = float((o_grads[n] - g).abs().max()) / max(float(g.abs().max()), floor)
        worst = max(worst, e)
        assert e < 5e-3, (n, e)
    print("whisper-small: oracle vs HF worst rel grad err", worst, "loss", loss.item())
    d = dict(labels=labels, loss=np.float32(loss.item()), logits=logits.detach().numpy().astype(np.float32),
             mel_stride50=mel[0][:, ::50].astype(np.float32), wav_head=wav[:, :64])
    d.update(grad_summary({n: g.numpy() for n, g in hg.items()}))
    np.savez_compressed(os.path.join(GOLD, "whisper_small.npz"), **d)
    print("whisper_small ok")


# ----------------------------------------------------------------------------- HF Trainer's weight-decay group
def gen_decay_names():
    """Names HF Trainer puts in the weight-decayed group (docker/transformers_modified/trainer.py:1013-1024:
    ``get_parameter_names(model, ALL_LAYERNORM_LAYERS)`` minus names containing "bias"), for both topologies."""
    import json
    import transformers
    from transformers.trainer_pt_utils import get_parameter_names
    out = {}
    for key, kw in (("tiny_base", {}), ("tiny_xlsr", dict(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True))):
        cfg = R.W2V2Config.tiny(**kw).deterministic()
        k = cfg.to_hf_kwargs()
        k["mask_time_prob"] = 0.05
        m = transformers.Wav2Vec2ForCTC(transformers.Wav2Vec2Config(**k))
        names = get_parameter_names(m, [torch.nn.LayerNorm])  # ALL_LAYERNORM_LAYERS of the reference's transformers 4.2x
        out[key] = sorted(n for n in names if "bias" not in n)
        assert set(out[key]) <= set(R.param_shapes(cfg))
    with open(os.path.join(GOLD, "decay_names.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("decay_names ok", {k: len(v) for k, v in out.items()})


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    for w in sys.argv[1:] or ["base_proj", "xlsr_large", "whisper_small", "base_curve"]:
        globals()["gen_" + w]()


if __name__ == "__main__":
    main()
