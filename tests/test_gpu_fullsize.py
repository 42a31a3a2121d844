"""GPU parity at the FULL sizes BASELINE.json names (round-2 verdict items 1a-1d): wav2vec2-base (config 2), XLSR-large
with ragged utterances (config 5), Whisper-small encoder + CTC on one 30 s window (config 4), the 50-step base-config
matched-loss curve, and the optimizer kernels against their goldens.

Goldens (tests/golden/*.npz) come from ``oracle/gen_golden_full.py``: transformers.Wav2Vec2ForCTC / WhisperEncoder +
torch.optim.AdamW run in the build container.  Inputs and parameters are re-created from seeds here.  Two kinds of check:

* against the golden summaries made by the THIRD-PARTY classes: logits, loss, per-parameter gradient norms and three seeded
  random projections <g, r_k> per parameter (a sign flip or a permutation inside a matrix moves a projection by ~|g|);
* against the CPU oracle (pinned to those classes by the generator) run here on the host cores: EVERY gradient tensor in
  full, relative L2.

Tolerances are bf16 bars (the engine stores activations in bf16, the reference computes in fp32):
logits rel-L2 <= 2e-2, loss <= 2e-2, gradient tensors rel-L2 <= 6e-2, projections within 6e-2 |g|: the directions are
standard normal, so a projection error is |e| N(0,1) with |e| the tensor's L2 error and the RMS over the three directions
estimates |e| / |g| (chi distributed, 3 degrees of freedom): the engine's measured 1-2 % L2 error stays below 6e-2 (8e-2 for
the 24-layer XLSR-large, where bf16 rounding accumulates over twice the depth), while a sign flip or permutation inside the
tensor moves every projection by ~|g| N(0,1).
"""
import dataclasses

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).reshape(-1)
    b = torch.as_tensor(b, dtype=torch.float64).reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-12))


def _cfg(Wav2Vec2Config, oc):
    d = dataclasses.asdict(oc)
    d.pop("initializer_range")
    return Wav2Vec2Config(**d)


def _check_summary(model, z, tol_norm=6e-2, tol_proj=6e-2):
    """Golden gradient norms + random projections (made with the third-party classes) against the engine's gradients."""
    from oracle.gen_golden_full import proj_dirs
    gmax = float(z["grad_norms"].max())
    worst_n, worst_p = 0.0, 0.0
    for i, (n, nr, pr) in enumerate(zip(z["grad_names"], z["grad_norms"], z["grad_projs"])):
        g = model.grad(str(n)).double().reshape(-1).cpu()
        if str(n) in model._HEAD:  # inert padding classes are not part of the contract
            g = model.grad(str(n))[:model.config.vocab_size].double().reshape(-1).cpu()
        if nr < 1e-4 * gmax:  # numerically-zero gradients (k_proj.bias): only rounding noise on both sides
            assert float(g.norm()) < 1e-3 * gmax, n
            continue
        e = abs(float(g.norm()) - nr) / nr
        worst_n = max(worst_n, e)
        assert e < tol_norm, (str(n), "norm", float(g.norm()), nr)
        got = (proj_dirs(i, g.numel()).double() @ g).numpy()
        ep = float(np.abs(got - pr).max()) / nr
        worst_p = max(worst_p, ep)
        assert ep < tol_proj, (str(n), "projection", got, pr, nr)
    return worst_n, worst_p


def _check_full(model, ref_grads, tol=6e-2):
    gmax = max(float(g.abs().max()) for g in ref_grads.values())
    worst = ("", 0.0)
    for n, g in ref_grads.items():
        got = model.grad(n).cpu()
        if n in model._HEAD:
            got = got[:model.config.vocab_size]
        if float(g.abs().max()) < 2e-4 * gmax:
            assert float((got - g).abs().max()) < 1e-3 * gmax, n
            continue
        e = rel_l2(got, g)
        if e > worst[1]:
            worst = (n, e)
        assert e < tol, (n, e)
    return worst


# ------------------------------------------------------------------------------------------------ config 2
def test_base_gradients_projections_and_full_tensors(gold):
    """wav2vec2-base, B=2 x 10 s: gradient norms + 3 random projections per parameter vs the transformers golden, then
    every gradient tensor in full vs the CPU oracle (run here; pinned to transformers at 6e-5 by the generator)."""
    from oracle import w2v2_ref as R
    from oracle.gen_golden import base_inputs
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    z = gold("w2v2_base.npz")
    x, labels = base_inputs()
    assert np.abs(x[:, :64] - z["x_head"]).max() < 1e-6
    oc = R.W2V2Config.base().deterministic()
    params = R.init_params(oc, 69)
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    model.load_state_dict(params)
    out = model(torch.tensor(x), labels=torch.tensor(labels))
    assert rel_l2(out.logits.cpu(), z["logits"]) < 2e-2
    assert abs(out.loss.item() - float(z["loss"])) < 2e-2 * float(z["loss"])
    model.grads[:model.num_trainable].fill_(float("nan"))
    model.backward()
    wn, wp = _check_summary(model, z)
    print("base vs transformers golden: worst norm err", wn, "worst projection err / |g|", wp)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    loss, logits, grads = R.loss_and_grads(params, oc, torch.tensor(x), None, torch.tensor(labels))
    assert abs(loss.item() - float(z["loss"])) < 1e-4 * float(z["loss"])  # the oracle run here = the golden
    worst = _check_full(model, grads)
    print("base vs oracle, full tensors: worst", worst)


# ------------------------------------------------------------------------------------------------ config 5
def test_xlsr_large_ragged_vs_hf_golden(gold):
    """Wav2Vec2-large-XLSR topology at FULL size (24 x 1024, 16 heads, layer-norm feature encoder with bias, stable LN),
    two ragged utterances with attention mask: logits on valid frames, loss, gradient norms + projections vs transformers."""
    from oracle import w2v2_ref as R
    from oracle.gen_golden_full import xlsr_inputs
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    z = gold("w2v2_xlsr_large.npz")
    x, lens, labels = xlsr_inputs()
    assert np.abs(x[:, :64] - z["x_head"]).max() < 1e-6
    oc = R.W2V2Config.xlsr_large().deterministic()
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    model.load_state_dict(R.init_params(oc, 71))
    out = model(torch.tensor(x), lengths=torch.tensor(lens), labels=torch.tensor(labels))
    fl = z["frame_lens"]
    assert (out.frame_lens.cpu().numpy() == fl).all()
    for b in range(len(lens)):
        assert rel_l2(out.logits[b, :fl[b]].cpu(), z["logits"][b, :fl[b]]) < 2e-2
    assert abs(out.loss.item() - float(z["loss"])) < 2e-2 * float(z["loss"])
    model.grads[:model.num_trainable].fill_(float("nan"))
    model.backward()
    wn, wp = _check_summary(model, z, tol_proj=8e-2)
    print("xlsr-large vs transformers golden: worst norm err", wn, "worst projection err / |g|", wp)


# ------------------------------------------------------------------------------------------------ config 4
def test_whisper_small_window_vs_hf_golden(gold):
    """Whisper-small encoder (12 x 768, 1500 positions) + CTC head on ONE full 30 s window, log-mel from the HIP front end:
    features, logits, loss, gradient norms + projections vs transformers.WhisperEncoder (the composition is the build's,
    SURVEY.md section 0)."""
    import ssak_amd.hip as hip
    from oracle import whisper_ref as WR
    from oracle.gen_golden_full import whisper_inputs
    from ssak_amd.whisper import WhisperCTCConfig, WhisperEncoderForCTC
    z = gold("whisper_small.npz")
    wav, labels = whisper_inputs()
    assert np.abs(wav[:, :64] - z["wav_head"]).max() < 1e-7
    oc = WR.WhisperCTCConfig()
    model = WhisperEncoderForCTC(WhisperCTCConfig(vocab_size=oc.vocab_size)).train()
    model.load_state_dict(WR.init_params(oc, 73))
    mel = hip.logmel_whisper(torch.tensor(wav).cuda())
    assert mel.shape == (1, 80, 3000)
    assert np.abs(mel[0, :, ::50].cpu().numpy() - z["mel_stride50"]).max() < 2e-4
    out = model(mel, labels=torch.tensor(labels))
    assert out.logits.shape == (1, 1500, oc.vocab_size)
    assert rel_l2(out.logits.cpu(), z["logits"]) < 2e-2
    assert abs(out.loss.item() - float(z["loss"])) < 2e-2 * float(z["loss"])
    model.grads[:model.num_trainable].fill_(float("nan"))
    model.backward()
    wn, wp = _check_summary(model, z)
    print("whisper-small vs transformers golden: worst norm err", wn, "worst projection err / |g|", wp)


# ------------------------------------------------------------------------------------------------ a11
@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_adamw_kernels_vs_torch_golden(gold, wd):
    """ssak_grad_sumsq + ssak_adamw_step against tests/golden/adamw.npz (torch.optim.AdamW + clip_grad_norm_(1.0) +
    get_linear_schedule_with_warmup(2, 10), three steps with gradient norms ~31, ~0.3 and ~95: clipped, unclipped, clipped)."""
    import ssak_amd.hip as hip
    from oracle.adamw_ref import linear_warmup_lr
    z = gold("adamw.npz")
    dev = "cuda:0"
    p = torch.tensor(z["p0"]).to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    shadow = torch.zeros(p.numel(), dtype=torch.bfloat16, device=dev)
    nsq = torch.zeros(1, device=dev)
    ws = torch.empty(1024, device=dev)
    for i, g in enumerate(z["grads"]):
        gd = torch.tensor(g).to(dev)
        hip.check(hip.lib.ssak_grad_sumsq(hip.ptr(gd), gd.numel(), hip.ptr(nsq), hip.ptr(ws), 4096, hip.stream()))
        lr = linear_warmup_lr(1e-4, i, 2, 10)
        hip.check(hip.lib.ssak_adamw_step(hip.ptr(p), hip.ptr(gd), hip.ptr(m), hip.ptr(v), hip.ptr(shadow), p.numel(), hip.ptr(nsq),
                                          1.0, 1.0, lr, 0.9, 0.999, 1e-8, wd, i + 1, hip.stream()))
        assert abs(float(nsq.sqrt().item()) - float(z[f"norm_wd{wd}"][i])) < 1e-5 * float(z[f"norm_wd{wd}"][i])
        ref = z[f"p_wd{wd}"][i]
        err = float((p.cpu() - torch.tensor(ref)).abs().max())
        assert err < 5e-7, (i, err)  # parameters up to ~4 (fp32 ulp 2.4e-7), updates ~1e-4: two ulps = 5e-3 of an update
        assert torch.equal(shadow.float().cpu(), p.cpu().bfloat16().float())


def test_weight_decay_groups_match_hf_trainer(gold_json):
    """--weight_decay > 0: HF Trainer decays every parameter outside nn.LayerNorm modules whose name has no "bias"
    (docker/transformers_modified/trainer.py:1013-1024).  The decayed NAMES are pinned by tests/golden/decay_names.json (made
    with transformers' get_parameter_names on Wav2Vec2ForCTC); here one optimizer step with and one without decay from the
    same state: the difference must be exactly -lr * wd * p on the decayed tensors and nothing elsewhere."""
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer, decay_ranges
    for topo, kw in (("tiny_base", {}), ("tiny_xlsr", dict(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True))):
        for freeze in (True, False):
            oc = R.W2V2Config.tiny(**kw).deterministic()
            p0 = R.init_params(oc, 31)
            rng = np.random.default_rng(5)
            x = torch.tensor(R.zero_mean_unit_var_norm([rng.standard_normal(8000).astype(np.float32) for _ in range(2)])).cuda()
            labels = torch.tensor(R.pad_labels([[3, 4, 5, 6], [7, 8]])).cuda()
            lens = torch.tensor([8000, 8000]).cuda() if kw else None
            wd, lr = 0.1, 1e-3
            outs = []
            for w in (0.0, wd):
                model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc), freeze_feature_encoder=freeze).train()
                model.load_state_dict(p0)
                tr = Trainer(model, AdamW(model, lr=lr, weight_decay=w, warmup_steps=0, total_steps=10 ** 9, max_grad_norm=1.0))
                tr.train_step(x, lens, labels, raw=False)
                outs.append(model.state_dict())
                rg = decay_ranges(model)
                if freeze:  # matrices first, vectors after: two ranges
                    assert len(rg) == 2 and rg[0][2] and not rg[1][2] and rg[0][0] == 0 and rg[0][1] + rg[1][1] == model.num_trainable
            decayed = set(gold_json("decay_names.json")[topo])
            trainable = R.trainable_names(oc, freeze_feature_encoder=freeze)
            assert any(n.startswith("wav2vec2.feature_extractor") for n in trainable) == (not freeze)
            for n in trainable:
                delta = outs[1][n] - outs[0][n]
                want = -lr * wd * p0[n] if n in decayed else torch.zeros_like(p0[n])
                assert float((delta - want).abs().max()) < 2e-7 + 1e-6 * float(p0[n].abs().max()), (topo, freeze, n)


# ------------------------------------------------------------------------------------------------ matched loss, base config
@pytest.mark.parametrize("which", ["script_schedule", "fast_ramp"])
def test_base_matched_loss_50_steps_vs_hf_curve(gold, which):
    """SURVEY.md section 8d "Matched loss" on the HEADLINE config: wav2vec2-base, B=2 x 10 s, 50 optimizer steps of HF
    Trainer's inner loop, regularisers off -- golden curves made with transformers.Wav2Vec2ForCTC + torch.optim.AdamW +
    get_linear_schedule_with_warmup + clip_grad_norm_(1.0) (oracle/gen_golden_full.py).

    * script_schedule: the train script's own schedule (lr 1e-4, warmup_steps=500, wav2vec_train.py:353-384; lr reaches 1e-5 in
      50 steps, the loss falls 14.5 -> 5.8).  The bf16 HIP curve must stay within 2e-2 relative AT EVERY STEP.
    * fast_ramp: a 100x faster ramp (5 warm-up steps to 1e-4): the loss collapses to the blank plateau (3.3) within 7 steps
      through gradient-norm spikes of 166.  Through the descent (steps 0-12) the curves must agree to 5e-3 per step (measured:
      1.3e-3); on the plateau the fp32 reference itself scatters by +-2 % from step to step (its gradient norm jumps between 2
      and 12) and the trajectory is chaotic: replacing ONE weight-gradient kernel by another that agrees with it to 2e-7
      relative (fp32 summation order; tests/dev_pcw_check.py) moves the largest per-step deviation from 1.8e-2 (step 26) to 5.4e-2
      (step 17).  So there the per-step bar is 8e-2 and the bar that means something is 1e-2 on the means of 10-step windows."""
    from oracle import w2v2_ref as R
    from oracle.gen_golden_full import curve_inputs
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer
    z = gold("w2v2_base_curve.npz" if which == "script_schedule" else "w2v2_base_curve_fast.npz")
    steps = int(z["steps"])
    assert steps == 50 == len(z["loss"])
    oc = R.W2V2Config.base().deterministic()
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    model.load_state_dict(R.init_params(oc, 69))
    opt = AdamW(model, lr=float(z["base_lr"]), warmup_steps=int(z["warmup"]), total_steps=int(z["total"]),
                weight_decay=float(z["weight_decay"]), max_grad_norm=float(z["max_grad_norm"]))
    tr = Trainer(model, opt)
    batches = [(torch.tensor(x).cuda(), torch.tensor(l).cuda()) for x, l in curve_inputs()]
    got, norms = [], []
    for s in range(steps):
        assert abs(opt.current_lr() - float(z["lr"][s])) < 1e-12
        x, l = batches[s % len(batches)]
        loss = tr.train_step(x, None, l, raw=False)
        got.append(float(loss.item()))
        norms.append(opt.grad_norm())
    got, ref = np.array(got), z["loss"]
    rel = np.abs(got - ref) / np.abs(ref)
    print(which, "matched loss: first", got[0], ref[0], "last", got[-1], ref[-1], "max rel", rel.max(), "at step", int(rel.argmax()))
    assert ref[-8:].mean() < 0.6 * ref[:8].mean()  # the reference run actually learns (two passes over the 4 batches each)
    if which == "script_schedule":
        assert rel.max() < 2e-2
        gn = np.abs(np.array(norms) - z["grad_norm"]) / z["grad_norm"]
        print("grad-norm rel err: max", float(gn.max()))
        assert gn.max() < 0.15  # (the clip norm of a 90 M-element bf16-computed gradient)
        # where the 50 updates went: norms of the final parameters
        sd = model.state_dict()
        for n, nr in zip(z["param_names"], z["param_norms"]):
            t = sd[str(n)].double().reshape(-1)
            assert abs(float(t.norm()) - nr) < 2e-3 * nr + 1e-6, (str(n), float(t.norm()), nr)
    else:
        assert rel[:13].max() < 5e-3 and rel.max() < 8e-2
        for w in range(20, 50, 10):
            assert abs(got[w:w + 10].mean() - ref[w:w + 10].mean()) < 1e-2 * ref[w:w + 10].mean(), w
