"""GPU: the SpeechBrain-recipe head and step through the C ABI against oracle/sb_head_ref.py (plain torch
fp32).  Tolerances: activations and gradients pass through bf16 storage (8 significant bits) with fp32 accumulation and fp32
statistics, so tensors are compared by relative L2 error -- 1e-2 for single kernels, 3e-2 for the chained head's logits, 6e-2
for gradients through the whole encoder (the bound of tests/test_gpu_model.py).  The head's weight gradients at random
initialisation are ill-conditioned with respect to the forward values (see oracle/sb_head_ref.py: head_forward): they are
compared at 3e-2 against the oracle evaluated with the device's storage roundings and at 2e-1 against the plain fp32 oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-12))


@pytest.fixture(scope="module")
def hip():
    from ssak_amd import hip as h
    return h


def _ws(n):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device="cuda")


@pytest.mark.parametrize("dtype,B,n", [(torch.float32, 3, 16000), (torch.float32, 2, 4004), (torch.bfloat16, 4, 49 * 64),
                                       (torch.bfloat16, 2, 499 * 1024), (torch.float32, 3, 23457), (torch.bfloat16, 2, 1003)])
def test_utt_norm_fwd_bwd(hip, dtype, B, n):
    g = torch.Generator().manual_seed(n)
    x = (torch.randn(B, n, generator=g) * 0.3 + 0.1).to(dtype)
    dy = torch.randn(B, n, generator=g).to(dtype)
    xr = x.float().clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (n,), eps=1e-5)
    xd, dyd = x.cuda(), dy.cuda()
    y, dx = torch.empty_like(xd), torch.empty_like(xd)
    stats = torch.empty(B, 2, device="cuda")
    ws = _ws(hip.lib.ssak_utt_norm_workspace_bytes(B))
    isb = int(dtype == torch.bfloat16)
    hip.check(hip.lib.ssak_utt_norm_fwd(hip.ptr(xd), hip.ptr(y), B, n, isb, 1e-5, hip.ptr(stats), hip.ptr(ws), ws.numel(), hip.stream()))
    tol = 1e-5 if dtype == torch.float32 else 4e-3
    assert rel_l2(y.float().cpu(), yr.detach()) < tol
    assert np.allclose(stats[:, 0].cpu(), x.float().mean(1), atol=1e-5)
    assert np.allclose(stats[:, 1].cpu(), 1 / np.sqrt(x.float().var(1, unbiased=False) + 1e-5), rtol=1e-4)
    # the backward takes the forward's OUTPUT (as stored): reference gradient evaluated at the same stored y
    hip.check(hip.lib.ssak_utt_norm_bwd(hip.ptr(dyd), hip.ptr(y), hip.ptr(dx), B, n, isb, hip.ptr(stats), hip.ptr(ws), ws.numel(),
                                        hip.stream()))
    yr.backward(dy.float())
    assert rel_l2(dx.float().cpu(), xr.grad) < (1e-4 if dtype == torch.float32 else 8e-3)


def _bn_fwd(hip, a, gamma, beta, rm, rv, training, p, seed, stream_id):
    M, Cc = a.shape
    y = torch.empty_like(a)
    mean, rstd = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    ws = _ws(hip.lib.ssak_batchnorm_workspace_bytes(Cc))
    hip.check(hip.lib.ssak_batchnorm_act_fwd(hip.ptr(a), hip.ptr(y), M, Cc, hip.ptr(gamma), hip.ptr(beta), hip.ptr(rm), hip.ptr(rv),
                                             0.1, 1e-5, int(training), 0.01, p, C.c_uint64(seed), stream_id, hip.ptr(mean),
                                             hip.ptr(rstd), None, hip.ptr(ws), ws.numel(), hip.stream()))
    return y, mean, rstd


@pytest.mark.parametrize("M,Cc,p", [(700, 64, 0.0), (3 * 499, 1024, 0.0), (1234, 256, 0.15), (5, 8, 0.0)])
def test_batchnorm_act_fwd_bwd(hip, M, Cc, p):
    g = torch.Generator().manual_seed(M + Cc)
    a = (torch.randn(M, Cc, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    dy = torch.randn(M, Cc, generator=g).to(torch.bfloat16)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.2
    rm0, rv0 = torch.randn(Cc, generator=g) * 0.1, torch.rand(Cc, generator=g) + 0.5
    rm, rv = rm0.clone().cuda(), rv0.clone().cuda()
    ad, gd, bd = a.cuda(), gamma.cuda(), beta.cuda()
    y, mean, rstd = _bn_fwd(hip, ad, gd, bd, rm, rv, True, p, 77, 3)
    mask = None
    if p > 0:  # the mask is a function of (seed, stream, element index): recover it by comparing with the p = 0 output
        y0, _, _ = _bn_fwd(hip, ad, gd, bd, rm0.clone().cuda(), rv0.clone().cuda(), True, 0.0, 77, 3)
        mask = (y != 0) | (y0 == 0)
        keep = mask.float().mean().item()
        assert abs(keep - (1 - p)) < 0.01
        assert torch.allclose(y.float(), (y0.float() * mask / (1 - round(p * 65536) / 65536)).to(torch.bfloat16).float(), rtol=8e-3, atol=1e-6)
        y2, _, _ = _bn_fwd(hip, ad, gd, bd, rm0.clone().cuda(), rv0.clone().cuda(), True, p, 77, 3)
        assert torch.equal(y, y2)  # replayable
        mask = mask.cpu()
    # reference: nn.BatchNorm1d semantics over the M rows
    ar = a.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rmr, rvr = rm0.clone(), rv0.clone()
    z = torch.nn.functional.batch_norm(ar, rmr, rvr, gr, br, True, 0.1, 1e-5)
    yr = torch.nn.functional.leaky_relu(z, 0.01)
    if mask is not None:
        yr = yr * mask / (1 - p)
    assert rel_l2(y.float().cpu(), yr.detach()) < 6e-3
    assert np.allclose(mean.cpu(), a.float().mean(0), atol=1e-5)
    if M > 1:
        assert np.allclose(rm.cpu(), rmr, atol=1e-5) and np.allclose(rv.cpu(), rvr, rtol=1e-4, atol=1e-6)
    yr.backward(dy.float())
    da = torch.empty_like(ad)
    dg, db = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    ws = _ws(hip.lib.ssak_batchnorm_workspace_bytes(Cc))
    hip.check(hip.lib.ssak_batchnorm_act_bwd(hip.ptr(dy.cuda()), hip.ptr(ad), hip.ptr(da), M, Cc, hip.ptr(gd), hip.ptr(bd), hip.ptr(mean),
                                             hip.ptr(rstd), 0.01, p, C.c_uint64(77), 3, hip.ptr(dg), hip.ptr(db), None, None, hip.ptr(ws),
                                             ws.numel(), hip.stream()))
    assert rel_l2(da.float().cpu(), ar.grad) < 8e-3
    assert rel_l2(dg.cpu(), gr.grad) < 2e-3 and rel_l2(db.cpu(), br.grad) < 2e-3
    # evaluation mode: running statistics, no dropout
    ye, _, _ = _bn_fwd(hip, ad, gd, bd, rm, rv, False, p, 77, 3)
    ze = torch.nn.functional.leaky_relu(torch.nn.functional.batch_norm(a.float(), rm.cpu(), rv.cpu(), gamma, beta, False, 0.1, 1e-5), 0.01)
    assert rel_l2(ye.float().cpu(), ze) < 6e-3


def test_adadelta_vs_torch(hip):
    n = 10007
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(n, generator=g)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adadelta([pt], lr=1.0, rho=0.95, eps=1e-8)
    p, sq, acc = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    shadow = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    gn = torch.zeros(1, device="cuda")
    ws = torch.empty(1024, device="cuda")
    for step in range(6):
        gr = torch.randn(n, generator=g) * (30.0 if step == 2 else 0.01)  # step 2 exceeds the clip norm
        pt.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_([pt], 5.0)
        opt.step()
        gd = gr.cuda()
        hip.check(hip.lib.ssak_grad_sumsq(hip.ptr(gd), n, hip.ptr(gn), hip.ptr(ws), 4096, hip.stream()))
        hip.check(hip.lib.ssak_adadelta_step(hip.ptr(p), hip.ptr(gd), hip.ptr(sq), hip.ptr(acc), hip.ptr(shadow), n, hip.ptr(gn), 5.0,
                                             1.0, 1.0, 0.95, 1e-8, 0.0, hip.stream()))
    assert np.allclose(p.cpu(), pt.detach(), rtol=2e-5, atol=2e-6)
    assert torch.equal(shadow.cpu(), p.cpu().to(torch.bfloat16))
    # the joint norm over two buffers
    a, b = torch.randn(5000, generator=g).cuda(), torch.randn(3000, generator=g).cuda()
    hip.check(hip.lib.ssak_grad_sumsq(hip.ptr(a), 5000, hip.ptr(gn), hip.ptr(ws), 4096, hip.stream()))
    hip.check(hip.lib.ssak_grad_sumsq_add(hip.ptr(b), 3000, hip.ptr(gn), hip.ptr(ws), 4096, hip.stream()))
    assert np.isclose(gn.item(), float((a.double() ** 2).sum() + (b.double() ** 2).sum()), rtol=1e-5)


def _small_head(dropouts=(0.0, 0.0, 0.0), H=64, D=64, V=13):
    from ssak_amd.sb_head import CTCHead
    return CTCHead(H, D, V, dropouts=dropouts, seed=3)


@pytest.mark.parametrize("dropouts", [(0.0, 0.0, 0.0), (0.15, 0.15, 0.0)])
def test_head_forward_backward_vs_oracle(hip, dropouts):
    from oracle import sb_head_ref as S
    B, Fr, H, D, V = 3, 50, 64, 64, 13
    head = _small_head(dropouts, H, D, V)
    sd = {k: v for k, v in head.state_dict().items() if "running" not in k and "tracked" not in k}
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(B, Fr, H, generator=g).to(torch.bfloat16)
    tokens = torch.randint(1, V, (B, 9), generator=g)
    wav_lens, tok_lens = torch.tensor([1.0, 0.8, 0.63]), torch.tensor([1.0, 0.5, 0.7])
    head.train()
    logits = head(feats.cuda())
    saved = head._saved[0]
    masks = None
    if any(dropouts):  # y_i is the next block's GEMM operand: zero exactly where the element was dropped
        ys = [saved[1][0], saved[2][0], head._saved[1]]
        masks = [(y != 0).view(B, Fr, D).cpu() if p > 0 else None for y, p in zip(ys, dropouts)]
    in_lens = torch.round(wav_lens * Fr).int()
    tl = torch.round(tok_lens * 9).int()
    labels = torch.where(torch.arange(9)[None] < tl[:, None], tokens, torch.full_like(tokens, -1))
    loss, nll, dlogits = hip.ctc_loss(logits, in_lens, labels, 0, "mean", True, 1.0)
    dfeats = head.backward(dlogits, need_input_grad=True)
    rl, rlogits, rg32, _ = S.head_loss_and_grads(sd, feats.float(), tokens, wav_lens, tok_lens, masks, dropouts=dropouts)
    assert rel_l2(logits[..., :V].cpu(), rlogits) < 3e-2
    assert abs(loss.item() - rl.item()) < 2e-2 * rl.item()
    _, _, rg, rdf = S.head_loss_and_grads(sd, feats.float(), tokens, wav_lens, tok_lens, masks, dropouts=dropouts, bf16_storage=True)
    gmax = max(float(v.abs().max()) for v in rg.values())
    for n, r in rg.items():
        got = head.grad(n).cpu()
        if n.startswith("1.w."):
            assert float(got[V:].abs().max()) == 0.0  # the inert padding classes
            got = got[:V]
        if float(r.abs().max()) < 1e-4 * gmax:  # a Linear bias in front of BatchNorm has zero gradient
            assert float((got - r).abs().max()) < 2e-3 * gmax, n
        else:
            # 3e-2 without dropout; with it the 150 x 64 mask is one draw of the engine's generator and the small vectors (a BatchNorm
            # bias: 64 values) move by a few tenths of a per cent with the draw -- the round-5 masks read 3.07e-2 on one of them
            assert rel_l2(got, r) < (4e-2 if any(dropouts) else 3e-2), (n, rel_l2(got, r))
            assert rel_l2(got, rg32[n]) < 2e-1, (n, rel_l2(got, rg32[n]))
    assert rel_l2(dfeats.float().cpu(), rdf) < (4e-2 if any(dropouts) else 3e-2)  # (one draw of the masks: see above; 3.1e-2 with round 5's)
    # evaluation mode uses the running statistics the training pass just updated
    head.eval()
    le = head(feats.cuda())
    run = [(head.running_mean[i].cpu(), head.running_var[i].cpu()) for i in range(3)]
    re_ = S.head_forward(sd, feats.float(), None, False, dropouts=dropouts, running=run)
    assert rel_l2(le[..., :V].cpu(), re_) < 3e-2


@pytest.fixture(scope="module")
def tiny():
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from oracle import w2v2_ref as R
    import dataclasses
    oc = R.W2V2Config.tiny().deterministic()
    d = dataclasses.asdict(oc)
    d.pop("initializer_range")
    return oc, Wav2Vec2Config(**d), Wav2Vec2ForCTC, R


def test_forward_hidden_vs_oracle(tiny):
    oc, cfg, Wav2Vec2ForCTC, R = tiny
    p = R.init_params(oc, 11)
    model = Wav2Vec2ForCTC(cfg)
    model.load_state_dict(p)
    x = torch.randn(3, 6000, generator=torch.Generator().manual_seed(2))
    st = {}
    _, rlogits = R.forward(p, oc, x, stages=st)
    model.eval()
    hidden, _ = model.forward_hidden(x)
    assert hidden.dtype == torch.bfloat16 and hidden.shape == st["last_hidden"].shape
    assert rel_l2(hidden.float().cpu(), st["last_hidden"]) < 2e-2
    with pytest.raises(RuntimeError):
        model.backward_hidden(torch.zeros_like(hidden))  # no training-mode forward_hidden
    model.train()
    model(x, labels=torch.tensor([[1, 2], [3, 4], [5, 6]]))
    with pytest.raises(RuntimeError):
        model.backward_hidden(torch.zeros_like(hidden))  # the kept forward ended at the logits


def test_recipe_unfrozen_step_vs_oracle(tiny):
    """The unfrozen recipe, link by link and then as one fit_batch:
    (1) features = layer_norm(Wav2Vec2Model(layer_norm(wav))[0]) against the oracle;
    (2) loss, head gradients and d loss / d features against the oracle's autograd on the device's features;
    (3) the wav2vec2 gradients against the oracle's autograd through layer_norm(out) and the whole encoder with the device's
        d loss / d features injected (a linear functional of the features, so the comparison does not inherit (2)'s conditioning);
    (4) fit_batch reproduces exactly these gradients, then Adam / Adadelta move both parameter sets."""
    from ssak_amd.sb_head import Brain, CTCHead, TRAIN
    from oracle import sb_head_ref as S
    oc, cfg, Wav2Vec2ForCTC, R = tiny
    p = R.init_params(oc, 5)
    model = Wav2Vec2ForCTC(cfg)
    model.load_state_dict(p)
    head = CTCHead(cfg.hidden_size, 64, 13, dropouts=(0.0, 0.0, 0.0), seed=4)
    sd = {k: v for k, v in head.state_dict().items() if "running" not in k and "tracked" not in k}
    brain = Brain(model, head, freeze_wav2vec=False, lr=1.0, lr_wav2vec=1e-4)
    g = torch.Generator().manual_seed(8)
    wavs = torch.randn(3, 8000, generator=g) * 0.1 + 0.02
    tokens = torch.randint(1, 13, (3, 6), generator=g)
    wav_lens, tok_lens = torch.tensor([1.0, 0.9, 0.75]), torch.tensor([1.0, 0.5, 0.67])
    # (1) + (2)
    outputs = brain.compute_forward(wavs, wav_lens, TRAIN)
    loss = brain.compute_objectives(outputs, tokens, tok_lens, TRAIN)
    feats, stats = brain._fwd
    st = {}
    R.forward(p, oc, S.utt_norm(wavs), train=True, stages=st)
    assert rel_l2(feats.float().cpu(), S.utt_norm(st["last_hidden"])) < 2e-2
    dfeats = head.backward(brain._dlogits, need_input_grad=True)
    rl, _, hg, rdf = S.head_loss_and_grads(sd, feats.float().cpu(), tokens, wav_lens, tok_lens, None, dropouts=(0, 0, 0),
                                           bf16_storage=True)
    assert abs(loss.item() - rl.item()) < 2e-2 * rl.item()
    gmax = max(float(v.abs().max()) for v in hg.values())
    for n, r in hg.items():
        got = head.grad(n).cpu()[:r.shape[0]]
        if float(r.abs().max()) < 1e-4 * gmax:
            assert float((got - r).abs().max()) < 2e-3 * gmax, n
        else:
            assert rel_l2(got, r) < 3e-2, (n, rel_l2(got, r))
    assert rel_l2(dfeats.float().cpu(), rdf) < 3e-2
    head_grads = head.grads.clone()
    # (3)
    brain.backward_encoder(dfeats, feats, stats)
    names = [n for n in R.trainable_names(oc, True) if not n.startswith("lm_head")]
    q = {n: (t.detach().clone().requires_grad_(True) if n in names else t.detach()) for n, t in p.items()}
    st = {}
    R.forward(q, oc, S.utt_norm(wavs), train=True, stages=st)
    (S.utt_norm(st["last_hidden"]) * dfeats.float().cpu()).sum().backward()
    wg = {n: (q[n].grad if q[n].grad is not None else torch.zeros_like(q[n])) for n in names}
    gmax = max(float(v.abs().max()) for v in wg.values())
    worst = ("", 0.0)
    for n, r in wg.items():
        got = model.grad(n).cpu()
        if float(r.abs().max()) < 2e-4 * gmax:
            assert float((got - r).abs().max()) < 1e-3 * gmax, n
            continue
        e = rel_l2(got, r)
        worst = max(worst, (n, e), key=lambda t: t[1])
        assert e < 6e-2, (n, e)
    print("worst wav2vec2 gradient", worst)
    assert float(model.grad("lm_head.weight").abs().max()) == 0.0
    enc_grads = model.grads.clone()
    # (4) the same batch through fit_batch (dropout-free configuration: the step is a function of the data only)
    p_before, h_before = model.params.clone(), head.params.clone()
    loss2 = brain.fit_batch(wavs, wav_lens, tokens, tok_lens)
    assert loss2.item() == loss.item()
    assert torch.equal(head.grads, head_grads) and torch.equal(model.grads, enc_grads)
    assert not torch.equal(model.params, p_before) and not torch.equal(head.params, h_before)
    assert brain.wav2vec_optimizer.step_count == 1 and brain.optimizer_step == 1
    # Adam's first step moves every parameter with a non-zero gradient by lr (bias-corrected m / sqrt(v) = sign(g))
    moved = (model.params - p_before)[:model.num_trainable]
    nz = enc_grads[:model.num_trainable].abs() > 1e-6
    assert torch.allclose(moved[nz].abs(), torch.full_like(moved[nz], 1e-4), rtol=2e-2)


def test_recipe_frozen_steps_vs_oracle(tiny):
    """The default recipe (freeze_wav2vec: True): four Adadelta steps of the head on fixed features against the oracle loop
    (torch autograd + clip_grad_norm_(5.0) + torch.optim.Adadelta), then NewBob annealing at the validation stage."""
    from ssak_amd.sb_head import Brain, CTCHead, VALID
    from oracle import sb_head_ref as S
    oc, cfg, Wav2Vec2ForCTC, R = tiny
    p = R.init_params(oc, 6)
    model = Wav2Vec2ForCTC(cfg)
    model.load_state_dict(p)
    head = CTCHead(cfg.hidden_size, 64, 13, dropouts=(0.0, 0.0, 0.0), seed=9)
    sd = {k: v for k, v in head.state_dict().items() if "running" not in k and "tracked" not in k}
    vocab = ["<blank>", " "] + [chr(97 + i) for i in range(11)]
    brain = Brain(model, head, freeze_wav2vec=True, vocab=vocab)
    g = torch.Generator().manual_seed(3)
    wavs = torch.randn(4, 7000, generator=g) * 0.1
    tokens = torch.randint(1, 13, (4, 5), generator=g)
    wav_lens, tok_lens = torch.tensor([1.0, 0.9, 0.8, 1.0]), torch.tensor([1.0, 0.6, 0.8, 0.4])
    p_enc = model.params.clone()
    # oracle: features once (frozen, evaluation mode), then the optimisation loop on the head
    st = {}
    R.forward(p, oc, S.utt_norm(wavs), stages=st)
    feats = S.utt_norm(st["last_hidden"]).detach().to(torch.bfloat16).float()
    q = {n: t.clone().float().requires_grad_(True) for n, t in sd.items()}
    opt = torch.optim.Adadelta(q.values(), lr=1.0, rho=0.95, eps=1e-8)
    run = [(torch.zeros(64), torch.ones(64)) for _ in range(3)]
    ref_losses, losses = [], []
    for _ in range(4):
        opt.zero_grad()
        l = S.ctc_cost(S.head_forward(q, feats, None, True, dropouts=(0, 0, 0), running=run, bf16_storage=True), tokens, wav_lens,
                       tok_lens)
        l.backward()
        torch.nn.utils.clip_grad_norm_(q.values(), 5.0)
        opt.step()
        ref_losses.append(l.item())
        losses.append(brain.fit_batch(wavs, wav_lens, tokens, tok_lens).item())
    assert np.allclose(losses, ref_losses, rtol=4e-2), (losses, ref_losses)
    assert losses[-1] < losses[0]
    assert torch.equal(model.params, p_enc)  # frozen
    # Adadelta's first steps are sign-like (each about sqrt(eps / (1 - rho)) = 4.5e-4 whatever the gradient's size), so the
    # parameters are compared on the absolute scale of the four updates
    for n in q:
        diff = (head.param(n).cpu()[:q[n].shape[0]] - q[n].detach()).abs()
        assert float(diff.max()) < 6e-3 and float(diff.mean()) < 8e-4, (n, float(diff.max()), float(diff.mean()))  # (an element whose tiny gradient changes sign moves the other way)
        assert float((head.param(n).cpu()[:q[n].shape[0]] - sd[n]).abs().max()) > 1e-4, n  # and did move
    for i in range(3):
        assert np.allclose(head.running_mean[i].cpu(), run[i][0], atol=2e-2)
        assert np.allclose(head.running_var[i].cpu(), run[i][1], rtol=5e-2, atol=1e-3)
    # validation: loss + WER on the device, NewBob on the validation loss
    v1 = brain.evaluate_batch(wavs, wav_lens, tokens, tok_lens).item()
    s1 = brain.on_stage_end(VALID, v1)
    assert s1["lr_model"] == 1.0 and brain.model_optimizer.lr == 1.0 and 0.0 <= s1["WER"]
    s2 = brain.on_stage_end(VALID, v1 * 0.999)  # improvement below 0.25 %: anneal
    assert s2["lr_model"] == 1.0 and brain.model_optimizer.lr == pytest.approx(0.8)
    assert brain.lr_annealing_wav2vec.hyperparam_value == pytest.approx(0.9e-4)
    with torch.no_grad():
        ev = S.ctc_cost(S.head_forward({n: t.detach() for n, t in q.items()}, feats, None, False, dropouts=(0, 0, 0), running=run),
                        tokens, wav_lens, tok_lens).item()
    assert abs(v1 - ev) < 5e-2 * abs(ev)


# ------------------------------------------------------------------ data parallelism of the recipe step
def _sb_dp_worker(rank, world, port, out_dir):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)  # both ranks share the one card of the test box
    import dataclasses
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.sb_head import Brain, CTCHead
    oc = R.W2V2Config.tiny().deterministic()
    d = dataclasses.asdict(oc)
    d.pop("initializer_range")
    model = Wav2Vec2ForCTC(Wav2Vec2Config(**d))
    model.load_state_dict(R.init_params(oc, 5))
    head = CTCHead(oc.hidden_size, 64, 13, dropouts=(0.15, 0.15, 0.0), seed=4)  # same seed: identical replicas at the start
    brain = Brain(model, head, freeze_wav2vec=False, sync_batchnorm=False)  # per-rank statistics, as the reference's DDP
    g = torch.Generator().manual_seed(100 + rank)  # different data per rank
    wavs = torch.randn(3, 7000, generator=g) * 0.1
    tokens = torch.randint(1, 13, (3, 5), generator=g)
    ones = torch.ones(3)
    losses = [brain.fit_batch(wavs, ones, tokens, ones).item() for _ in range(3)]
    torch.save({"enc": model.params[:model.num_trainable].cpu(), "head": head.params.cpu(), "losses": losses,
                "run_mean": head.running_mean[0].cpu()}, os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_recipe_dp2_replicas_stay_identical(tmp_path):
    """Two ranks, different utterances: after three unfrozen fit_batch steps (head gradients in one all-reduce, wav2vec2
    gradients in the engine's bucketed ones, one joint clip coefficient from the reduced buffers) both replicas hold bitwise
    the same parameters, while losses and -- with sync_batchnorm=False, the reference's behaviour -- BatchNorm running statistics
    are per rank."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_sb_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a["enc"], b["enc"]) and torch.equal(a["head"], b["head"])
    assert a["losses"] != b["losses"] and not torch.equal(a["run_mean"], b["run_mean"])
    assert all(np.isfinite(a["losses"])) and all(np.isfinite(b["losses"]))


def _sb_problem():
    import dataclasses
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    oc = R.W2V2Config.tiny().deterministic()
    d = dataclasses.asdict(oc)
    d.pop("initializer_range")
    g = torch.Generator().manual_seed(77)
    wavs = torch.randn(6, 7000, generator=g) * 0.1
    tokens = torch.randint(1, 13, (6, 5), generator=g)  # equal target lengths: shard means average exactly
    return oc, Wav2Vec2Config(**d), R.init_params(oc, 5), wavs, tokens


def _sb_syncbn_worker(rank, world, port, out_dir):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.sb_head import Brain, CTCHead
    oc, cfg, p0, wavs, tokens = _sb_problem()
    model = Wav2Vec2ForCTC(cfg)
    model.load_state_dict(p0)
    head = CTCHead(cfg.hidden_size, 64, 13, dropouts=(0.0, 0.0, 0.0), seed=4)
    brain = Brain(model, head, freeze_wav2vec=False, sync_batchnorm=True)
    assert head.sync_bn
    mine = slice(rank * 3, rank * 3 + 3)
    ones = torch.ones(3)
    loss = brain.fit_batch(wavs[mine], ones, tokens[mine], ones).item()
    torch.save({"head_grads": head.grads.cpu() / world, "enc_grads": model.grads[:model.num_trainable].cpu() / world, "loss": loss,
                "run_mean": head.running_mean[0].cpu(), "run_var": head.running_var[0].cpu(), "head": head.params.cpu()},
               os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_recipe_dp2_sync_batchnorm_equals_single_process(tmp_path):
    """Synchronised BatchNorm: two ranks holding half the batch each, statistics and backward totals exchanged
    as 2 C + 1 doubles per normalisation, reproduce the single-process step on the whole batch -- same running statistics,
    same averaged gradients for the head and for wav2vec2 (up to the summation order of the bf16 GEMMs), same update."""
    import socket
    import torch.multiprocessing as mp
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.sb_head import Brain, CTCHead
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_sb_syncbn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    oc, cfg, p0, wavs, tokens = _sb_problem()
    model = Wav2Vec2ForCTC(cfg)
    model.load_state_dict(p0)
    head = CTCHead(cfg.hidden_size, 64, 13, dropouts=(0.0, 0.0, 0.0), seed=4)
    brain = Brain(model, head, freeze_wav2vec=False)
    ones = torch.ones(6)
    loss = brain.fit_batch(wavs, ones, tokens, ones).item()
    assert abs(0.5 * (a["loss"] + b["loss"]) - loss) < 1e-3 * abs(loss)
    for r in (a, b):
        assert torch.allclose(r["run_mean"], head.running_mean[0].cpu(), atol=1e-5)
        assert torch.allclose(r["run_var"], head.running_var[0].cpu(), rtol=1e-4, atol=1e-6)
        assert rel_l2(r["head_grads"], head.grads.cpu()) < 1e-2
        assert rel_l2(r["enc_grads"], model.grads[:model.num_trainable].cpu()) < 1e-2
    assert torch.equal(a["head"], b["head"])
    # and per-rank statistics (sync_batchnorm=False) do differ from the global ones: the switch is live
    assert not torch.allclose(a["run_mean"], torch.zeros_like(a["run_mean"]))


def test_head_kernels_reject_bad_arguments(hip):
    """Argument errors come back as ValueError through the C ABI's status (never a launch with mismatched shapes)."""
    x = torch.zeros(4, 24, dtype=torch.bfloat16, device="cuda")
    y = torch.empty_like(x)
    f = torch.zeros(24, device="cuda")
    ws = _ws(hip.lib.ssak_batchnorm_workspace_bytes(24))
    args = (hip.ptr(x), hip.ptr(y), 4, 24, hip.ptr(f), hip.ptr(f))
    with pytest.raises(ValueError):  # evaluation mode without running statistics
        hip.check(hip.lib.ssak_batchnorm_act_fwd(*args, None, None, 0.1, 1e-5, 0, 0.01, 0.0, C.c_uint64(0), 0, hip.ptr(f), hip.ptr(f), None,
                                                 hip.ptr(ws), ws.numel(), hip.stream()))
    with pytest.raises(ValueError):  # workspace too small
        hip.check(hip.lib.ssak_batchnorm_act_fwd(*args, None, None, 0.1, 1e-5, 1, 0.01, 0.0, C.c_uint64(0), 0, hip.ptr(f), hip.ptr(f), None,
                                                 hip.ptr(ws), 16, hip.stream()))
    x7 = torch.zeros(4, 28, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError):  # C not a multiple of 8
        hip.check(hip.lib.ssak_batchnorm_act_fwd(hip.ptr(x7), hip.ptr(x7), 4, 28, hip.ptr(f), hip.ptr(f), None, None, 0.1, 1e-5, 1, 0.01, 0.0,
                                                 C.c_uint64(0), 0, hip.ptr(f), hip.ptr(f), None, hip.ptr(ws), ws.numel(), hip.stream()))
    uws = _ws(hip.lib.ssak_utt_norm_workspace_bytes(2))
    w = torch.zeros(2, 1001, device="cuda")
    with pytest.raises(ValueError):  # workspace too small
        hip.check(hip.lib.ssak_utt_norm_fwd(hip.ptr(w), hip.ptr(w), 2, 1001, 0, 1e-5, None, hip.ptr(uws), 8, hip.stream()))
    with pytest.raises(ValueError):
        hip.check(hip.lib.ssak_adadelta_step(None, hip.ptr(f), hip.ptr(f), hip.ptr(f), None, 24, None, 0.0, 1.0, 1.0, 0.95, 1e-8, 0.0,
                                             hip.stream()))
    from ssak_amd.sb_head import CTCHead
    with pytest.raises(ValueError):
        CTCHead(100, 64, 13)  # input_dim must be a multiple of 8
    head = CTCHead(64, 64, 13, dropouts=(0.0, 0.0, 0.0))
    with pytest.raises(RuntimeError):
        head.backward(torch.zeros(1, 2, head.Vp, device="cuda"))  # no training-mode forward
    # a single row / a single utterance still normalise (variance 0 -> rstd = 1 / sqrt(eps))
    one = torch.full((1, 64), 3.0, dtype=torch.bfloat16, device="cuda")
    y1, mean, rstd = _bn_fwd(hip, one, torch.ones(64, device="cuda"), torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda"),
                             torch.ones(64, device="cuda"), True, 0.0, 1, 0)
    assert torch.all(y1 == 0) and torch.allclose(mean, torch.full_like(mean, 3.0)) and torch.allclose(rstd, torch.full_like(rstd, 1e5 ** 0.5), rtol=1e-3)


_HPARAMS = """# written by the test: the recipe's keys with hyperpyyaml tags, small sizes
num_epochs: 6
lr: 1.0
lr_wav2vec: 0.0001
sorting: random
batch_size: 4
test_batch_size: 4
min_duration: 0
max_duration: 15
freeze_wav2vec: True
eval_steps: 4
debug: False
seed: 1234
__set_seed: !apply:torch.manual_seed [!ref <seed>]
train: !PLACEHOLDER
valid: !PLACEHOLDER
output_folder_prefix: ''
base_model: !PLACEHOLDER
dnn_neurons: 64
output_neurons: 30
blank_index: 0
model_opt_class: !name:torch.optim.Adadelta
    lr: !ref <lr>
    rho: 0.95
    eps: 1.e-8
lr_annealing_model: !new:speechbrain.nnet.schedulers.NewBobScheduler
    initial_value: !ref <lr>
    improvement_threshold: 0.0025
    annealing_factor: 0.8
    patient: 0
lr_annealing_wav2vec: !new:speechbrain.nnet.schedulers.NewBobScheduler
    initial_value: !ref <lr_wav2vec>
    improvement_threshold: 0.0025
    annealing_factor: 0.9
    patient: 0
"""


@pytest.mark.timeout(900)
def test_speechbrain_recipe_cli_on_kaldi_folder(tmp_path):
    """`python -m ssak_amd.train_speechbrain HPARAMS.yaml --train=... --valid=... --base_model=...` on a synthetic Kaldi
    corpus: the yaml's hyperpyyaml tags are read, the validation loss goes down over the epochs (frozen wav2vec2, Adadelta on
    the head), train_log.txt has the recipe's fields, the best checkpoint becomes final/, a rerun resumes and has nothing left
    to do, and --freeze_wav2vec=False trains the encoder too."""
    import dataclasses
    import json
    import os
    import subprocess
    import sys
    from ssak_amd import data as D
    from ssak_amd.checkpoint import save_pretrained
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.synth import VOCAB, synth_text, synth_wave
    from oracle import w2v2_ref as R
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(0)
    kd = tmp_path / "kaldi"
    (kd / "audio").mkdir(parents=True)
    with open(kd / "wav.scp", "w") as fw, open(kd / "text", "w") as ft, open(kd / "utt2dur", "w") as fd:
        for i in range(12):
            n = int(rng.integers(16000, 24000))
            D.write_wav(str(kd / "audio" / f"u{i}.wav"), synth_wave(rng, n))
            fw.write(f"utt{i} {kd}/audio/u{i}.wav\n")
            ft.write(f"utt{i} {synth_text(rng, 3, 6)}\n")
            fd.write(f"utt{i} {n / 16000:.3f}\n")
    oc = dataclasses.replace(R.W2V2Config.tiny(), layerdrop=0.0)
    d = dataclasses.asdict(oc)
    d.pop("initializer_range")
    base = Wav2Vec2ForCTC(Wav2Vec2Config(**d))
    base.load_state_dict(R.init_params(oc, 1))
    save_pretrained(base, D.CharTokenizer(VOCAB), str(tmp_path / "base"))
    del base
    hp = tmp_path / "hparams.yaml"
    hp.write_text(_HPARAMS)
    env = dict(os.environ, PYTHONPATH=root)
    cmd = [sys.executable, "-m", "ssak_amd.train_speechbrain", str(hp), f"--train={kd}", f"--valid={kd}", f"--base_model={tmp_path / 'base'}",
           f"--output_folder_prefix={tmp_path}/out_"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    runs = [p for p in os.listdir(tmp_path) if p.startswith("out_sb_")]
    assert len(runs) == 1 and "_frTrue_lr1.0-0.0001_bs4_s1234_random" in runs[0]
    run = tmp_path / runs[0]
    lines = open(run / "train_log.txt").read().strip().split("\n")
    assert len(lines) >= 6  # 3 batches per epoch: one validation at each epoch end, plus every 4th step
    assert all(k in lines[0] for k in ("epoch: 1", "total_samples:", "total_audio_h:", "lr_model: 1", "lr_wav2vec:", "valid loss:", "valid WER:"))
    vloss = [float(l.split("valid loss: ")[1].split(",")[0]) for l in lines]
    assert vloss[-1] < vloss[0]
    assert (run / "final" / "model.ckpt").exists() and (run / "final" / "vocab.json").exists()
    assert 1 <= len([c for c in os.listdir(run / "save") if c.startswith("CKPT-")]) <= 2
    best = json.load(open(run / "final" / "meta.json"))
    assert best["WER"] == min(float(l.split("valid WER: ")[1]) for l in lines) or best["WER"] <= float(lines[-1].split("valid WER: ")[1])
    r2 = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0 and "resuming from" in r2.stdout
    assert open(run / "train_log.txt").read().strip().split("\n") == lines  # nothing left to train
    r3 = subprocess.run(cmd + ["--freeze_wav2vec=False", "--num_epochs=1"], env=env, capture_output=True, text=True, timeout=600)
    assert r3.returncode == 0, r3.stderr[-2000:]
    run3 = [p for p in os.listdir(tmp_path) if p.startswith("out_sb_") and "_frFalse_" in p]
    assert len(run3) == 1 and (tmp_path / run3[0] / "final" / "wav2vec2.ckpt").exists()
