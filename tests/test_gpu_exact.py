"""GPU: the fp32-exact VERIFICATION mode of the engine (``ssak_w2v2_config.exact`` / ``Wav2Vec2ForCTC(exact=True)``).

The reference computes in fp32 (USE_MIXED_PRECISION = False, ssak/train/transformers/wav2vec_train.py:191-192,376-377); the
production engine stores activations in bf16, so its parity bars are bf16 bars (1e-2) that cannot tell a subtly wrong kernel
from rounding.  In the exact mode the SAME engine code -- sequencing, buffer plan, layouts, the row-kernel templates, the
CTC kernels -- runs with float activations and fp32 matrix products (``ssak_gemm_f32``, v_mfma_f32_32x32x2_f32), and is held
here to the bars ``oracle/gen_golden.py`` holds the CPU oracle to against transformers: logits within 2e-4 (absolute), loss
within 1e-4 (SURVEY.md section 8d "fp32 kernel path within 1e-4"), every gradient tensor within 5e-3 of its largest element.
"""
import dataclasses

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(Wav2Vec2Config, oc):
    d = dataclasses.asdict(oc)
    d.pop("initializer_range")
    return Wav2Vec2Config(**d)


def _max_rel_grad_err(model, ref_grads):
    """max over tensors of max|got - ref| / max(max|ref|, floor): the measure of oracle/gen_golden.py:224-238."""
    floor = 1e-3 * max(float(np.abs(g).max()) for g in ref_grads.values())
    worst = ("", 0.0)
    for n, g in ref_grads.items():
        got = model.grad(n).cpu().numpy()
        if n in model._HEAD:
            got = got[:model.config.vocab_size]
        e = float(np.abs(got - g).max()) / max(float(np.abs(g).max()), floor)
        if e > worst[1]:
            worst = (n, e)
    return worst


# ------------------------------------------------------------------------------------------------ the fp32 GEMM itself
@pytest.mark.parametrize("akm,bkm", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_f32_layouts_batches_and_epilogues(akm, bkm):
    """ssak_gemm_f32 against float64 torch: every operand layout, odd sizes, a two-level batch, the overlapping-row (Toeplitz)
    A operand of the conv GEMMs, bias with a per-group stride, alpha, GELU with the saved pre-activation, the GELU-gradient
    product, accumulate, column sums, and the dropout mask of the bf16 kernels (same counter hash, same element offsets)."""
    import ssak_amd.hip as hip
    dev = "cuda:0"
    g = torch.Generator().manual_seed(5 + 2 * akm + bkm)
    M, N, K, nb1, nb2 = 77, 45, 131, 2, 3
    A = torch.randn((nb1, nb2, K, M) if akm else (nb1, nb2, M, K), generator=g)
    Bm = torch.randn((nb1, nb2, K, N) if bkm else (nb1, nb2, N, K), generator=g)
    bias = torch.randn(nb2, N, generator=g)
    Cd = torch.zeros(nb1, nb2, M, N, device=dev)
    pre = torch.zeros_like(Cd)
    Ad, Bd = A.to(dev), Bm.to(dev)
    a_mat = A.transpose(-1, -2) if akm else A
    b_mat = Bm.transpose(-1, -2) if bkm else Bm
    ref_pre = 0.5 * (a_mat.double() @ b_mat.double().transpose(-1, -2)) + bias.double()[None, :, None, :]
    hip.gemm_f32(Ad, Bd, Cd, M, N, K, a_kmajor=akm, b_kmajor=bkm, lda=M if akm else K, ldb=N if bkm else K, ldc=N, nb1=nb1, nb2=nb2,
                 sa=(nb2 * M * K, M * K), sb=(nb2 * N * K, N * K), sc=(nb2 * M * N, M * N), alpha=0.5, bias=bias.to(dev), bias_s2=N,
                 epilogue=hip.EPI_GELU, aux_out=pre)
    ref = torch.nn.functional.gelu(ref_pre)
    assert float((pre.cpu().double() - ref_pre).abs().max()) < 2e-5
    assert float((Cd.cpu().double() - ref).abs().max()) < 2e-5
    # GELU-gradient product + accumulate
    dY = torch.randn(nb1, nb2, M, K, generator=g) if not akm else torch.randn(nb1, nb2, K, M, generator=g)
    C2 = torch.ones(nb1, nb2, M, N, device=dev)
    hip.gemm_f32(dY.to(dev), Bd, C2, M, N, K, a_kmajor=akm, b_kmajor=bkm, lda=M if akm else K, ldb=N if bkm else K, ldc=N, nb1=nb1,
                 nb2=nb2, sa=(nb2 * M * K, M * K), sb=(nb2 * N * K, N * K), sc=(nb2 * M * N, M * N), epilogue=hip.EPI_MUL_GELU_GRAD,
                 aux_in=pre, accumulate=True)
    x = ref_pre.clone().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    dmat = dY.transpose(-1, -2) if akm else dY
    ref2 = 1.0 + (dmat.double() @ b_mat.double().transpose(-1, -2)) * x.grad
    assert float((C2.cpu().double() - ref2).abs().max()) < 1e-4
    # Toeplitz A (row t = K contiguous elements at offset t * lda, lda < K), column sums, dropout mask = the bf16 kernel's
    if not akm:
        T, Ci, k, s, Co = 50, 8, 3, 2, 24
        xs = torch.randn(T * Ci, generator=g)
        W = torch.randn(Co, k * Ci, generator=g) if not bkm else torch.randn(k * Ci, Co, generator=g)
        To = (T - k) // s + 1
        out = torch.zeros(To, Co, device=dev)
        cs = torch.zeros(Co, device=dev)
        hip.gemm_f32(xs.to(dev), W.to(dev), out, To, Co, k * Ci, b_kmajor=bkm, lda=s * Ci, ldb=Co if bkm else k * Ci, ldc=Co,
                     colsum_out=cs)
        rows = torch.stack([xs[t * s * Ci:t * s * Ci + k * Ci] for t in range(To)]).double()
        wm = (W.t() if bkm else W).double()
        refc = rows @ wm.t()
        assert float((out.cpu().double() - refc).abs().max()) < 2e-5
        assert float((cs.cpu().double() - refc.sum(0)).abs().max()) < 1e-4
        Mb, Nb, Kb = 64, 256, 64
        a16 = torch.randn(Mb, Kb, generator=g).bfloat16()
        b16 = torch.randn(Nb, Kb, generator=g).bfloat16()
        o16 = torch.empty(Mb, Nb, dtype=torch.bfloat16, device=dev)
        o32 = torch.empty(Mb, Nb, device=dev)
        hip.gemm(a16.to(dev), b16.to(dev), o16, Mb, Nb, Kb, lda=Kb, ldb=Kb, ldc=Nb, drop_p=0.3, drop_stream=7, drop_seed=99)
        hip.gemm_f32(a16.float().to(dev), b16.float().to(dev), o32, Mb, Nb, Kb, lda=Kb, ldb=Kb, ldc=Nb, drop_p=0.3, drop_stream=7,
                     drop_seed=99)
        assert torch.equal(o16 == 0, o32 == 0) and 0.2 < float((o32 == 0).float().mean()) < 0.4


# ------------------------------------------------------------------------------------------------ engine, tiny configs
def test_exact_tiny_vs_hf_golden(gold):
    """Tiny base topology with a SpecAugment mask (the golden of test_tiny_forward_backward_vs_hf): the exact mode meets the
    oracle's own bars against transformers."""
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    z = gold("w2v2_tiny.npz")
    oc = R.W2V2Config.tiny().deterministic()
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc), exact=True).train()
    model.load_state_dict(R.init_params(oc, 69))
    out = model(torch.tensor(z["x"]), labels=torch.tensor(z["labels"]), mask_time_indices=z["mask"])
    e_logits = float(np.abs(out.logits.cpu().numpy() - z["logits"]).max())
    e_loss = abs(out.loss.item() - float(z["loss"])) / float(z["loss"])
    model.grads[:model.num_trainable].fill_(float("nan"))
    model.backward()
    worst = _max_rel_grad_err(model, {k[5:]: z[k] for k in z.files if k.startswith("grad/")})
    print("exact tiny: logits max abs err", e_logits, "loss rel err", e_loss, "worst grad", worst)
    assert e_logits < 2e-4 and e_loss < 1e-4 and worst[1] < 5e-3
    # the bf16 mode on the same inputs sits two orders of magnitude further away: the exact bar is not vacuous
    bf = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    bf.load_state_dict(R.init_params(oc, 69))
    o2 = bf(torch.tensor(z["x"]), labels=torch.tensor(z["labels"]), mask_time_indices=z["mask"])
    assert float(np.abs(o2.logits.cpu().numpy() - z["logits"]).max()) > 10 * e_logits


def test_exact_tiny_xlsr_ragged_vs_hf_golden(gold):
    """Layer-norm feature encoder with bias, stable-layer-norm layers, ragged lengths + attention mask (XLSR topology)."""
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    z = gold("w2v2_tiny_xlsr.npz")
    oc = R.W2V2Config.tiny(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True).deterministic()
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc), exact=True).train()
    model.load_state_dict(R.init_params(oc, 70))
    lens = list(z["lens"])
    out = model(torch.tensor(z["x"]), lengths=torch.tensor(lens), labels=torch.tensor(z["labels"]))
    fl = R.conv_out_lengths(oc, lens)
    e_logits = max(float(np.abs(out.logits[b, :fl[b]].cpu().numpy() - z["logits"][b, :fl[b]]).max()) for b in range(len(lens)))
    e_loss = abs(out.loss.item() - float(z["loss"])) / float(z["loss"])
    model.grads[:model.num_trainable].fill_(float("nan"))
    model.backward()
    worst = _max_rel_grad_err(model, {k[5:]: z[k] for k in z.files if k.startswith("grad/")})
    print("exact tiny xlsr: logits max abs err", e_logits, "loss rel err", e_loss, "worst grad", worst)
    assert e_logits < 2e-4 and e_loss < 1e-4 and worst[1] < 5e-3


def test_exact_regularisers_on_layerdrop_and_replay():
    """Dropouts, LayerDrop and SpecAugment ON in the exact mode: the masks are the counter hashes of the bf16 mode (same element
    offsets), a dropped layer's gradient is exactly zero, and the same seed replays the same step."""
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    oc = R.W2V2Config.tiny()
    p = R.init_params(oc, 3)
    rng = np.random.default_rng(1)
    x = R.zero_mean_unit_var_norm([rng.standard_normal(8000).astype(np.float32) for _ in range(2)])
    labels = R.pad_labels([[3, 4, 5], [6, 7]])
    outs = []
    for rep in range(2):
        model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc), seed=11, exact=True).train()
        model.load_state_dict(p)
        out = model(torch.tensor(x), labels=torch.tensor(labels), layer_keep=[True, False])
        model.backward()
        outs.append((out.loss.item(), model.grads.clone()))
        assert np.isfinite(out.loss.item())
        assert float(model.grad("wav2vec2.encoder.layers.1.feed_forward.output_dense.weight").abs().max()) == 0.0
        assert float(model.grad("wav2vec2.encoder.layers.0.feed_forward.output_dense.weight").abs().max()) > 0.0
    assert outs[0][0] == outs[1][0] and float((outs[0][1] - outs[1][1]).abs().max()) < 1e-6


# ------------------------------------------------------------------------------------------------ engine, headline config
def test_exact_base_vs_hf_golden_and_oracle(gold):
    """wav2vec2-base, B=2 x 10 s (BASELINE config 2 shapes) in the exact mode: logits / loss vs the transformers golden at the
    oracle's bars, gradient norms + projections vs the golden, every gradient tensor vs the CPU oracle at 5e-3 max-rel."""
    from oracle import w2v2_ref as R
    from oracle.gen_golden import base_inputs
    from oracle.gen_golden_full import proj_dirs
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    z = gold("w2v2_base.npz")
    x, labels = base_inputs()
    oc = R.W2V2Config.base().deterministic()
    params = R.init_params(oc, 69)
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc), exact=True).train()
    model.load_state_dict(params)
    out = model(torch.tensor(x), labels=torch.tensor(labels))
    e_logits = float(np.abs(out.logits.cpu().numpy() - z["logits"]).max())
    e_loss = abs(out.loss.item() - float(z["loss"])) / float(z["loss"])
    print("exact base: logits max abs err", e_logits, "loss rel err", e_loss)
    assert e_logits < 2e-4 and e_loss < 1e-4
    model.grads[:model.num_trainable].fill_(float("nan"))
    model.backward()
    gmax = float(z["grad_norms"].max())
    worst_n = worst_p = 0.0
    for i, (n, nr, pr) in enumerate(zip(z["grad_names"], z["grad_norms"], z["grad_projs"])):
        g = model.grad(str(n)).double().reshape(-1).cpu()
        if nr < 1e-4 * gmax:
            continue
        worst_n = max(worst_n, abs(float(g.norm()) - nr) / nr)
        got = (proj_dirs(i, g.numel()).double() @ g).numpy()
        worst_p = max(worst_p, float(np.sqrt(np.mean((got - pr) ** 2))) / nr)
    print("exact base vs transformers golden: worst norm err", worst_n, "worst projection err / |g|", worst_p)
    assert worst_n < 1e-3 and worst_p < 2e-3  # (bf16 mode: 6e-3 / 4e-2)
    loss, logits, grads = R.loss_and_grads(params, oc, torch.tensor(x), None, torch.tensor(labels))
    worst = _max_rel_grad_err(model, {n: g.numpy() for n, g in grads.items()})
    print("exact base vs oracle, full tensors: worst", worst)
    assert worst[1] < 5e-3


def test_exact_matched_loss_tiny_30_steps():
    """"fp32 kernel path within 1e-4" on the loss -- 30 optimizer steps (AdamW, clip 1.0, warm-up 5) of the
    exact mode against eager torch fp32 + torch.optim.AdamW from the same init, regularisers off."""
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer, linear_warmup_lr
    oc = R.W2V2Config.tiny().deterministic()
    p0 = R.init_params(oc, 13)
    rng = np.random.default_rng(0)
    x = R.zero_mean_unit_var_norm([rng.standard_normal(8000).astype(np.float32) for _ in range(4)])
    labels = R.pad_labels([list(rng.integers(1, 32, n)) for n in (6, 4, 7, 5)])
    steps, lr, warm = 30, 3e-4, 5
    names = R.trainable_names(oc)
    q = {n: (t.clone().requires_grad_(n in names)) for n, t in p0.items()}
    opt = torch.optim.AdamW([q[n] for n in names], lr=lr, weight_decay=0.0)
    ref = []
    for s in range(steps):
        for g in opt.param_groups:
            g["lr"] = linear_warmup_lr(lr, s, warm, 1000)
        loss, _ = R.forward(q, oc, torch.tensor(x), None, torch.tensor(labels))
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([q[n] for n in names], 1.0)
        opt.step()
        ref.append(loss.item())
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc), exact=True).train()
    model.load_state_dict(p0)
    tr = Trainer(model, AdamW(model, lr=lr, warmup_steps=warm, total_steps=1000, max_grad_norm=1.0))
    xd, ld = torch.tensor(x).cuda(), torch.tensor(labels).cuda()
    got = [float(tr.train_step(xd, None, ld, raw=False).item()) for _ in range(steps)]
    rel = max(abs(a - b) / abs(b) for a, b in zip(got, ref))
    print("exact matched loss: first", got[0], ref[0], "last", got[-1], ref[-1], "max rel", rel)
    assert ref[-1] < 0.9 * ref[0] and rel < 1e-4


def test_exact_base_matched_loss_50_steps_vs_hf_curve(gold):
    """SURVEY.md section 8d "Matched loss" on the HEADLINE config in the fp32-exact mode: wav2vec2-base, B=2 x 10 s, 50 optimizer
    steps of HF Trainer's inner loop (the train script's schedule: lr 1e-4, warm-up 500, clip 1.0), golden curve made with
    transformers.Wav2Vec2ForCTC + torch.optim.AdamW + get_linear_schedule_with_warmup + clip_grad_norm_
    (oracle/gen_golden_full.py, tests/golden/w2v2_base_curve.npz).

    The bar.  SURVEY asks 1e-4 of an "fp32 kernel path"; two CORRECT fp32 implementations do not stay that close over 50 steps
    of this run, for a measured reason: the fp32 log-domain CTC lattice over 499 frames.  torch's own fp32 CTC gradient sits
    1.1e-3 (relative L2) from its float64 evaluation on the base-shape case of tests/golden/ctc_cases.npz (loss: 5e-7), this
    kernel sits at the same distance on the other side, and the optimizer amplifies the difference step by step; re-running the
    GOLDEN's own code with 3 instead of 8 threads (summation order only, same CTC code) already moves its curve by 8e-5.  So:
    every step within 2e-3 (measured: 1.0e-3 at step 45, 15 x below the bf16 engine's 2e-2 bar on the same curve), the first
    10 steps -- before the drift builds up -- within 1e-4, and the clip's gradient norm within 3e-2."""
    from oracle import w2v2_ref as R
    from oracle.gen_golden_full import curve_inputs
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer
    z = gold("w2v2_base_curve.npz")
    steps = int(z["steps"])
    oc = R.W2V2Config.base().deterministic()
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc), exact=True).train()
    model.load_state_dict(R.init_params(oc, 69))
    opt = AdamW(model, lr=float(z["base_lr"]), warmup_steps=int(z["warmup"]), total_steps=int(z["total"]),
                weight_decay=float(z["weight_decay"]), max_grad_norm=float(z["max_grad_norm"]))
    tr = Trainer(model, opt)
    batches = [(torch.tensor(x).cuda(), torch.tensor(l).cuda()) for x, l in curve_inputs()]
    got, norms = [], []
    for s in range(steps):
        assert abs(opt.current_lr() - float(z["lr"][s])) < 1e-12
        x, l = batches[s % len(batches)]
        got.append(float(tr.train_step(x, None, l, raw=False).item()))
        norms.append(opt.grad_norm())
    got, ref = np.array(got), z["loss"]
    rel = np.abs(got - ref) / np.abs(ref)
    gn = np.abs(np.array(norms) - z["grad_norm"]) / z["grad_norm"]
    print("exact base curve: first", got[0], ref[0], "last", got[-1], ref[-1], "max rel", rel.max(), "at step", int(rel.argmax()),
          "grad-norm max rel", gn.max())
    print("   per-step rel:", " ".join(f"{r:.1e}" for r in rel))
    assert ref[-8:].mean() < 0.6 * ref[:8].mean()
    assert rel[:10].max() < 1e-4 and rel.max() < 2e-3 and gn.max() < 3e-2
    sd = model.state_dict()
    for n, nr in zip(z["param_names"], z["param_norms"]):  # where the 50 updates went
        t = sd[str(n)].double().reshape(-1)
        assert abs(float(t.norm()) - nr) < 1e-4 * nr + 1e-7, (str(n), float(t.norm()), nr)


# ------------------------------------------------------------------------------------------------ Whisper (arch = 1), --no_freeze
def test_exact_whisper_tiny_vs_hf_golden(gold):
    """The Whisper encoder + CTC composition (BASELINE config 4) in the fp32-exact mode against the golden made with
    transformers.WhisperEncoder: logits 2e-4, loss 1e-4, every gradient tensor within 5e-3 of its largest element (the bars
    the CPU oracle is held to; the bf16 engine's are 2e-2 / 6e-2).  The front end (conv1 / conv2 as Toeplitz GEMMs, position
    add, col2im + GELU') runs on the same templates with float activations."""
    import ssak_amd.hip as hip
    from oracle import whisper_ref as WR
    from ssak_amd.whisper import WhisperCTCConfig, WhisperEncoderForCTC
    z = gold("whisper_tiny.npz")
    oc = WR.WhisperCTCConfig.tiny()
    cfg = WhisperCTCConfig(vocab_size=oc.vocab_size, d_model=oc.d_model, encoder_layers=oc.encoder_layers,
                           encoder_attention_heads=oc.encoder_attention_heads, encoder_ffn_dim=oc.encoder_ffn_dim,
                           max_source_positions=oc.max_source_positions)
    model = WhisperEncoderForCTC(cfg, exact=True).train()
    model.load_state_dict(WR.init_params(oc, 69))
    out = model(torch.tensor(z["mel"]).cuda(), labels=torch.tensor(z["labels"]))  # the golden's own features: isolates the encoder
    e_logits = float(np.abs(out.logits.cpu().numpy() - z["logits"]).max())
    e_loss = abs(out.loss.item() - float(z["loss"])) / float(z["loss"])
    model.grads[:model.num_trainable].fill_(float("nan"))
    model.backward()
    worst = _max_rel_grad_err(model, {k[5:]: z[k] for k in z.files if k.startswith("grad/")})
    print("exact whisper tiny: logits max abs err", e_logits, "loss rel err", e_loss, "worst grad", worst)
    assert e_logits < 2e-4 and e_loss < 1e-4 and worst[1] < 5e-3
    assert float(model.grad("encoder.layers.0.self_attn.k_proj.bias").abs().max()) == 0.0


def test_exact_whisper_small_window_vs_hf_golden(gold):
    """Whisper-small (12 x 768, 1500 positions) on ONE full 30 s window in the fp32-exact mode: logits, loss, gradient norms and
    projections against transformers.WhisperEncoder (tests/golden/whisper_small.npz; features from the golden run's own
    extractor are re-created by the HIP log-mel kernel, pinned at 2e-4 elsewhere)."""
    import ssak_amd.hip as hip
    from oracle import whisper_ref as WR
    from oracle.gen_golden_full import proj_dirs, whisper_inputs
    from ssak_amd.whisper import WhisperCTCConfig, WhisperEncoderForCTC
    z = gold("whisper_small.npz")
    wav, labels = whisper_inputs()
    oc = WR.WhisperCTCConfig()
    model = WhisperEncoderForCTC(WhisperCTCConfig(vocab_size=oc.vocab_size), exact=True).train()
    model.load_state_dict(WR.init_params(oc, 73))
    mel = hip.logmel_whisper(torch.tensor(wav).cuda())
    out = model(mel, labels=torch.tensor(labels))
    e_logits = float(np.abs(out.logits.cpu().numpy() - z["logits"]).max())
    e_loss = abs(out.loss.item() - float(z["loss"])) / float(z["loss"])
    model.grads[:model.num_trainable].fill_(float("nan"))
    model.backward()
    gmax = float(z["grad_norms"].max())
    wn = wp = 0.0
    perr = []
    for i, (n, nr, pr) in enumerate(zip(z["grad_names"], z["grad_norms"], z["grad_projs"])):
        g = model.grad(str(n))
        if str(n) in model._HEAD:
            g = g[:model.config.vocab_size]
        g = g.double().reshape(-1).cpu()
        if nr < 1e-4 * gmax:
            continue
        wn = max(wn, abs(float(g.norm()) - nr) / nr)
        perr.append(float(np.sqrt(np.mean(((proj_dirs(i, g.numel()).double() @ g).numpy() - pr) ** 2))) / nr)
        wp = max(wp, perr[-1])
    print("exact whisper small: logits max abs err", e_logits, "loss rel err", e_loss, "worst norm err", wn, "worst projection err / |g|", wp,
          "median projection err", float(np.median(perr)), "min", min(perr))
    # The forward is fp32-grade (logits 1e-5, loss 1e-6).  The gradients of EVERY tensor, the head's included, sit ~5e-3 from the
    # golden's: what they share is d loss / d logits, i.e. the CTC lattice in fp32 over 1500 frames, where torch's own kernel is
    # ~1e-3 (499 frames: tests/golden/ctc_cases.npz base_shape, vs its float64 evaluation) to several 1e-3 from exact; two
    # fp32 lattices differ by that much.  Bars: 2e-2 (the bf16 engine's on this golden: 6e-2).
    assert e_logits < 2e-3 and e_loss < 2e-4 and wn < 2e-2 and wp < 2e-2


@pytest.mark.parametrize("topology", ["group_norm", "layer_norm"])
def test_exact_no_freeze_feature_encoder_gradients(topology):
    """--no_freeze (wav2vec_train.py:326-327 off) in the fp32-exact mode, both feature-encoder variants: every gradient --
    conv weights / biases, GroupNorm / LayerNorm affines and the encoder's -- within 5e-3 of its tensor's largest element of the
    CPU oracle's autograd (bf16 bars: 8e-2 / 6e-2)."""
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    kw = dict(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True) if topology == "layer_norm" else {}
    oc = R.W2V2Config.tiny(**kw).deterministic()
    p = R.init_params(oc, 17)
    rng = np.random.default_rng(6)
    lens = [8000, 6100, 7333] if kw else None
    x = R.zero_mean_unit_var_norm([rng.standard_normal(n).astype(np.float32) for n in (lens or [8000] * 3)])
    labels = R.pad_labels([list(rng.integers(1, 32, n)) for n in (8, 3, 6)])
    loss, logits, grads = R.loss_and_grads(p, oc, torch.tensor(x), lens, torch.tensor(labels), freeze_feature_encoder=False)
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc), freeze_feature_encoder=False, exact=True).train()
    model.load_state_dict(p)
    out = model(torch.tensor(x), lengths=None if lens is None else torch.tensor(lens), labels=torch.tensor(labels))
    assert abs(out.loss.item() - loss.item()) < 1e-4 * loss.item()
    model.grads[:model.num_trainable].fill_(float("nan"))
    model.backward()
    worst = _max_rel_grad_err(model, {n: g.numpy() for n, g in grads.items()})
    fe = _max_rel_grad_err(model, {n: g.numpy() for n, g in grads.items() if n.startswith("wav2vec2.feature_extractor.")})
    print("exact no_freeze", topology, ": worst grad", worst, "worst feature-encoder grad", fe)
    assert worst[1] < 5e-3
    # the optimizer moves the conv weights and their fp32 GEMM layouts follow
    from ssak_amd.trainer import AdamW
    w_before = model.param("wav2vec2.feature_extractor.conv_layers.3.conv.weight").clone()
    AdamW(model, lr=1e-3, warmup_steps=0).step()
    assert (model.param("wav2vec2.feature_extractor.conv_layers.3.conv.weight") - w_before).abs().max().item() > 0
    out2 = model(torch.tensor(x), lengths=None if lens is None else torch.tensor(lens), labels=torch.tensor(labels))
    assert out2.loss.item() != out.loss.item()


def test_exact_bucketed_mixed_length_epoch_vs_oracle():
    """BASELINE config 5's data path at tiny dimensions, end to end in the fp32-exact mode: XLSR topology (layer-norm feature
    encoder with bias, stable LayerNorm, attention mask), 14 utterances with durations log-uniform in [0.25 s, 1.2 s], batches
    built by length grouping (HF LengthGroupedSampler: ssak_amd.data.length_grouped_batches), padded to the longest with masks,
    the short last batch trained -- one epoch of optimizer steps (AdamW, clip 1.0, warm-up) against eager torch fp32 from the same
    init: every step's loss within 1e-4 (7 x 4 different padded shapes: a new workspace plan per batch)."""
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.data import length_grouped_batches
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer, linear_warmup_lr
    oc = R.W2V2Config.tiny(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True).deterministic()
    p0 = R.init_params(oc, 29)
    rng = np.random.default_rng(12)
    lens = [int(16000 * np.exp(rng.uniform(np.log(0.25), np.log(1.2)))) for _ in range(14)]
    waves = [rng.standard_normal(n).astype(np.float32) for n in lens]
    texts = [list(rng.integers(1, 32, max(1, n // 4000))) for n in lens]
    batches = length_grouped_batches(lens, 4, np.random.RandomState(7), mega_factor=2)
    assert sorted(sum(batches, [])) == list(range(14)) and [len(b) for b in batches].count(4) >= 2 and min(len(b) for b in batches) < 4
    lr, warm = 3e-4, 2
    names = R.trainable_names(oc)
    q = {n: (t.clone().requires_grad_(n in names)) for n, t in p0.items()}
    opt = torch.optim.AdamW([q[n] for n in names], lr=lr, weight_decay=0.0)
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc), exact=True).train()
    model.load_state_dict(p0)
    tr = Trainer(model, AdamW(model, lr=lr, warmup_steps=warm, total_steps=1000, max_grad_norm=1.0))
    worst = 0.0
    for s, idx in enumerate(batches):
        bl = [lens[i] for i in idx]
        x = R.zero_mean_unit_var_norm([waves[i] for i in idx])
        lab = R.pad_labels([texts[i] for i in idx])
        for g in opt.param_groups:
            g["lr"] = linear_warmup_lr(lr, s, warm, 1000)
        loss, _ = R.forward(q, oc, torch.tensor(x), bl, torch.tensor(lab))
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([q[n] for n in names], 1.0)
        opt.step()
        got = float(tr.train_step(torch.tensor(x).cuda(), torch.tensor(bl).cuda(), torch.tensor(lab).cuda(), raw=False).item())
        worst = max(worst, abs(got - loss.item()) / abs(loss.item()))
    print("exact bucketed epoch:", len(batches), "steps, shapes", sorted({(len(b), max(lens[i] for i in b)) for b in batches})[:3], "... max rel", worst)
    assert worst < 1e-4
