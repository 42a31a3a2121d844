"""GPU parity of the HIP acoustic-model engine (through the C ABI) against the golden vectors made by
transformers.Wav2Vec2ForCTC and against the CPU oracle.

Tolerances: the engine computes in bf16 (fp32 accumulate, fp32 LayerNorm/softmax/CTC statistics) while the
reference is fp32 (bf16 is a build-side choice).  bf16 has 8 significant bits, so tensors
are compared by relative L2 error: logits <= 2e-2, loss <= 2e-2 (BASELINE.md "matched loss"), gradients <= 6e-2.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-12))


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available()
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from oracle import w2v2_ref as R
    return Wav2Vec2Config, Wav2Vec2ForCTC, R


def _cfg_from_oracle(Wav2Vec2Config, oc):
    import dataclasses
    d = dataclasses.asdict(oc)
    d.pop("initializer_range")
    return Wav2Vec2Config(**d)


def _check_grads(model, ref_grads, tol, floor_frac=2e-2):
    gmax = max(float(np.abs(g).max()) for g in ref_grads.values())
    worst = ("", 0.0)
    for n, g in ref_grads.items():
        got = model.grad(n).cpu().numpy()
        if np.abs(g).max() < floor_frac * 1e-2 * gmax:  # numerically-zero gradients (k_proj.bias): absolute check
            assert np.abs(got - g).max() < 1e-3 * gmax, n
            continue
        e = rel_l2(got, g)
        if e > worst[1]:
            worst = (n, e)
        assert e < tol, (n, e)
    return worst


def test_tiny_forward_backward_vs_hf(mods, gold):
    Wav2Vec2Config, Wav2Vec2ForCTC, R = mods
    z = gold("w2v2_tiny.npz")
    oc = R.W2V2Config.tiny().deterministic()
    model = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, oc))
    model.load_state_dict(R.init_params(oc, 69))
    model.train()
    out = model(torch.tensor(z["x"]), labels=torch.tensor(z["labels"]), mask_time_indices=z["mask"])
    assert rel_l2(out.logits.cpu().numpy(), z["logits"]) < 2e-2
    assert abs(out.loss.item() - float(z["loss"])) < 2e-2 * float(z["loss"])
    model.grads[:model.num_trainable].fill_(float("nan"))  # an earlier step's values must be overwritten or zeroed, never kept
    model.backward()
    grads = {k[5:]: z[k] for k in z.files if k.startswith("grad/")}
    worst = _check_grads(model, grads, 6e-2)
    print("tiny worst grad", worst)
    # eval-mode forward (inference path, shared layer buffers) gives the same logits
    model.eval()
    out2 = model(torch.tensor(z["x"]), mask_time_indices=z["mask"])
    assert torch.equal(out2.logits, out.logits)


@pytest.mark.parametrize("exact", [False, True])
def test_raw_input_folds_the_normalisation_into_the_feature_encoder(mods, exact):
    """SSAK_W2V2_OPT_RAW_INPUT: the group-norm model fed RAW full-length waveforms gives the logits, loss and gradients of the
    same model fed ssak_wave_normalize(waveforms) (the normalisation rides in conv0's GroupNorm statistics), in the bf16 engine
    and in the fp32-exact mode; the Trainer takes that path by itself for raw full-length batches and lands on the same update
    as with the separate pass (SSAK_FOLD_NORM=0); the layer-norm topology and an unfrozen feature encoder refuse the option."""
    import os
    from ssak_amd import hip
    from ssak_amd.trainer import AdamW, Trainer
    Wav2Vec2Config, Wav2Vec2ForCTC, R = mods
    oc = R.W2V2Config.tiny().deterministic()
    rng = np.random.default_rng(7)
    raw = torch.tensor((rng.standard_normal((3, 6007)) * 0.07 + 0.01).astype(np.float32)).cuda()
    labels = torch.tensor(R.pad_labels([list(rng.integers(1, 32, 4)) for _ in range(3)]))
    xn = hip.wave_normalize(raw, None)

    def make():
        m = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, oc), exact=exact).train()
        m.load_state_dict(R.init_params(oc, 5))
        return m
    a, b = make(), make()
    oa = a(xn, labels=labels)
    ob = b(raw, labels=labels, raw_input=True)
    tol = 2e-5 if exact else 1e-2
    assert rel_l2(ob.logits.cpu().numpy(), oa.logits.cpu().numpy()) < tol
    assert abs(ob.loss.item() - oa.loss.item()) < tol * abs(oa.loss.item())
    a.backward()
    b.backward()
    ga, gb = a.grads[:a.num_trainable].cpu(), b.grads[:b.num_trainable].cpu()
    assert float((gb - ga).norm() / ga.norm()) < (1e-4 if exact else 3e-2)
    # the same handle goes back to normalised input (evaluation does)
    oc2 = b(xn, labels=labels)
    assert rel_l2(oc2.logits.cpu().numpy(), oa.logits.cpu().numpy()) < (1e-6 if exact else 1e-2)
    # the trainer folds by itself
    ta, tb = make(), make()
    tr_b = Trainer(tb, AdamW(tb, lr=1e-3, warmup_steps=2, total_steps=100))
    os.environ["SSAK_FOLD_NORM"] = "0"
    try:
        tr_a = Trainer(ta, AdamW(ta, lr=1e-3, warmup_steps=2, total_steps=100))
    finally:
        del os.environ["SSAK_FOLD_NORM"]
    assert tr_b._fold_norm and not tr_a._fold_norm
    for _ in range(3):
        la, lb = tr_a.train_step(raw, None, labels.cuda()), tr_b.train_step(raw, None, labels.cuda())
    assert abs(la.item() - lb.item()) < (1e-4 if exact else 2e-2) * abs(la.item())
    # topologies without the fold
    xl = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, R.W2V2Config.tiny(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True).deterministic()))
    assert not xl.can_fold_normalisation()
    with pytest.raises(ValueError):
        xl(raw, raw_input=True)
    nf = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, oc), freeze_feature_encoder=False)
    assert not nf.can_fold_normalisation()


def test_tiny_ragged_lengths_vs_oracle(mods):
    """Attention mask path (lengths): padded frames zeroed, padded keys masked, CTC input lengths shortened."""
    Wav2Vec2Config, Wav2Vec2ForCTC, R = mods
    oc = R.W2V2Config.tiny().deterministic()
    p = R.init_params(oc, 5)
    rng = np.random.default_rng(0)
    lens = [8000, 5000, 6500, 3000]
    x = R.zero_mean_unit_var_norm([rng.standard_normal(n).astype(np.float32) for n in lens])
    labels = R.pad_labels([list(rng.integers(1, 32, n)) for n in (7, 3, 5, 2)])
    loss, logits, grads = R.loss_and_grads(p, oc, torch.tensor(x), lens, torch.tensor(labels))
    model = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, oc)).train()
    model.load_state_dict(p)
    out = model(torch.tensor(x), lengths=torch.tensor(lens), labels=torch.tensor(labels))
    fl = R.conv_out_lengths(oc, lens)
    assert (out.frame_lens.cpu().numpy() == fl).all()
    for b in range(4):  # only valid frames are defined by the reference
        assert rel_l2(out.logits[b, :fl[b]].cpu().numpy(), logits[b, :fl[b]].numpy()) < 2e-2
    assert abs(out.loss.item() - loss.item()) < 2e-2 * loss.item()
    model.grads[:model.num_trainable].fill_(float("nan"))  # an earlier step's values must be overwritten or zeroed, never kept
    model.backward()
    _check_grads(model, {n: g.numpy() for n, g in grads.items()}, 6e-2)


def test_tiny_xlsr_topology_vs_hf(mods, gold):
    """Layer-norm feature encoder with conv bias + stable-layer-norm (pre-LN) layers + attention mask on ragged
    lengths (the XLSR-53 shape of BASELINE config 5), against the transformers golden."""
    Wav2Vec2Config, Wav2Vec2ForCTC, R = mods
    z = gold("w2v2_tiny_xlsr.npz")
    oc = R.W2V2Config.tiny(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True).deterministic()
    model = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, oc)).train()
    model.load_state_dict(R.init_params(oc, 70))
    lens = list(z["lens"])
    out = model(torch.tensor(z["x"]), lengths=torch.tensor(lens), labels=torch.tensor(z["labels"]))
    fl = R.conv_out_lengths(oc, lens)
    for b in range(len(lens)):
        assert rel_l2(out.logits[b, :fl[b]].cpu().numpy(), z["logits"][b, :fl[b]]) < 2e-2
    assert abs(out.loss.item() - float(z["loss"])) < 2e-2 * float(z["loss"])
    model.grads[:model.num_trainable].fill_(float("nan"))  # an earlier step's values must be overwritten or zeroed, never kept
    model.backward()
    worst = _check_grads(model, {k[5:]: z[k] for k in z.files if k.startswith("grad/")}, 6e-2)
    print("tiny xlsr worst grad", worst)


def test_xlsr_layerdrop_and_dropout_vs_oracle(mods):
    """stable-LN with a dropped layer (explicit layer_keep) and SpecAugment mask, dropouts off: against the oracle."""
    Wav2Vec2Config, Wav2Vec2ForCTC, R = mods
    import dataclasses
    oc = R.W2V2Config.tiny(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True, num_hidden_layers=3).deterministic()
    p = R.init_params(oc, 9)
    rng = np.random.default_rng(2)
    x = R.zero_mean_unit_var_norm([rng.standard_normal(8000).astype(np.float32) for _ in range(2)])
    labels = R.pad_labels([[3, 4, 5, 6], [7, 8]])
    mask = R.compute_mask_indices((2, 24), 0.2, 3, None, 2, rng=np.random.RandomState(3))
    keep = [True, False, True]
    loss, logits, grads = R.loss_and_grads(p, oc, torch.tensor(x), None, torch.tensor(labels),
                                           mask_time_indices=torch.tensor(mask), layer_keep=keep)
    model = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, oc)).train()
    model.load_state_dict(p)
    out = model(torch.tensor(x), labels=torch.tensor(labels), mask_time_indices=mask, layer_keep=keep)
    assert rel_l2(out.logits.cpu().numpy(), logits.numpy()) < 2e-2
    model.grads[:model.num_trainable].fill_(float("nan"))  # an earlier step's values must be overwritten or zeroed, never kept
    model.backward()
    g = {n: v.numpy() for n, v in grads.items()}
    _check_grads(model, g, 6e-2)
    assert float(model.grad("wav2vec2.encoder.layers.1.attention.out_proj.weight").abs().max()) == 0.0
    assert float(model.grad("wav2vec2.encoder.layers.2.layer_norm.weight").abs().max()) > 0.0
    # dropouts on: finite and replayable
    oc2 = dataclasses.replace(oc, attention_dropout=0.1, hidden_dropout=0.1, activation_dropout=0.1, final_dropout=0.1)
    losses = []
    for rep in range(2):
        m2 = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, oc2), seed=5).train()
        m2.load_state_dict(p)
        o2 = m2(torch.tensor(x), labels=torch.tensor(labels), layer_keep=keep)
        m2.backward()
        losses.append((o2.loss.item(), m2.grads.clone()))
    assert np.isfinite(losses[0][0]) and losses[0][0] == losses[1][0]
    assert (losses[0][1] - losses[1][1]).abs().max().item() < 1e-6


# (wav2vec2-base at full size -- logits, loss, gradient norms, random projections and every gradient tensor in full -- is
# tests/test_gpu_fullsize.py::test_base_gradients_projections_and_full_tensors)


def test_dropout_replay_and_layerdrop(mods):
    """Stochastic regularisers on: same seed -> identical step (masks are counter-based and replayed in the
    backward); gradient of a dropped layer is exactly zero; loss stays finite."""
    Wav2Vec2Config, Wav2Vec2ForCTC, R = mods
    oc = R.W2V2Config.tiny()
    p = R.init_params(oc, 3)
    rng = np.random.default_rng(1)
    x = R.zero_mean_unit_var_norm([rng.standard_normal(8000).astype(np.float32) for _ in range(2)])
    labels = R.pad_labels([[3, 4, 5], [6, 7]])
    outs = []
    for rep in range(2):
        model = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, oc), seed=11).train()
        model.load_state_dict(p)
        out = model(torch.tensor(x), labels=torch.tensor(labels), layer_keep=[True, False])
        model.grads[:model.num_trainable].fill_(float("nan"))  # an earlier step's values must be overwritten or zeroed, never kept
        model.backward()
        outs.append((out.loss.item(), model.grads.clone()))
        assert np.isfinite(out.loss.item())
        assert float(model.grad("wav2vec2.encoder.layers.1.feed_forward.output_dense.weight").abs().max()) == 0.0
        assert float(model.grad("wav2vec2.encoder.layers.0.feed_forward.output_dense.weight").abs().max()) > 0.0
    assert outs[0][0] == outs[1][0]
    assert (outs[0][1] - outs[1][1]).abs().max().item() < 1e-6


def test_trainer_bucketed_allreduce_single_rank(mods):
    """The data-parallel code path on one GPU (world_size 1 over RCCL): gradient ranges announced by the engine are
    disjoint, cover [0, num_trainable) and the bucketed step equals the plain step."""
    import os
    import torch.distributed as dist
    Wav2Vec2Config, Wav2Vec2ForCTC, R = mods
    from ssak_amd.trainer import AdamW, Trainer
    oc = R.W2V2Config.tiny().deterministic()
    p = R.init_params(oc, 3)
    rng = np.random.default_rng(1)
    x = torch.tensor(R.zero_mean_unit_var_norm([rng.standard_normal(8000).astype(np.float32) for _ in range(2)])).cuda()
    labels = torch.tensor(R.pad_labels([[3, 4, 5], [6, 7]])).cuda()

    def run(distributed, exchange="torch"):
        model = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, oc), seed=11).train()
        model.load_state_dict(p)
        seen = []
        if distributed:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        try:
            tr = Trainer(model, AdamW(model, warmup_steps=0, lr=1e-3), exchange=exchange)
            if distributed:
                orig = tr._on_grads_ready
                model.set_grad_ready_callback(lambda o, c: (seen.append((o, c)), orig(o, c)))
            for _ in range(2):
                loss = tr.train_step(x, None, labels, raw=False)
            torch.cuda.synchronize()
        finally:
            if distributed:
                dist.destroy_process_group()
        return model.params.clone(), float(loss.item()), seen

    p0, l0, _ = run(False)
    p1, l1, seen = run(True)
    # the same buckets through the library's own ssak_allreduce (librccl.so dlopen'ed behind the C ABI, a communicator of one
    # rank): the exchange a host without torch.distributed would run
    p2, l2, seen2 = run(True, exchange="c")
    assert abs(l0 - l2) <= 1e-5 * abs(l0) and (p0 - p2).abs().max().item() < 0.5e-3 and seen2 == seen
    # not bitwise: under a process group the clip norm is summed bucket by bucket as the all-reduces complete (a different fp32
    # summation order: 23.555115 vs 23.555113), the first update differs in the last bit of a few parameters, and one flipped
    # bf16 rounding of a shadow weight moves the SECOND step's gradients by bf16 noise (Adam turns that into fractions of lr)
    assert abs(l0 - l1) <= 1e-5 * abs(l0) and (p0 - p1).abs().max().item() < 0.5e-3
    per_step = seen[:len(seen) // 2]
    spans = sorted(per_step)
    assert spans[0][0] == 0 and all(a[0] + a[1] <= b[0] for a, b in zip(spans, spans[1:]))
    n_train = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, oc)).num_trainable
    assert spans[-1][0] + spans[-1][1] == n_train
    assert sum(c for _, c in spans) >= n_train - 8 * len(spans)  # only alignment padding is left out
    # ... and they are exactly the static bucket list computed from the configuration (ssak_w2v2_grad_ranges), in order
    assert per_step == Wav2Vec2ForCTC.grad_ranges(_cfg_from_oracle(Wav2Vec2Config, oc))


def test_whisper_encoder_ctc_vs_hf(gold):
    """BASELINE config 4 composition (WhisperEncoder -> Linear -> CTC; no reference call site): tiny
    dimensions, log-mel features computed by the HIP front end (a13) from the waveform, against the golden made with
    transformers.WhisperEncoder."""
    import ssak_amd.hip as hip
    from ssak_amd.whisper import WhisperCTCConfig, WhisperEncoderForCTC
    from oracle import whisper_ref as WR
    z = gold("whisper_tiny.npz")
    oc = WR.WhisperCTCConfig.tiny()
    cfg = WhisperCTCConfig(vocab_size=oc.vocab_size, d_model=oc.d_model, encoder_layers=oc.encoder_layers,
                           encoder_attention_heads=oc.encoder_attention_heads, encoder_ffn_dim=oc.encoder_ffn_dim,
                           max_source_positions=oc.max_source_positions)
    model = WhisperEncoderForCTC(cfg).train()
    model.load_state_dict(WR.init_params(oc, 69))
    mel = hip.logmel_whisper(torch.tensor(z["wav"]).cuda(), n_samples=16000)
    assert np.abs(mel.cpu().numpy() - z["mel"]).max() < 2e-4
    out = model(mel, labels=torch.tensor(z["labels"]))
    assert out.logits.shape == (2, 50, 32)
    assert rel_l2(out.logits.cpu().numpy(), z["logits"]) < 2e-2
    assert abs(out.loss.item() - float(z["loss"])) < 2e-2 * float(z["loss"])
    model.grads[:model.num_trainable].fill_(float("nan"))  # an earlier step's values must be overwritten or zeroed, never kept
    model.backward()
    grads = {k[5:]: z[k] for k in z.files if k.startswith("grad/")}
    worst = _check_grads(model, grads, 6e-2)
    print("whisper tiny worst grad", worst)
    assert float(model.grad("encoder.layers.0.self_attn.k_proj.bias").abs().max()) == 0.0


def test_no_freeze_feature_encoder_gradients(mods):
    """--no_freeze (ssak/train/transformers/wav2vec_train.py:326-327 off): gradients of the conv feature encoder
    (conv weights, GroupNorm affine) against the oracle's autograd; tiny dims, 1599 conv0 frames per utterance."""
    Wav2Vec2Config, Wav2Vec2ForCTC, R = mods
    oc = R.W2V2Config.tiny().deterministic()
    p = R.init_params(oc, 17)
    rng = np.random.default_rng(6)
    x = R.zero_mean_unit_var_norm([rng.standard_normal(8000).astype(np.float32) for _ in range(3)])
    labels = R.pad_labels([list(rng.integers(1, 32, n)) for n in (8, 3, 6)])
    loss, logits, grads = R.loss_and_grads(p, oc, torch.tensor(x), None, torch.tensor(labels), freeze_feature_encoder=False)
    model = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, oc), freeze_feature_encoder=False).train()
    model.load_state_dict(p)
    assert model.num_trainable == model.num_params
    out = model(torch.tensor(x), labels=torch.tensor(labels))
    assert abs(out.loss.item() - loss.item()) < 2e-2 * loss.item()
    model.grads[:model.num_trainable].fill_(float("nan"))  # an earlier step's values must be overwritten or zeroed, never kept
    model.backward()
    fe = {n: g.numpy() for n, g in grads.items() if n.startswith("wav2vec2.feature_extractor.")}
    assert len(fe) == 9
    worst = _check_grads(model, fe, 8e-2)
    print("no_freeze worst FE grad", worst)
    _check_grads(model, {n: g.numpy() for n, g in grads.items() if not n.startswith("wav2vec2.feature_extractor.")}, 6e-2)
    # and the optimizer moves the conv weights (their GEMM layouts are refreshed after the step)
    from ssak_amd.trainer import AdamW
    w_before = model.param("wav2vec2.feature_extractor.conv_layers.3.conv.weight").clone()
    AdamW(model, lr=1e-3, warmup_steps=0).step()
    assert (model.param("wav2vec2.feature_extractor.conv_layers.3.conv.weight") - w_before).abs().max().item() > 0


def test_no_freeze_layer_norm_feature_encoder_gradients(mods):
    """--no_freeze with the XLSR-style feature encoder (conv + bias -> LayerNorm -> GELU on every layer, stable-layer-norm
    encoder, ragged lengths): conv weights / biases and LayerNorm affines against the oracle's autograd."""
    Wav2Vec2Config, Wav2Vec2ForCTC, R = mods
    oc = R.W2V2Config.tiny(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True).deterministic()
    p = R.init_params(oc, 23)
    rng = np.random.default_rng(8)
    lens = [8000, 6100, 7333]
    waves = [rng.standard_normal(n).astype(np.float32) for n in lens]
    x = R.zero_mean_unit_var_norm(waves)
    labels = R.pad_labels([list(rng.integers(1, 32, n)) for n in (7, 3, 5)])
    loss, logits, grads = R.loss_and_grads(p, oc, torch.tensor(x), lens, torch.tensor(labels), freeze_feature_encoder=False)
    model = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, oc), freeze_feature_encoder=False).train()
    model.load_state_dict(p)
    assert model.num_trainable == model.num_params
    out = model(torch.tensor(x), lengths=torch.tensor(lens), labels=torch.tensor(labels))
    assert abs(out.loss.item() - loss.item()) < 2e-2 * loss.item()
    model.grads[:model.num_trainable].fill_(float("nan"))  # an earlier step's values must be overwritten or zeroed, never kept
    model.backward()
    fe = {n: g.numpy() for n, g in grads.items() if n.startswith("wav2vec2.feature_extractor.")}
    assert len(fe) == 7 * 4  # conv weight + bias, LayerNorm weight + bias per layer
    worst = _check_grads(model, fe, 8e-2)
    print("no_freeze (layer-norm FE) worst FE grad", worst)
    _check_grads(model, {n: g.numpy() for n, g in grads.items() if not n.startswith("wav2vec2.feature_extractor.")}, 6e-2)


def test_fragment_ordered_weights_change_no_bit(mods):
    """SSAK_W2V2_OPT_FRAGMENT_WEIGHTS (opt-in): a training forward copies the kept layers' projection weights into the
    B-direct GEMM's fragment order and the forward products that pay read them instead of staging the weight through LDS.
    Since round 4 the option's off state runs those products on the four-wave kernel, whose K order is rotated per row
    panel, so the two states differ by bf16 summation-order noise, not by a bit pattern: logits and every gradient must agree
    to that noise (a wrong fragment map, a stale copy or a copy of a dropped layer gives O(1) errors) -- with LayerDrop (a
    dropped layer's copies are not refreshed and not read), dropout, and across an optimizer-style weight change between two
    steps (the copies are refreshed by every training forward).  The fragment form itself is bit-exact against fp32 on
    integer operands in tests/test_gpu_ops.py."""
    import ssak_amd.hip as hip
    Wav2Vec2Config, Wav2Vec2ForCTC, R = mods
    oc = R.W2V2Config.base(num_hidden_layers=3)
    p = R.init_params(oc, 5)
    rng = np.random.default_rng(3)
    B = 32  # 32 x 499 frames, the train step: the N = 768 products run on 192-row tiles
    x = torch.tensor(R.zero_mean_unit_var_norm([rng.standard_normal(160000).astype(np.float32) for _ in range(B)]))
    labels = torch.tensor(R.pad_labels([list(rng.integers(1, 32, 8)) for _ in range(B)]))
    keep = np.array([1, 0, 1], dtype=np.uint8)
    outs = []
    for frag in (1, 0):
        model = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, oc)).train()
        model.set_option(hip.W2V2_OPT_FRAGMENT_WEIGHTS, frag)
        # the comparison form is the eight-wave LDS kernel with the weight read K-major in the input-gradient products (round 4:
        # by default those read transposed copies on the four-wave kernel, whose K order is rotated per row panel)
        model.set_option(hip.W2V2_OPT_DYNAMIC_TILES, 1)
        model.set_option(hip.W2V2_OPT_TRANSPOSED_WEIGHTS, 0)
        model.load_state_dict(p)
        steps = []
        for step in range(2):
            out = model(x, labels=labels, layer_keep=keep if step == 0 else np.ones(3, dtype=np.uint8), dropout_seed=100 + step)
            model.grads[:model.num_trainable].fill_(float("nan"))
            model.backward()
            steps.append((out.logits.clone(), model.grads[:model.num_trainable].clone()))
            # an optimizer-style update of the bound buffers between the steps
            model.params[:model.num_trainable].mul_(1.01)
            model.sync_weights()
        outs.append(steps)
    assert hip.gemm_uses_fragments(B * 499, 768, 3072, pads_are_zero=True)
    for (lg1, g1), (lg0, g0) in zip(*outs):
        assert torch.isfinite(g1).all()
        e_l = float((lg1.float() - lg0.float()).norm() / lg0.float().norm())
        e_g = float((g1.float() - g0.float()).norm() / g0.float().norm())
        assert e_l < 5e-3 and e_g < 3e-2, (e_l, e_g)


@pytest.mark.parametrize("geometry", ["base_cg48", "xlsr_cg64"])
def test_positional_conv_direct_kernel_equals_toeplitz_gemm(mods, geometry):
    """The grouped positional convolution as a direct convolution with its input window resident in LDS (posconv.hip; group
    widths 48 = wav2vec2-base and 64 = XLSR-large, 128 taps) against the Toeplitz-GEMM form of rounds 1-2 on the same engine
    (per-handle option SSAK_W2V2_OPT_POSCONV_DIRECT): same logits and the same gradients -- the positional convolution's own
    (weight_g / weight_v / bias) and everything upstream of it (feature projection), which only the input-gradient pass
    reaches -- to bf16 summation-order noise, on ragged utterances whose frame counts (499, 312, 77 / 150) exercise the edge tile, and
    both against the CPU oracle at the usual bf16 bars."""
    import ssak_amd.hip as hip
    Wav2Vec2Config, Wav2Vec2ForCTC, R = mods
    if geometry == "base_cg48":
        oc = R.W2V2Config.base(num_hidden_layers=1).deterministic()
        lens = None
        n_samples = [160000, 100000, 25000]
    else:
        oc = R.W2V2Config.xlsr_large(num_hidden_layers=1).deterministic()
        lens = [48200, 24900]
        n_samples = lens
    p = R.init_params(oc, 41)
    rng = np.random.default_rng(9)
    if lens is None:  # group-norm model: no attention mask -> equal lengths per batch; run the three lengths as three batches
        batches = [(R.zero_mean_unit_var_norm([rng.standard_normal(n).astype(np.float32)] * 2), None) for n in n_samples]
    else:
        batches = [(R.zero_mean_unit_var_norm([rng.standard_normal(n).astype(np.float32) for n in lens]), lens)]
    labels = R.pad_labels([list(rng.integers(1, 32, 6)), list(rng.integers(1, 32, 4))])
    names = ["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0", "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1",
             "wav2vec2.encoder.pos_conv_embed.conv.bias", "wav2vec2.feature_projection.projection.weight", "wav2vec2.feature_projection.layer_norm.weight"]
    for x, ln in batches:
        outs = []
        for direct in (1, 0):
            model = Wav2Vec2ForCTC(_cfg_from_oracle(Wav2Vec2Config, oc)).train()
            model.set_option(hip.W2V2_OPT_POSCONV_DIRECT, direct)
            model.load_state_dict(p)
            out = model(torch.tensor(x), lengths=None if ln is None else torch.tensor(ln), labels=torch.tensor(labels))
            model.grads[:model.num_trainable].fill_(float("nan"))
            model.backward()
            outs.append((out.logits.float().cpu().numpy(), out.frame_lens, {n: model.grad(n).cpu().numpy().copy() for n in names}))
        (lg1, fl, g1), (lg0, _, g0) = outs
        nf = [lg1.shape[1]] * lg1.shape[0] if fl is None else fl.cpu().numpy()
        for b, f in enumerate(nf):
            assert rel_l2(lg1[b, :f], lg0[b, :f]) < 5e-3, (geometry, x.shape, b)
        for n in names:
            assert np.isfinite(g1[n]).all()
            assert rel_l2(g1[n], g0[n]) < 1e-2, (geometry, x.shape, n, rel_l2(g1[n], g0[n]))
    # the last batch against the CPU oracle
    x, ln = batches[-1]
    loss, logits, grads = R.loss_and_grads(p, oc, torch.tensor(x), ln, torch.tensor(labels))
    for b, f in enumerate(nf):
        assert rel_l2(lg1[b, :f], logits[b, :f].numpy()) < 2e-2
    for n in names:
        assert rel_l2(g1[n], grads[n].numpy()) < 6e-2, n
