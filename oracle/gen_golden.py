"""Generate the golden vectors under tests/golden/.  TEST INFRASTRUCTURE ONLY.

Run in the BUILD container (``python -m oracle.gen_golden``); it imports the third-party classes
the reference calls on its hot path (``transformers.Wav2Vec2ForCTC`` &c., ``torch``), runs them
on seeded inputs and writes small ``.npz`` fixtures.  It also asserts that the oracle's own
restatements (``oracle/w2v2_ref.py``, ``ctc_ref.py``, ``logmel_ref.py``, ``adamw_ref.py``) agree
with those classes, i.e. it is the pinning step.  Nothing in ``tests/`` or on
the GPU box needs ``transformers`` or ``/root/reference``: they read the ``.npz`` files only.
"""
from __future__ import annotations

import os
import shutil
import sys

import numpy as np
import torch

from . import adamw_ref, ctc_ref, logmel_ref
from . import w2v2_ref as R

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def synth_wave(rng: np.random.Generator, n: int) -> np.ndarray:
    """BASELINE.md synthetic utterance: Gaussian sigma 0.1 + 220/440/880 Hz sines (amp 0.05), clipped."""
    t = np.arange(n) / 16000.0
    x = rng.standard_normal(n) * 0.1
    for f in (220.0, 440.0, 880.0):
        x = x + 0.05 * np.sin(2 * np.pi * f * t + rng.uniform(0, 2 * np.pi))
    return np.clip(x, -1, 1).astype(np.float32)


def hf_model(cfg: R.W2V2Config, params):
    import transformers
    kw = cfg.to_hf_kwargs()
    kw["mask_time_prob"] = max(kw["mask_time_prob"], 0.05)  # keeps masked_spec_embed registered; inert under eval()
    m = transformers.Wav2Vec2ForCTC(transformers.Wav2Vec2Config(**kw))
    m.load_state_dict(params, strict=True)
    return m


def gen_ctc():
    rng = np.random.default_rng(7)
    cases = {}

    def add(name, F, V, labels, in_lens, reduction, zero_inf, scale=1.0):
        B = len(labels)
        logits = (rng.standard_normal((B, F, V)) * scale).astype(np.float32)
        lab = R.pad_labels(labels)
        lt = torch.tensor(logits, requires_grad=True)
        lm = torch.tensor(lab) >= 0
        lp = torch.nn.functional.log_softmax(lt, dim=-1, dtype=torch.float32).transpose(0, 1)
        loss = torch.nn.functional.ctc_loss(lp, torch.tensor(lab).masked_select(lm), torch.tensor(in_lens),
                                            lm.sum(-1), blank=0, reduction=reduction, zero_infinity=zero_inf)
        loss.backward()
        # the same through torch in float64: the tight pin for the float64 restatement (torch's fp32 lattice
        # carries ~1 ulp(nll) of log-domain error, i.e. ~1e-4 relative on long utterances)
        l64 = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
        lp64 = torch.nn.functional.log_softmax(l64, dim=-1).transpose(0, 1)
        loss64 = torch.nn.functional.ctc_loss(lp64, torch.tensor(lab).masked_select(lm), torch.tensor(in_lens),
                                              lm.sum(-1), blank=0, reduction=reduction, zero_infinity=zero_inf)
        loss64.backward()
        o_loss, o_grad, _ = ctc_ref.ctc_loss_and_grad(logits, lab, in_lens, 0, reduction, zero_inf)
        assert abs(o_loss - loss64.item()) <= 1e-9 * max(1, abs(loss64.item())), (name, o_loss, loss64.item())
        assert np.abs(o_grad - l64.grad.numpy()).max() < 1e-9, (name, np.abs(o_grad - l64.grad.numpy()).max())
        assert abs(o_loss - loss.item()) <= 1e-4 * max(1, abs(loss.item())), (name, o_loss, loss.item())
        assert np.abs(o_grad - lt.grad.numpy()).max() < 2e-3 * np.abs(o_grad).max() + 1e-7, name
        for k, v in dict(logits=logits, labels=lab, in_lens=np.asarray(in_lens, np.int32), loss=np.float32(loss.item()),
                         grad=lt.grad.numpy(), loss64=np.float64(loss64.item()), grad64=l64.grad.numpy(),
                         reduction=np.array(reduction), zero_inf=np.array(zero_inf)).items():
            cases[f"{name}/{k}"] = v

    add("basic", 50, 32, [list(rng.integers(1, 32, 12)), list(rng.integers(1, 32, 7))], [50, 41], "mean", True)
    add("repeats", 30, 8, [[3, 3, 3, 4, 4], [1, 1], [2]], [30, 30, 10], "mean", True)
    add("empty_target", 20, 8, [[], [5, 6]], [20, 15], "mean", True)
    add("infeasible", 12, 8, [[1, 1, 1, 1, 1, 1, 1], [2, 3]], [12, 12], "mean", True)  # needs 13 frames
    add("sum_reduction", 40, 32, [list(rng.integers(1, 32, 10))] * 3, [40, 33, 25], "sum", True)
    add("peaky", 64, 32, [list(rng.integers(1, 32, 20))] * 2, [64, 60], "mean", True, scale=8.0)
    add("base_shape", 499, 32, [list(rng.integers(1, 32, n)) for n in (120, 60, 97, 1)], [499, 499, 300, 499], "mean", True)
    add("exact_fit", 9, 8, [[1, 2, 3, 4, 5], [1, 1, 2, 2]], [9, 6], "mean", True)
    np.savez_compressed(os.path.join(GOLD, "ctc_cases.npz"), **cases)
    print("ctc_cases ok")


def gen_features():
    import transformers
    rng = np.random.default_rng(11)
    waves = [synth_wave(rng, n) * s + o for n, s, o in ((4000, 1.0, 0.0), (2500, 0.3, 0.1), (3999, 2.0, -0.5), (400, 1.0, 0.0))]
    fe = transformers.Wav2Vec2FeatureExtractor(return_attention_mask=True)
    out = fe(waves, sampling_rate=16000, padding="longest", return_tensors="np")
    mine = R.zero_mean_unit_var_norm(waves)
    assert np.abs(mine - out["input_values"]).max() < 1e-5
    d = {f"wave{i}": w for i, w in enumerate(waves)}
    d["input_values"] = out["input_values"].astype(np.float32)
    d["attention_mask"] = out["attention_mask"].astype(np.int32)
    # frame-length table (modeling_wav2vec2.py:997-1016) through the HF implementation
    m = transformers.Wav2Vec2ForCTC(transformers.Wav2Vec2Config(num_hidden_layers=1))
    T = np.array([400, 401, 479, 480, 639, 640, 1000, 15999, 16000, 16001, 79999, 160000, 240000, 480000, 2240400])
    Fh = m._get_feat_extract_output_lengths(torch.tensor(T)).numpy()
    assert (Fh == R.conv_out_lengths(R.W2V2Config(), T)).all()
    d["len_T"], d["len_F"] = T, Fh
    # collator label padding (wav2vec_train.py:89-100)
    labs = [[5, 6, 7, 8], [9], [10, 11]]
    d["labels_padded"] = R.pad_labels(labs)
    np.savez_compressed(os.path.join(GOLD, "features.npz"), **d)
    # SpecAugment mask indices with numpy global RNG, as HF draws them
    from transformers.models.wav2vec2.modeling_wav2vec2 import _compute_mask_indices
    d = {}
    for i, (B, S, lens) in enumerate(((4, 499, None), (3, 499, [499, 300, 120]), (2, 49, [49, 20]))):
        np.random.seed(100 + i)
        am = None if lens is None else (torch.arange(S)[None, :] < torch.tensor(lens)[:, None]).long()
        hf = _compute_mask_indices((B, S), 0.05, 10, attention_mask=am, min_masks=2)
        mine = R.compute_mask_indices((B, S), 0.05, 10, lens, 2, rng=np.random.RandomState(100 + i))
        assert (hf == mine).all(), i
        d[f"mask{i}"] = hf
        d[f"lens{i}"] = np.array(lens if lens else [S] * B)
    np.savez_compressed(os.path.join(GOLD, "specaug.npz"), **d)
    # greedy decode with a synthetic 32-symbol vocabulary
    import json
    import tempfile
    vocab = ["<pad>", "<s>", "</s>", "<unk>", "|"] + [chr(ord("a") + i) for i in range(26)] + ["'"]
    with tempfile.TemporaryDirectory() as td:
        vf = os.path.join(td, "vocab.json")
        json.dump({t: i for i, t in enumerate(vocab)}, open(vf, "w"))
        tok = transformers.Wav2Vec2CTCTokenizer(vf, unk_token="<unk>", pad_token="<pad>", word_delimiter_token="|")
        ids = rng.integers(0, 32, size=(6, 60))
        ids[0, :10] = 0
        ids[1] = np.repeat(rng.integers(0, 32, 12), 5)
        ids[2, ::2] = 0
        ids[ids == 1] = 0
        ids[ids == 2] = 0
        ids[ids == 3] = 0
        hf_txt = tok.batch_decode(ids)
        onehot = np.eye(32, dtype=np.float32)[ids]
        mine = [R.ids_to_text(x, vocab) for x in R.greedy_ctc_ids(onehot)]
        # HF additionally collapses runs of spaces produced by consecutive delimiters
        mine = [" ".join(s.split()) for s in mine]
        assert mine == [" ".join(s.split()) for s in hf_txt], (mine, hf_txt)
    np.savez_compressed(os.path.join(GOLD, "greedy.npz"), ids=ids, text=np.array(hf_txt), vocab=np.array(vocab))
    print("features/specaug/greedy ok")


def gen_logmel():
    import transformers
    fe = transformers.WhisperFeatureExtractor()
    rng = np.random.default_rng(5)
    w = synth_wave(rng, 52345)
    a = fe(w, sampling_rate=16000, return_tensors="np").input_features[0]
    assert np.abs(a - logmel_ref.log_mel(w)).max() < 1e-4
    d = dict(wave=w, mel_stride7=a[:, ::7].astype(np.float32), mel_head=a[:, :40].astype(np.float32),
             filters=fe.mel_filters.astype(np.float32))
    src = "/root/reference/tests/data/audio/bonjour.wav"  # data file held by the reference's tests (PCM16 mono 16 kHz)
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(GOLD, "bonjour.wav"))
        import wave as _w
        with _w.open(src) as f:
            pcm = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
        b = fe(pcm, sampling_rate=16000, return_tensors="np").input_features[0]
        assert np.abs(b - logmel_ref.log_mel(pcm)).max() < 1e-4
        d["bonjour_mel_head"] = b[:, :130].astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "logmel.npz"), **d)
    print("logmel ok")


def gen_adamw():
    import transformers
    rng = np.random.default_rng(3)
    n = 1000
    p0 = rng.standard_normal(n).astype(np.float32)
    gs = [rng.standard_normal(n).astype(np.float32) * s for s in (1.0, 0.01, 3.0)]
    d = {"p0": p0, "grads": np.stack(gs)}
    for wd in (0.0, 0.01):
        p = torch.nn.Parameter(torch.tensor(p0))
        opt = torch.optim.AdamW([p], lr=1e-4, weight_decay=wd)
        sch = transformers.get_linear_schedule_with_warmup(opt, 2, 10)
        mp, mm, mv = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
        outs, norms = [], []
        for i, g in enumerate(gs):
            p.grad = torch.tensor(g)
            nrm = torch.nn.utils.clip_grad_norm_([p], 1.0)
            opt.step()
            sch.step()
            tot, coef = adamw_ref.clip_coef([g], 1.0)
            lr = adamw_ref.linear_warmup_lr(1e-4, i, 2, 10)
            mp, mm, mv = adamw_ref.adamw_step(mp, g * np.float32(coef), mm, mv, i + 1, lr, weight_decay=wd)
            assert abs(tot - nrm.item()) < 1e-3 * tot
            assert np.abs(mp - p.detach().numpy()).max() < 1e-6, (i, np.abs(mp - p.detach().numpy()).max())
            outs.append(p.detach().numpy().copy())
            norms.append(nrm.item())
        d[f"p_wd{wd}"] = np.stack(outs)
        d[f"norm_wd{wd}"] = np.array(norms, np.float32)
    # the reference's own fixture pins the schedule: lr after steps 1,2 = 2e-7,4e-7 (trainer_state.json:12-13,27-28)
    assert abs(adamw_ref.linear_warmup_lr(1e-4, 1, 500, 10000) - 2e-7) < 1e-12
    assert abs(adamw_ref.linear_warmup_lr(1e-4, 2, 500, 10000) - 4e-7) < 1e-12
    np.savez_compressed(os.path.join(GOLD, "adamw.npz"), **d)
    print("adamw ok")


def run_hf(cfg, params, x, lengths, labels, mask=None):
    m = hf_model(cfg, params)
    m.freeze_feature_encoder()
    m.eval()  # deterministic parity configuration: dropouts/layerdrop/specaug inactive, grads still flow
    am = None
    if lengths is not None:
        am = (torch.arange(x.shape[1])[None, :] < torch.tensor(lengths)[:, None]).long()
    if mask is not None:
        # route the explicit SpecAugment mask through Wav2Vec2Model.forward(mask_time_indices=...)
        out = m.wav2vec2(torch.tensor(x), attention_mask=am, mask_time_indices=torch.tensor(mask))
        logits = m.lm_head(out[0])
        lm = torch.tensor(labels) >= 0
        il = m._get_feat_extract_output_lengths(am.sum(-1) if am is not None else torch.full((x.shape[0],), x.shape[1]))
        lp = torch.nn.functional.log_softmax(logits, dim=-1, dtype=torch.float32).transpose(0, 1)
        loss = torch.nn.functional.ctc_loss(lp, torch.tensor(labels).masked_select(lm), il, lm.sum(-1), blank=0,
                                            reduction="mean", zero_infinity=True)
    else:
        o = m(torch.tensor(x), attention_mask=am, labels=torch.tensor(labels))
        loss, logits = o.loss, o.logits
    loss.backward()
    grads = {n: p.grad.detach().numpy() for n, p in m.named_parameters() if p.grad is not None}
    return loss.item(), logits.detach().numpy(), grads


def check_ref(cfg, params, x, lengths, labels, hf, mask=None, tol=2e-4):
    loss, logits, grads = R.loss_and_grads(params, cfg, torch.tensor(x), lengths, torch.tensor(labels),
                                           mask_time_indices=None if mask is None else torch.tensor(mask))
    hl, hlog, hg = hf
    assert abs(loss.item() - hl) < tol * max(1, abs(hl)), (loss.item(), hl)
    assert np.abs(logits.numpy() - hlog).max() < tol, np.abs(logits.numpy() - hlog).max()
    worst = 0.0
    # floor for the relative error: k_proj.bias gradients are identically zero in exact arithmetic
    # (softmax is shift-invariant along keys), so only rounding noise is left there
    floor = 1e-3 * max(np.abs(g).max() for g in hg.values())
    for n, g in hg.items():
        e = np.abs(grads[n].numpy() - g).max() / max(np.abs(g).max(), floor)
        worst = max(worst, e)
        assert e < 5e-3, (n, e)
    return worst


def gen_w2v2():
    rng = np.random.default_rng(1234)
    # ---- tiny, base topology (group norm, post-LN), no attention mask, with SpecAugment mask
    cfg = R.W2V2Config.tiny().deterministic()
    params = R.init_params(cfg, seed=69)
    x = R.zero_mean_unit_var_norm([synth_wave(rng, 8000) for _ in range(3)])
    Fr = int(R.conv_out_lengths(cfg, 8000))
    labels = R.pad_labels([list(rng.integers(1, 32, n)) for n in (9, 4, 6)])
    mask = R.compute_mask_indices((3, Fr), 0.2, 3, None, 2, rng=np.random.RandomState(0))
    hf = run_hf(cfg, params, x, None, labels, mask)
    w = check_ref(cfg, params, x, None, labels, hf, mask)
    print("tiny/base-topology: oracle vs HF worst rel grad err", w)
    stages = {}
    R.forward(params, cfg, torch.tensor(x), None, torch.tensor(labels), mask_time_indices=torch.tensor(mask), stages=stages)
    d = dict(x=x, labels=labels, mask=mask, loss=np.float32(hf[0]), logits=hf[1])
    for n, g in hf[2].items():
        d["grad/" + n] = g.astype(np.float32)
    for n in ("conv0", "conv6", "feat_proj", "pos_conv_added", "encoder_in", "layer0", "layer1"):
        d["stage/" + n] = stages[n].detach().numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "w2v2_tiny.npz"), **d)
    # ---- tiny, XLSR topology (layer norm convs with bias, pre-LN), ragged lengths + attention mask
    cfg2 = R.W2V2Config.tiny(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True).deterministic()
    params2 = R.init_params(cfg2, seed=70)
    lens = [8000, 5000, 6500]
    x2 = R.zero_mean_unit_var_norm([synth_wave(rng, n) for n in lens])
    labels2 = R.pad_labels([list(rng.integers(1, 32, n)) for n in (7, 3, 5)])
    hf2 = run_hf(cfg2, params2, x2, lens, labels2)
    w = check_ref(cfg2, params2, x2, lens, labels2, hf2)
    print("tiny/xlsr-topology: oracle vs HF worst rel grad err", w)
    d = dict(x=x2, lens=np.array(lens), labels=labels2, loss=np.float32(hf2[0]), logits=hf2[1])
    for n, g in hf2[2].items():
        d["grad/" + n] = g.astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "w2v2_tiny_xlsr.npz"), **d)
    # ---- base config, BASELINE shapes (B=2, T=160000 -> 499 frames): logits + loss + grad summaries
    cfg3 = R.W2V2Config.base().deterministic()
    params3 = R.init_params(cfg3, seed=69)
    x3 = R.zero_mean_unit_var_norm([synth_wave(rng, 160000) for _ in range(2)])
    labels3 = R.pad_labels([list(rng.integers(5, 32, n)) for n in (100, 64)])
    hf3 = run_hf(cfg3, params3, x3, None, labels3)
    w = check_ref(cfg3, params3, x3, None, labels3, hf3)
    print("base: oracle vs HF worst rel grad err", w, "loss", hf3[0])
    d = dict(labels=labels3, loss=np.float32(hf3[0]), logits=hf3[1].astype(np.float32), wave_seed=np.int64(1234))
    # inputs are regenerated from the seed on the GPU box; keep a checksum to detect drift
    d["x_checksum"] = np.array([x3.astype(np.float64).sum(), np.abs(x3).astype(np.float64).sum()])
    d["x_head"] = x3[:, :64]
    names, norms, heads = [], [], []
    for n, g in hf3[2].items():
        names.append(n)
        norms.append(np.sqrt((g.astype(np.float64) ** 2).sum()))
        heads.append(g.reshape(-1)[:16].astype(np.float32))
    d["grad_names"], d["grad_norms"], d["grad_heads"] = np.array(names), np.array(norms), np.stack(heads)
    np.savez_compressed(os.path.join(GOLD, "w2v2_base.npz"), **d)
    print("w2v2 ok")


def gen_whisper():
    """Whisper encoder + CTC head composition (BASELINE config 4; no reference call site): tiny dims, 100 mel frames."""
    import transformers
    from . import whisper_ref as WR
    cfg = WR.WhisperCTCConfig.tiny()
    p = WR.init_params(cfg, 69)
    hc = transformers.WhisperConfig(num_mel_bins=cfg.num_mel_bins, d_model=cfg.d_model, encoder_layers=cfg.encoder_layers,
                                    encoder_attention_heads=cfg.encoder_attention_heads, encoder_ffn_dim=cfg.encoder_ffn_dim,
                                    max_source_positions=cfg.max_source_positions, dropout=0.0, attention_dropout=0.0,
                                    activation_dropout=0.0, encoder_layerdrop=0.0, decoder_layers=1, decoder_attention_heads=4,
                                    decoder_ffn_dim=128)
    from transformers.models.whisper.modeling_whisper import WhisperEncoder
    enc = WhisperEncoder(hc)
    sd = {k[len("encoder."):]: v for k, v in p.items() if k.startswith("encoder.")}
    enc.load_state_dict(sd, strict=True)
    head = torch.nn.Linear(cfg.d_model, cfg.vocab_size)
    head.weight.data.copy_(p["ctc_head.weight"])
    head.bias.data.copy_(p["ctc_head.bias"])
    rng = np.random.default_rng(21)
    wav = np.stack([synth_wave(rng, 16000) for _ in range(2)])
    mel = np.stack([logmel_ref.log_mel(w, n_samples=16000) for w in wav])  # [2, 80, 100]
    labels = R.pad_labels([list(rng.integers(1, 32, n)) for n in (9, 5)])
    enc.train(False)
    hs = enc(torch.tensor(mel)).last_hidden_state
    logits = head(hs)
    lm = torch.tensor(labels) >= 0
    lp = torch.nn.functional.log_softmax(logits, dim=-1, dtype=torch.float32).transpose(0, 1)
    loss = torch.nn.functional.ctc_loss(lp, torch.tensor(labels).masked_select(lm), torch.full((2,), hs.shape[1]), lm.sum(-1),
                                        blank=0, reduction="mean", zero_infinity=True)
    loss.backward()
    o_loss, o_logits, o_grads = WR.loss_and_grads(p, cfg, torch.tensor(mel), torch.tensor(labels))
    assert abs(o_loss.item() - loss.item()) < 1e-4 * abs(loss.item())
    assert np.abs(o_logits.numpy() - logits.detach().numpy()).max() < 2e-4
    d = dict(wav=wav, mel=mel.astype(np.float32), labels=labels, loss=np.float32(loss.item()), logits=logits.detach().numpy())
    hg = {"encoder." + n: q.grad for n, q in enc.named_parameters() if q.grad is not None}
    hg["ctc_head.weight"], hg["ctc_head.bias"] = head.weight.grad, head.bias.grad
    floor = 1e-3 * max(float(g.abs().max()) for g in hg.values())
    for n, g in hg.items():
        e = float((o_grads[n] - g).abs().max()) / max(float(g.abs().max()), floor)
        assert e < 5e-3, (n, e)
        d["grad/" + n] = g.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "whisper_tiny.npz"), **d)
    print("whisper ok; loss", loss.item())


def base_inputs():
    """Re-create gen_w2v2()'s base-config inputs (used by tests on the GPU box).  Must replay the
    same RNG stream as gen_w2v2 up to the base section."""
    rng = np.random.default_rng(1234)
    for _ in range(3):
        synth_wave(rng, 8000)
    for n in (9, 4, 6):
        rng.integers(1, 32, n)
    for n in (8000, 5000, 6500):
        synth_wave(rng, n)
    for n in (7, 3, 5):
        rng.integers(1, 32, n)
    x3 = R.zero_mean_unit_var_norm([synth_wave(rng, 160000) for _ in range(2)])
    labels3 = R.pad_labels([list(rng.integers(5, 32, n)) for n in (100, 64)])
    return x3, labels3


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    which = sys.argv[1:] or ["ctc", "features", "logmel", "adamw", "w2v2", "whisper"]
    for w in which:
        globals()["gen_" + w]()


if __name__ == "__main__":
    main()
