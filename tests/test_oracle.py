"""CPU: the oracle restatements against the committed golden vectors (made by oracle/gen_golden.py
from transformers / torch, the libraries the reference calls on its hot path)."""
import os

import numpy as np
import pytest
import torch

from oracle import adamw_ref, ctc_ref, logmel_ref
from oracle import w2v2_ref as R
from conftest import ctc_case_names


def test_ctc_oracle_vs_torch(gold):
    z = gold("ctc_cases.npz")
    for name in ctc_case_names(z):
        g = lambda k: z[f"{name}/{k}"]
        loss, grad, nll = ctc_ref.ctc_loss_and_grad(g("logits"), g("labels"), g("in_lens"), 0,
                                                     str(g("reduction")), bool(g("zero_inf")))
        assert abs(loss - float(g("loss64"))) <= 1e-9 * max(1.0, abs(float(g("loss64")))), name
        assert np.abs(grad - g("grad64")).max() < 1e-9, name
        # torch's fp32 lattice (what the reference runs) within fp32 log-domain rounding
        assert abs(loss - float(g("loss"))) <= 1e-4 * max(1.0, abs(loss)), name
        assert np.abs(grad - g("grad")).max() <= 2e-3 * np.abs(grad).max() + 1e-7, name


def test_ctc_infeasible_is_zeroed(gold):
    z = gold("ctc_cases.npz")
    loss, grad, nll = ctc_ref.ctc_loss_and_grad(z["infeasible/logits"], z["infeasible/labels"], z["infeasible/in_lens"])
    assert nll[0] == 0.0 and np.all(grad[0] == 0.0) and nll[1] > 0
    loss2, _, nll2 = ctc_ref.ctc_loss_and_grad(z["infeasible/logits"], z["infeasible/labels"], z["infeasible/in_lens"],
                                               zero_infinity=False)
    assert np.isinf(loss2)


def test_normalize_lengths_labels(gold):
    z = gold("features.npz")
    waves = [z[f"wave{i}"] for i in range(4)]
    out = R.zero_mean_unit_var_norm(waves)
    assert np.abs(out - z["input_values"]).max() < 1e-5
    assert (out[1, 2500:] == 0).all()
    assert (R.conv_out_lengths(R.W2V2Config(), z["len_T"]) == z["len_F"]).all()
    assert int(R.conv_out_lengths(R.W2V2Config(), 160000)) == 499
    assert (R.pad_labels([[5, 6, 7, 8], [9], [10, 11]]) == z["labels_padded"]).all()


def test_specaug_indices(gold):
    z = gold("specaug.npz")
    for i, (B, S) in enumerate(((4, 499), (3, 499), (2, 49))):
        lens = list(z[f"lens{i}"])
        m = R.compute_mask_indices((B, S), 0.05, 10, None if i == 0 else lens, 2, rng=np.random.RandomState(100 + i))
        assert (m == z[f"mask{i}"]).all()


def test_greedy_decode(gold):
    z = gold("greedy.npz")
    vocab = [str(v) for v in z["vocab"]]
    onehot = np.eye(32, dtype=np.float32)[z["ids"]]
    mine = [" ".join(R.ids_to_text(x, vocab).split()) for x in R.greedy_ctc_ids(onehot)]
    assert mine == [" ".join(str(t).split()) for t in z["text"]]


def test_logmel(gold):
    z = gold("logmel.npz")
    assert np.abs(logmel_ref.mel_filters() - z["filters"]).max() < 1e-6
    m = logmel_ref.log_mel(z["wave"])
    assert m.shape == (80, 3000)
    assert np.abs(m[:, ::7] - z["mel_stride7"]).max() < 1e-4
    assert np.abs(m[:, :40] - z["mel_head"]).max() < 1e-4


def test_logmel_fp32_fft_route_is_within_the_bar(gold):
    """The device kernel's route (fp32, 200-point complex Stockham FFT 5 x 5 x 8 + real-input split) restated in numpy: its distance
    from the float64 oracle on the golden wave and on the weak-bin case that rules out 16-bit operand splits (a full-scale tone over
    noise at 1e-3 of its amplitude) -- the bar on the device is 2e-4."""
    from oracle import logmel_fft_ref
    z = gold("logmel.npz")
    w = z["wave"]
    assert np.abs(logmel_fft_ref.log_mel_fp32_fft(w) - logmel_ref.log_mel(w)).max() < 2e-5
    rng = np.random.default_rng(0)
    t = np.arange(160000) / 16000.0
    tone = (0.5 * np.sin(2 * np.pi * 440.0 * t) + 0.001 * rng.standard_normal(t.size)).astype(np.float32)
    assert np.abs(logmel_fft_ref.log_mel_fp32_fft(tone, 160000) - logmel_ref.log_mel(tone, 160000)).max() < 1.5e-4
    zf = logmel_fft_ref.stockham_fft200((rng.standard_normal((3, 200)) + 1j * rng.standard_normal((3, 200))).astype(np.complex64))
    assert zf.shape == (3, 200)


def test_logmel_bonjour(gold):
    import wave
    z = gold("logmel.npz")
    with wave.open(os.path.join(os.path.dirname(__file__), "golden", "bonjour.wav")) as f:
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
    assert np.abs(logmel_ref.log_mel(pcm)[:, :130] - z["bonjour_mel_head"]).max() < 1e-4


def test_adamw_clip_schedule(gold):
    z = gold("adamw.npz")
    for wd in (0.0, 0.01):
        p, m, v = z["p0"].copy(), np.zeros(1000, np.float32), np.zeros(1000, np.float32)
        for i, g in enumerate(z["grads"]):
            tot, coef = adamw_ref.clip_coef([g], 1.0)
            assert abs(tot - z[f"norm_wd{wd}"][i]) < 1e-3 * tot
            p, m, v = adamw_ref.adamw_step(p, g * np.float32(coef), m, v, i + 1,
                                           adamw_ref.linear_warmup_lr(1e-4, i, 2, 10), weight_decay=wd)
            assert np.abs(p - z[f"p_wd{wd}"][i]).max() < 1e-6
    # lr values pinned by the reference's own golden (tests/expected/train_transformers/trainer_state.json:12-13,27-28)
    assert abs(adamw_ref.linear_warmup_lr(1e-4, 1, 500, 10000) - 2e-7) < 1e-12
    assert abs(adamw_ref.linear_warmup_lr(1e-4, 2, 500, 10000) - 4e-7) < 1e-12


def _check_model(z, cfg, seed, lens, mask, tol_logits=2e-4):
    p = R.init_params(cfg, seed)
    loss, logits, grads = R.loss_and_grads(p, cfg, torch.tensor(z["x"]), lens, torch.tensor(z["labels"]),
                                           mask_time_indices=mask)
    assert abs(loss.item() - float(z["loss"])) < 2e-4 * max(1, abs(float(z["loss"])))
    assert np.abs(logits.numpy() - z["logits"]).max() < tol_logits
    gk = [k for k in z.files if k.startswith("grad/")]
    floor = 1e-3 * max(np.abs(z[k]).max() for k in gk)
    for k in gk:
        e = np.abs(grads[k[5:]].numpy() - z[k]).max() / max(np.abs(z[k]).max(), floor)
        assert e < 5e-3, (k, e)
    return p


def test_w2v2_tiny_vs_hf(gold):
    z = gold("w2v2_tiny.npz")
    cfg = R.W2V2Config.tiny().deterministic()
    p = _check_model(z, cfg, 69, None, torch.tensor(z["mask"]))
    st = {}
    R.forward(p, cfg, torch.tensor(z["x"]), None, None, mask_time_indices=torch.tensor(z["mask"]), stages=st)
    for k in z.files:
        if k.startswith("stage/"):
            assert np.abs(st[k[6:]].detach().numpy() - z[k]).max() < 1e-4, k


def test_w2v2_tiny_xlsr_vs_hf(gold):
    z = gold("w2v2_tiny_xlsr.npz")
    cfg = R.W2V2Config.tiny(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True).deterministic()
    _check_model(z, cfg, 70, list(z["lens"]), None)


def test_w2v2_base_vs_hf(gold):
    from oracle.gen_golden import base_inputs
    z = gold("w2v2_base.npz")
    x, labels = base_inputs()
    assert np.abs(x[:, :64] - z["x_head"]).max() < 1e-6 and (labels == z["labels"]).all()
    cfg = R.W2V2Config.base().deterministic()
    p = R.init_params(cfg, 69)
    assert sum(t.numel() for t in p.values()) == 94396320
    torch.set_num_threads(os.cpu_count())
    loss, logits, grads = R.loss_and_grads(p, cfg, torch.tensor(x), None, torch.tensor(labels))
    assert logits.shape == (2, 499, 32)
    assert abs(loss.item() - float(z["loss"])) < 1e-4 * float(z["loss"])
    assert np.abs(logits.numpy() - z["logits"]).max() < 2e-4
    for n, nr, hd in zip(z["grad_names"], z["grad_norms"], z["grad_heads"]):
        g = grads[str(n)].numpy()
        assert abs(np.sqrt((g.astype(np.float64) ** 2).sum()) - nr) <= 2e-3 * nr + 1e-7, n


# ------------------------------------------------------------------ forced alignment
ALIGN_CASES = ["tiny", "base", "blank5", "garbage", "flat", "tight", "infeasible"]


@pytest.mark.parametrize("name", ALIGN_CASES)
def test_align_oracle_vs_torch_golden(gold, name):
    """Bit-exact against the trellis / path of the reference's OWN get_trellis / backtrack (oracle/gen_golden_align.py imports
    /root/reference/ssak/utils/align_transcriptions.py and writes align.npz from them: row f1 is pinned)."""
    from oracle import align_ref
    z = gold("align.npz")
    em, tok, blank = z[f"{name}_emission"], z[f"{name}_tokens"].tolist(), int(z[f"{name}_blank"])
    tr = align_ref.get_trellis(em, tok, blank, bool(z[f"{name}_garbage"]))
    assert np.array_equal(tr, z[f"{name}_trellis"])
    if not int(z[f"{name}_ok"]):
        with pytest.raises(RuntimeError):
            align_ref.backtrack(tr, em, tok, blank)
        return
    path = align_ref.backtrack(tr, em, tok, blank)
    assert [p.token_index for p in path] == z[f"{name}_path_token"].tolist()
    assert [p.time_index for p in path] == z[f"{name}_path_time"].tolist()
    assert np.allclose([p.score for p in path], z[f"{name}_path_score"], rtol=1e-6, atol=0)


def test_align_trellis_is_best_path_by_enumeration():
    """trellis[F, L] equals the best score over ALL admissible frame labellings (tiny cases, exhaustive)."""
    from oracle import align_ref
    rng = np.random.default_rng(0)
    for F, V, L in [(6, 4, 2), (7, 5, 3), (8, 4, 4), (9, 3, 1)]:
        em = np.log(rng.dirichlet(np.ones(V), size=F)).astype(np.float32)
        tok = rng.integers(1, V, L).tolist()
        tr = align_ref.get_trellis(em, tok, 0)
        assert abs(float(tr[F, L]) - align_ref.best_score_by_enumeration(em, tok, 0)) < 1e-4


def test_align_path_properties_and_segments(gold):
    from oracle import align_ref
    z = gold("align.npz")
    em, tok = z["base_emission"], z["base_tokens"].tolist()
    path = align_ref.backtrack(align_ref.get_trellis(em, tok, 0), em, tok, 0)
    ti = [p.token_index for p in path]
    tt = [p.time_index for p in path]
    assert ti[0] == 0 and ti[-1] == len(tok) - 1 and all(b - a in (0, 1) for a, b in zip(ti, ti[1:]))
    assert all(b - a == 1 for a, b in zip(tt, tt[1:]))  # one point per frame, contiguous
    assert all(0.0 <= p.score <= 1.0 for p in path)
    transcript = "".join("ab cd"[t % 5] for t in tok)
    segs = align_ref.merge_repeats(transcript, path)
    assert len(segs) == len(tok) and segs[0].start == tt[0] and segs[-1].end == tt[-1] + 1
    assert all(a.end == b.start for a, b in zip(segs, segs[1:]))
    words = align_ref.merge_words(segs)
    assert "".join(w.label for w in words) == transcript.replace(" ", "")


@pytest.mark.parametrize("name", [c for c in ALIGN_CASES if c != "infeasible"])
def test_align_segments_vs_reference_merge(gold, gold_json, name):
    """merge_repeats / merge_words: the oracle's and the product's host mirrors (ssak_amd/align.py) against the segments the
    reference's own functions produced from the golden paths (align_segments.json, written by oracle/gen_golden_align.py from
    the imported ssak/utils/align_transcriptions.py:141-175)."""
    from oracle import align_ref
    from ssak_amd import align as prod
    z = gold("align.npz")
    want = gold_json("align_segments.json")["cases"][name]
    for mod in (align_ref, prod):
        path = [mod.Point(int(a), int(b), float(c)) for a, b, c in zip(z[f"{name}_path_token"], z[f"{name}_path_time"], z[f"{name}_path_score"])]
        segs = mod.merge_repeats(want["transcript"], path)
        assert [[s.label, s.start, s.end] for s in segs] == [w[:3] for w in want["segments"]]
        assert np.allclose([s.score for s in segs], [w[3] for w in want["segments"]], rtol=1e-12, atol=0)
        words = mod.merge_words(segs)
        assert [[s.label, s.start, s.end] for s in words] == [w[:3] for w in want["words"]]
        assert np.allclose([s.score for s in words], [w[3] for w in want["words"]], rtol=1e-12, atol=0)


# ------------------------------------------------------------------ evaluation metric
def test_wer_oracle_known_answers():
    from oracle import wer_ref
    from ssak_amd.synth import VOCAB
    enc = lambda s: [VOCAB.index("|" if c == " " else c) for c in s]
    assert wer_ref.word_edits("the cat sat".split(), "the cat sat".split()) == 0
    assert wer_ref.word_edits("the cat sat".split(), "the hat sat down".split()) == 2   # 1 substitution + 1 insertion
    assert wer_ref.word_edits("a b c d".split(), "b c".split()) == 2                     # 2 deletions
    assert wer_ref.word_edits([], "x y".split()) == 2
    assert wer_ref.format_words_for_wer("l'ami <unk> d'ici") == "l' ami d' ici"
    # argmax rows with repeats and blanks (a blank keeps the double l apart); labels ungrouped with -100 padding:
    # hypothesis "hello word" against "hello world" -> 1 edit / 2 words
    pred = [0] + enc("hheel") + [0] + enc("lo") + [0] + enc("  wworr") + [0] + enc("d") + [0, 0]
    lab = enc("hello world") + [-100] * 5
    e, n, wer = wer_ref.compute_metrics([pred], [lab], VOCAB, 0)
    assert e.tolist() == [1] and n.tolist() == [2] and wer == 0.5
    lab2 = enc("hello word") + [-100] * 6
    e, n, wer = wer_ref.compute_metrics([pred, pred], [lab, lab2], VOCAB, 0)
    assert e.tolist() == [1, 0] and n.tolist() == [2, 2] and wer == 0.25


# ------------------------------------------------------------------ audio ingest
@pytest.mark.parametrize("orig,new", [(44100, 16000), (48000, 16000), (8000, 16000), (22050, 16000)])
def test_resample_oracle_properties(orig, new):
    """torchaudio is absent (parity unpinned): the restated sinc resampler is held to its defining properties."""
    from oracle import resample_ref
    n = orig // 2  # half a second
    t = np.arange(n) / orig
    f0 = 440.0
    x = (0.5 * np.sin(2 * np.pi * f0 * t + 0.3)).astype(np.float32)
    y = resample_ref.resample(x, orig, new)
    assert len(y) == int(np.ceil(new * n / orig))
    ty = np.arange(len(y)) / new
    want = 0.5 * np.sin(2 * np.pi * f0 * ty + 0.3)
    core = slice(200, len(y) - 200)  # away from the zero-padded edges
    assert np.abs(y[core] - want[core]).max() < 2e-3
    dc = resample_ref.resample(np.ones(n, np.float32), orig, new)
    assert np.abs(dc[core] - 1.0).max() < 2e-3  # unit DC gain (rolloff 0.99 leaves < 0.2 % ripple)
    assert np.array_equal(resample_ref.resample(x, new, new), x)
    k, width, o, nn = resample_ref.sinc_resample_kernel(orig, new)
    assert k.shape == (nn, 1, 2 * width + o)


# ------------------------------------------------------------------ SpeechBrain-recipe head
def test_sb_head_oracle_vs_torch_modules():
    """oracle/sb_head_ref.py against the torch modules speechbrain wraps (nn.Linear / nn.BatchNorm1d on [B, C, T] /
    LeakyReLU), torch.optim.Adadelta and hand-computed NewBob values."""
    from oracle import sb_head_ref as S
    g = torch.Generator().manual_seed(0)
    B, T, H, D, V = 3, 11, 16, 24, 7
    sd = {}
    for k, din in ((1, H), (2, D), (3, D)):
        sd[f"0.linear{k}.w.weight"] = torch.randn(D, din, generator=g) * 0.3
        sd[f"0.linear{k}.w.bias"] = torch.randn(D, generator=g) * 0.1
        sd[f"0.bn{k}.norm.weight"] = torch.rand(D, generator=g) + 0.5
        sd[f"0.bn{k}.norm.bias"] = torch.randn(D, generator=g) * 0.1
    sd["1.w.weight"], sd["1.w.bias"] = torch.randn(V, D, generator=g), torch.randn(V, generator=g)
    feats = torch.randn(B, T, H, generator=g)
    mods, h = [], feats
    run = [(torch.zeros(D), torch.ones(D)) for _ in range(3)]
    for k in (1, 2, 3):
        lin = torch.nn.Linear(h.shape[-1], D)
        bn = torch.nn.BatchNorm1d(D)
        lin.weight.data, lin.bias.data = sd[f"0.linear{k}.w.weight"].clone(), sd[f"0.linear{k}.w.bias"].clone()
        bn.weight.data, bn.bias.data = sd[f"0.bn{k}.norm.weight"].clone(), sd[f"0.bn{k}.norm.bias"].clone()
        h = torch.nn.functional.leaky_relu(bn(lin(h).transpose(1, 2)).transpose(1, 2), 0.01)
        mods.append(bn)
    want = torch.nn.functional.linear(h, sd["1.w.weight"], sd["1.w.bias"])
    got = S.head_forward(sd, feats, None, True, dropouts=(0, 0, 0), running=run)
    assert torch.allclose(got, want, atol=1e-6)
    for i, bn in enumerate(mods):
        assert torch.allclose(run[i][0], bn.running_mean) and torch.allclose(run[i][1], bn.running_var)
    # utterance normalisation = layer_norm over everything but the batch axis
    x = torch.randn(2, 5, 8, generator=g)
    y = S.utt_norm(x)
    assert torch.allclose(y[1], (x[1] - x[1].mean()) / torch.sqrt(x[1].var(unbiased=False) + 1e-5), atol=1e-6)
    # CTC lengths: relative -> absolute by rounding; torch "mean" reduction
    logits = torch.randn(2, 12, 6, generator=g)
    tokens = torch.tensor([[1, 2, 3, 0], [4, 5, 0, 0]])
    l = S.ctc_cost(logits, tokens, [1.0, 0.74], [0.75, 0.5])
    lp = torch.log_softmax(logits, -1).transpose(0, 1)
    per = torch.nn.functional.ctc_loss(lp, tokens, torch.tensor([12, 9]), torch.tensor([3, 2]), 0, reduction="none")
    assert torch.allclose(l, (per / torch.tensor([3.0, 2.0])).mean())
    # Adadelta
    p = torch.randn(50, generator=g)
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.Adadelta([pt], lr=1.0, rho=0.95, eps=1e-8)
    pn, sq, acc = p.numpy().copy(), np.zeros(50, np.float32), np.zeros(50, np.float32)
    for _ in range(4):
        gr = torch.randn(50, generator=g)
        pt.grad = gr.clone()
        opt.step()
        S.adadelta_step(pn, gr.numpy(), sq, acc)
    assert np.allclose(pn, pt.detach().numpy(), rtol=1e-6, atol=1e-7)
    assert S.clip_coef([torch.full((4,), 5.0)], 5.0) == pytest.approx(0.5, rel=1e-6)
    # NewBob: 10 -> 9 improves 10 %; 9 -> 8.99 improves 0.11 % < 0.25 %: anneal; 8.0 -> 8.0: anneal again
    assert S.new_bob([10, 9, 8.99, 8.0, 8.0], 1.0, 0.8) == pytest.approx([1.0, 1.0, 0.8, 0.8, 0.64])
    assert S.new_bob([5.0, 5.0, 5.0], 1e-4, 0.9, patient=1) == pytest.approx([1e-4, 1e-4, 0.9e-4])


@pytest.mark.parametrize("p", [0.05, 0.1, 0.25, 0.5])
def test_dropout_mask_generator_quality(p):
    """The statistics the round-5 mask generator (one integer multiply per element: oracle/dropout_hash.py, common.h) is held to on a
    4 096 x 3 072 site: drop rate within 3 sigma; drop indicators uncorrelated between neighbouring columns / rows / diagonal
    neighbours / sites / seeds (|rho| < 2e-3: the iid standard error is 2.8e-4, the rank-one structure of the words widens it);
    per-row and per-column drop counts binomial (variance ratio within 6 %)."""
    from oracle import dropout_hash as DH
    q = DH.quality_report(p)
    assert abs(q["rate"] - DH.thresh16(p) / 65536.0) < 3 * q["rate_sigma"], q
    for k in ("col1", "col2", "col64", "row1", "diag", "site", "seed"):
        assert abs(q[k]) < 2e-3, (k, q)
    assert abs(q["row_count_var"] - 1) < 0.06 and abs(q["col_count_var"] - 1) < 0.06, q


@pytest.mark.parametrize("site_kind,cols,p", [("hid1", 768, 0.05), ("act", 3072, 0.05), ("attn", 499, 0.1)])
def test_dropout_masks_of_the_closest_row_pairs(site_kind, cols, p):
    """Where the one-multiply generator is weakest: two rows' words differ by ONE fixed odd factor r = rowkey_i / rowkey_j (mod 2^32)
    for every column, so a pair with a tiny |r| has masks that are functions of each other.  Over the 15 968 rows of a base site
    (B = 32 x 499 frames; for the attention site the first 15 968 (utterance, head, query) rows) and two of the engine's real step
    seeds, the 1 000 ordered row pairs with the smallest |r| are enumerated (oracle/dropout_hash.py smallest_key_ratio_pairs):
      * every pair with |r| > 7 agrees on its keep bits like independent rows, within 4 sigma (binomial over the site's columns);
      * the handful with |r| <= 7 follow the closed form of the structure (agreement 1 - 2p + 2p/r for r > 0, 1 - 2p for r < 0 --
        r = +1, identical masks, needs two row hashes that differ in bit 0 only) within 4 sigma, and there are at most 8 such pairs of
        1.3e8: at most 16 rows of 15 968 share part of their mask with another row.
    The training-dynamics comparison against torch's generator is tests/test_gpu_dropout.py::test_regularised_training_dynamics."""
    from oracle import dropout_hash as DH
    rows = 15968
    pe = DH.thresh16(p) / 65536.0
    q = pe * pe + (1 - pe) ** 2
    sig = np.sqrt(q * (1 - q) / cols)
    for layer, seed in zip((3, 7), DH.step_seed_sequence(69, 2)):
        site = {"hid1": DH.ds_hid1, "act": DH.ds_act, "attn": DH.ds_attn}[site_kind](layer)
        r, i, j = DH.smallest_key_ratio_pairs(seed, site, rows, top=1000)
        assert len(r) == 1000 and (np.abs(r[:-1]) <= np.abs(r[1:])).all() and (r % 2 != 0).all()
        agree = DH.keep_agreement(seed, site, i, j, cols, p)
        far = np.abs(r) > 7
        z = (agree[far] - q) / sig
        assert np.abs(z).max() < 4.0, (site_kind, hex(seed), float(np.abs(z).max()), int(r[far][np.argmax(np.abs(z))]))
        near = ~far
        assert near.sum() <= 8, (site_kind, hex(seed), r[near].tolist())
        for rr, a in zip(r[near], agree[near]):
            want = DH.ratio_agreement_expected(int(rr), p)
            assert abs(a - want) < 4.0 * np.sqrt(max(want * (1 - want), 1e-4) / cols), (site_kind, hex(seed), int(rr), float(a), want)


def test_dropout_ratio_structure_closed_form():
    """The closed form used above, on constructed keys: words w and r w (mod 2^32) for uniform w agree on their keep bits with
    probability 1 - 2p + 2p/r (r = 3, 5, 7), 1 - 2p (r < 0); from |r| ~ 1/p on the excess over independence is below 2 p^2."""
    from oracle import dropout_hash as DH
    rng = np.random.default_rng(5)
    w = rng.integers(0, 1 << 32, 2_000_000, dtype=np.uint64)
    for p in (0.05, 0.1):
        t = np.uint64(DH.thresh16(p) << 16)
        pe = DH.thresh16(p) / 65536.0
        for r in (3, 5, 7, -1, -3, 21, 101):
            w2 = (w * np.uint64(r % (1 << 32))) & np.uint64(0xFFFFFFFF)
            a = float(((w >= t) == (w2 >= t)).mean())
            if abs(r) * pe < 1:
                assert abs(a - DH.ratio_agreement_expected(r, p)) < 1e-3, (p, r, a)
            assert a - (pe * pe + (1 - pe) ** 2) < 2 * pe / abs(r) + 1e-3


KNOWN_COLMUL = [4102127511, 2960243731, 3929676619, 1507823451]
KNOWN_ROWKEY = [2898601965, 3335941947, 1865774827]
KNOWN_MASK = [[1, 1, 1, 1, 0, 1, 1, 0], [1, 0, 0, 1, 0, 1, 1, 1], [1, 1, 1, 1, 0, 1, 0, 1], [1, 1, 1, 1, 1, 1, 0, 0], [1, 1, 0, 0, 1, 1, 0, 1]]


def test_dropout_mask_known_answers():
    """Known answers of the integer functions (guards the numpy restatement itself against an accidental edit; the device pin is
    tests/test_gpu_dropout.py::test_dropout_hash_matches_device)."""
    from oracle import dropout_hash as DH
    assert [int(v) for v in DH.colmul(np.arange(4, dtype=np.uint64))] == KNOWN_COLMUL
    assert [int(v) for v in DH.rowkey(0xDEADBEEFCAFEF00D, 18, np.arange(3, dtype=np.uint64))] == KNOWN_ROWKEY
    m = DH.keep_mask(7, 3, (5, 8), 0.25)
    assert m.astype(int).tolist() == KNOWN_MASK
