"""GPU: forced alignment through the C ABI against the torch-op goldens and the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = ["tiny", "base", "blank5", "garbage", "flat", "tight", "infeasible"]


@pytest.fixture(scope="module")
def align():
    from ssak_amd import align as A
    return A


@pytest.mark.parametrize("name", CASES)
def test_align_golden_bit_exact(align, gold, name):
    """Trellis identical bit for bit (fp32 add / max, double cumsum of column 0, the inf borders), same path."""
    z = gold("align.npz")
    em = torch.from_numpy(z[f"{name}_emission"]).cuda()
    tok, blank = z[f"{name}_tokens"].tolist(), int(z[f"{name}_blank"])
    trellis = align.get_trellis(em, tok, blank_id=blank, first_as_garbage=bool(z[f"{name}_garbage"]))
    got, want = trellis.cpu().numpy(), z[f"{name}_trellis"]
    if name == "garbage":
        # column 0 of this variant is log(1 - exp(.)): torch's vectorised exp / log differ in the last ulp between host
        # CPUs (the golden was written on another machine), so only this case is compared with a tolerance
        fin = np.isfinite(want)
        assert np.array_equal(fin, np.isfinite(got)) and np.array_equal(got[~fin], want[~fin])
        assert np.allclose(got[fin], want[fin], rtol=2e-6, atol=1e-5)
    else:
        assert np.array_equal(got, want)
    if not int(z[f"{name}_ok"]):
        with pytest.raises(RuntimeError, match="Failed to align"):
            align.backtrack(trellis, em, tok, blank_id=blank)
        return
    path = align.backtrack(trellis, em, tok, blank_id=blank)
    assert [p.token_index for p in path] == z[f"{name}_path_token"].tolist()
    assert [p.time_index for p in path] == z[f"{name}_path_time"].tolist()
    assert np.allclose([p.score for p in path], z[f"{name}_path_score"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("F,V,L", [(700, 40, 300), (1500, 32, 1100), (3000, 51, 2500)])
def test_align_vs_oracle_multi_wave(align, F, V, L):
    """Transcripts wider than one wave / one workgroup pass (several columns per thread) against the CPU oracle."""
    from oracle import align_ref
    g = torch.Generator().manual_seed(F + L)
    logits = torch.randn(F, V, generator=g) * 2
    tok = torch.randint(1, V, (L,), generator=g).tolist()
    pos = sorted(torch.randperm(F, generator=g)[:L].tolist())
    logits[:, 0] += 3.0
    for k, p in enumerate(pos):
        logits[p, tok[k]] += 8.0
    em = torch.log_softmax(logits, -1)
    ref_tr = align_ref.get_trellis(em.numpy(), tok, 0)
    ref_path = align_ref.backtrack(ref_tr, em.numpy(), tok, 0)
    trellis, path = align.forced_align(em.cuda(), tok, 0)
    assert np.array_equal(trellis.cpu().numpy(), ref_tr)
    assert [(p.token_index, p.time_index) for p in path] == [(p.token_index, p.time_index) for p in ref_path]
    assert np.allclose([p.score for p in path], [p.score for p in ref_path], rtol=1e-6)


def test_align_long_audio_properties(align):
    """A long recording (12 000 frames, 3 000 characters; K = 3 columns per thread): checks that need no CPU trellis loop.
    The recursion is verified cell by cell in one vectorised expression (every row from the row above it, bit-exact),
    column 0 and the borders directly, and the walk by the oracle's backtrack run on the DEVICE trellis."""
    from oracle import align_ref
    F, V, L = 12000, 32, 3000
    g = torch.Generator().manual_seed(9)
    logits = torch.randn(F, V, generator=g)
    tok = torch.randint(1, V, (L,), generator=g).tolist()
    pos = sorted(torch.randperm(F, generator=g)[:L].tolist())
    logits[:, 0] += 3.0
    for k, p in enumerate(pos):
        logits[p, tok[k]] += 9.0
    em_h = torch.log_softmax(logits, -1)
    trellis, path = align.forced_align(em_h.cuda(), tok, 0)
    tr, em = trellis.cpu().numpy(), em_h.numpy()
    et = em[:, tok]
    want = np.maximum(tr[:-1, 1:] + em[:, :1], np.maximum(tr[:-1, 1:] + et, tr[:-1, :-1] + et))
    assert np.array_equal(tr[1:, 1:], want)
    c0 = np.cumsum(em[:, 0].astype(np.float64)).astype(np.float32)
    assert tr[0, 0] == 0 and np.array_equal(tr[1:F + 1 - L, 0], c0[:F - L]) and np.isinf(tr[F + 1 - L:, 0]).all()
    assert (tr[0, 1:] == -np.inf).all() and np.isfinite(tr[F, L])
    ref = align_ref.backtrack(tr, em, tok, 0)
    assert [(p.token_index, p.time_index) for p in path] == [(p.token_index, p.time_index) for p in ref]
    assert np.allclose([p.score for p in path], [p.score for p in ref], rtol=1e-6)
    ti = np.array([p.token_index for p in path])
    tt = np.array([p.time_index for p in path])
    assert ti[0] == 0 and ti[-1] == L - 1 and set(np.diff(ti).tolist()) <= {0, 1} and (np.diff(tt) == 1).all()
    assert tt[-1] + 1 == int(np.argmax(tr[:, L]))
    segs = align.merge_repeats("".join(chr(97 + t % 26) for t in tok), path)
    assert len(segs) == L and all(a.end == b.start for a, b in zip(segs, segs[1:]))


def test_align_rejects_bad_tokens(align):
    em = torch.log_softmax(torch.randn(20, 8), -1).cuda()
    with pytest.raises(IndexError):
        align.forced_align(em, [1, 9, 2], 0)
    with pytest.raises(IndexError):
        align.forced_align(em, [], 0)


# ------------------------------------------------------------------------------------------------ batched launch + the tool
def test_batched_alignment_equals_single_launches():
    """ssak_ctc_forced_align_batch: ragged utterances (frames 20..700, transcripts 1..180 tokens, one infeasible) in ONE launch
    give, utterance by utterance, exactly the trellis, path and scores of the single-utterance entry -- both variants of
    column 0."""
    import ssak_amd.align as A
    rng = np.random.default_rng(11)
    V = 40
    shapes = [(700, 180), (20, 3), (333, 1), (499, 120), (64, 64), (30, 31)]  # the last: more tokens than frames -> no path
    ems = [torch.log_softmax(torch.tensor(rng.standard_normal((F, V)).astype(np.float32) * 2), dim=-1) for F, _ in shapes]
    toks = [list(rng.integers(1, V, L)) for _, L in shapes]
    for garbage in (False, True):
        batch = A.forced_align_batch(ems, toks, blank_id=0, first_as_garbage=garbage, want_trellis=True)
        for (F, L), em, tk, (tr_b, path_b) in zip(shapes, ems, toks, batch):
            tr_1, path_1 = A.forced_align(em, tk, 0, garbage)
            assert tr_b.shape == (F + 1, L + 1) and torch.equal(tr_b.cpu(), tr_1.cpu())
            assert (path_b is None) == (path_1 is None)
            if path_1 is not None:
                assert [(p.token_index, p.time_index, p.score) for p in path_b] == [(p.token_index, p.time_index, p.score) for p in path_1]
        assert batch[-1][1] is None
        without = A.forced_align_batch(ems, toks, blank_id=0, first_as_garbage=garbage, want_trellis=False)
        for (_, pa), (_, pb) in zip(batch, without):
            assert (pa is None and pb is None) or [(p.token_index, p.time_index, p.score) for p in pa] == [(p.token_index, p.time_index, p.score) for p in pb]


@pytest.mark.parametrize("refine", [None, 0.3])
def test_split_long_audio_kaldifolder_cut_points_vs_oracle(tmp_path, refine):
    """The consumer of the alignment (tools/align_audio_transcript.py:121-335 in the reference): a synthetic Kaldi folder -- two
    recordings, one with `segments` entries, utterances of 2..11 s, special words, an isolated punctuation mark, an empty line --
    is split at max_duration 4 s by ssak_amd.tools.align_audio_transcript (ONE alignment launch per batch of utterances).  The
    expected files come from oracle/align_tool_ref.py, which walks the utterances one at a time as the reference does, on the
    SAME emissions: text / utt2spk / utt2dur / segments must be identical line for line (cut points to the millisecond as
    printed), copied utterances included, in both the plain and the --refine_timestamps mode."""
    from oracle import align_tool_ref as OT
    from oracle import w2v2_ref as R
    from ssak_amd import align as A
    from ssak_amd import data as D
    from ssak_amd.checkpoint import save_pretrained
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.infer import compute_log_probas, transformers_load_model
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.synth import VOCAB, synth_wave
    from ssak_amd.tools import align_audio_transcript as T
    import dataclasses
    rng = np.random.default_rng(5)
    kd = tmp_path / "in"
    (kd / "audio").mkdir(parents=True)
    recs = {"recA": 26.0, "recB": 11.0}
    for name, secs in recs.items():
        D.write_wav(str(kd / "audio" / f"{name}.wav"), synth_wave(rng, int(secs * 16000)))
    def words(n):  # n random words of 2..6 letters
        return " ".join("".join(chr(97 + int(c)) for c in rng.integers(0, 26, int(rng.integers(2, 7)))) for _ in range(n))

    utts = [("recA-u1", "recA", 0.0, 11.0, words(22) + " , " + words(3)), ("recA-u2", "recA", 11.5, 14.0, words(4) + " <noise>"),
            ("recA-u3", "recA", 14.0, 23.5, "<laugh> " + words(18)), ("recA-u4", "recA", 23.5, 23.502, "<noise>"),
            ("recB-u1", "recB", 0.0, 11.0, words(20))]
    with open(kd / "wav.scp", "w") as fw:
        for name in recs:
            fw.write(f"{name} {kd}/audio/{name}.wav\n")
    with open(kd / "text", "w") as ft, open(kd / "utt2spk", "w") as fs, open(kd / "utt2dur", "w") as fd, open(kd / "segments", "w") as fg:
        for uid, rec, a, b, text in utts:
            ft.write(f"{uid} {text}\n")
            fs.write(f"{uid} spk_{rec}\n")
            fd.write(f"{uid} {b - a:.3f}\n")
            fg.write(f"{uid} {rec} {a} {b}\n")
    oc = R.W2V2Config.tiny().deterministic()
    d = dataclasses.asdict(oc)
    d.pop("initializer_range")
    base = Wav2Vec2ForCTC(Wav2Vec2Config(**d))
    base.load_state_dict(R.init_params(oc, 4))
    save_pretrained(base, D.CharTokenizer(VOCAB), str(tmp_path / "model"))
    del base
    out = tmp_path / "out"
    T.split_long_audio_kaldifolder(str(kd), str(out), str(tmp_path / "model"), min_duration=0.005, max_duration=4, refine_timestamps=refine,
                                   batch_size=2)
    # ---- the oracle's walk, utterance by utterance, on the same emissions
    model = transformers_load_model(str(tmp_path / "model"))
    labels, blank_id = A.get_model_vocab(model)
    want = {k: [] for k in ("text", "utt2spk", "utt2dur", "segments")}
    n_cut = 0
    for uid, rec, a, b, text in utts:
        norm = T.custom_text_normalization(text, lang="fr")
        if not norm or b - a <= 0.005:
            continue
        if b - a <= 4 and not refine:
            want["text"].append(f"{uid} {norm}\n")
            want["utt2spk"].append(f"{uid} spk_{rec}\n")
            want["utt2dur"].append(f"{uid} {float(f'{b - a:.3f}')}\n")
            want["segments"].append(f"{uid} {rec} {a} {b}\n")
            continue
        words = []
        for w in norm.split():
            if words and all(c in " " + OT.PUNCTUATION for c in w):
                words[-1] += " " + w
            else:
                words.append(w)
        spoken = [T.custom_word_normalization(w, lang="fr", **T.labels_to_norm_args(labels)) for w in words]
        s0, s1 = (max(0, a - refine), b + refine) if refine else (a, b)
        audio = D.load_audio(f"{kd}/audio/{rec}.wav", s0, s1, 16000)
        em = compute_log_probas(model, audio).numpy()
        nf, _, wsegs = OT.word_segments_from_words(em, spoken, labels, blank_id, first_as_garbage=bool(refine))
        lines = OT.cut_lines(uid, rec, f"spk_{rec}", s0, words, wsegs, nf, len(audio), 16000, 4, refine)
        n_cut += len(lines["text"])
        for k in want:
            want[k] += lines[k]
    assert n_cut >= (4 if refine else 6)  # the long utterances really were cut (with a garbage column an untrained model leaves few frames per word)
    for k in want:
        got = open(out / k).read().splitlines(keepends=True)
        assert got == sorted(want[k], key=lambda l: l.split(" ", 1)[0]), k
    assert open(out / "wav.scp").read() == open(kd / "wav.scp").read()
    spk2utt = dict(l.split(" ", 1) for l in open(out / "spk2utt").read().splitlines())
    assert set(spk2utt) == {"spk_recA", "spk_recB"} and "recB-u1_cut01" in spk2utt["spk_recB"]
