"""CPU: host-side logic of the product (Kaldi ingest, tokenizer, schedules, batching) against the golden vectors and
the conventions of the reference (file:line in the docstrings of ssak_amd/data.py)."""
import os

import numpy as np
import pytest

from ssak_amd import data as D
from ssak_amd.model import compute_mask_indices, conv_out_lengths
from ssak_amd.config import Wav2Vec2Config
from ssak_amd.trainer import linear_warmup_lr
from ssak_amd.train import word_error_rate, build_parser
from ssak_amd.synth import synth_batch, VOCAB


@pytest.fixture()
def kaldi(tmp_path, monkeypatch):
    rng = np.random.default_rng(0)
    audio = tmp_path / "audio"
    audio.mkdir()
    for name, n in (("a.wav", 16000), ("b c.wav", 24000), ("long.wav", 80000)):
        D.write_wav(str(audio / name), (rng.standard_normal(n) * 0.1).astype(np.float32))
    monkeypatch.setenv("DATAPATH", str(tmp_path))
    k1 = tmp_path / "k1"
    k1.mkdir()
    (k1 / "wav.scp").write_text("utt_a sox $DATAPATH/audio/a.wav -t wav -r 16k -b 16 -c 1 - |\n"
                                "utt_b sox '$DATAPATH/audio/b c.wav' -t wav -r 16k -b 16 -c 1 - |\n")
    (k1 / "text").write_text("utt_a bonjour <noise> le monde\nutt_b allô\n")
    (k1 / "utt2dur").write_text("utt_a 1.0\nutt_b 1.5\n")
    k2 = tmp_path / "k2"
    k2.mkdir()
    (k2 / "wav.scp").write_text(f"rec {audio}/long.wav\n")
    (k2 / "text").write_text("rec-s1 un deux\nrec-s2 trois\nrec-s3\n")
    (k2 / "segments").write_text("rec-s1 rec 0.50 1.50\nrec-s2 rec 2.00 4.25\nrec-s3 rec 4.5 4.6\n")
    (tmp_path / "list.txt").write_text("$DATAPATH/k1    1\n$DATAPATH/k2 2\n")
    return tmp_path


def test_wavscp_forms(kaldi):
    w = D.parse_kaldi_wavscp(str(kaldi / "k1" / "wav.scp"))
    assert w["utt_a"] == str(kaldi / "audio" / "a.wav") and w["utt_b"] == str(kaldi / "audio" / "b c.wav")
    (kaldi / "bad.scp").write_text("x ffmpeg -i foo.mp3 - |\n")
    with pytest.raises(RuntimeError):
        D.parse_kaldi_wavscp(str(kaldi / "bad.scp"))
    (kaldi / "flac.scp").write_text("x flac -c -d -s -f /tmp/foo.flac |\n")
    assert D.parse_kaldi_wavscp(str(kaldi / "flac.scp"))["x"] == "/tmp/foo.flac"
    # any whitespace separates the fields, as line.split() does in ssak/utils/kaldi.py:13 (tab-separated wav.scp files exist)
    (kaldi / "tabs.scp").write_text("utt1\t/data/a.wav\nutt2\tsox /d/a.wav -t wav - |\nutt3   /data/c.wav  \n\nutt4 \t sox\t'/d/x y.wav' -t wav - |\n")
    assert D.parse_kaldi_wavscp(str(kaldi / "tabs.scp")) == {"utt1": "/data/a.wav", "utt2": "/d/a.wav", "utt3": "/data/c.wav", "utt4": "/d/x y.wav"}
    (kaldi / "empty.scp").write_text("utt1\n")
    with pytest.raises(RuntimeError):
        D.parse_kaldi_wavscp(str(kaldi / "empty.scp"))


def test_kaldi_folder_semantics(kaldi):
    u = D.load_kaldi(str(kaldi / "k1"))
    assert [x.id for x in u] == ["utt_a", "utt_b"] and u[1].end == 1.5 and u[0].start == 0.0
    u = D.load_kaldi(str(kaldi / "k2"), min_duration=0.5, max_duration=15)
    assert [x.id for x in u] == ["rec-s1", "rec-s2"] and u[1].text == "trois"       # too-short segment dropped
    assert D.load_kaldi(str(kaldi / "k2"))[2].text == ""                               # empty transcript kept
    u = D.load_kaldi(str(kaldi / "list.txt"))                                          # list file, weight 2 duplicates
    assert len(u) == 2 + 2 * 3
    u = D.load_kaldi(f"{kaldi}/k1,{kaldi}/k2")
    assert len(u) == 5
    with pytest.raises(RuntimeError):
        D.load_kaldi(str(kaldi / "nope"))
    os.remove(kaldi / "k1" / "utt2dur")
    with pytest.raises(RuntimeError):
        D.load_kaldi(str(kaldi / "k1"))
    u = D.load_kaldi(str(kaldi / "k2"), sort_by_len=-1)
    assert u[0].id == "rec-s2"


def test_audio_segments_and_errors(kaldi):
    full = D.load_audio(str(kaldi / "audio" / "long.wav"))
    seg = D.load_audio(str(kaldi / "audio" / "long.wav"), 2.0, 4.25)
    assert len(full) == 80000 and len(seg) == int(4.25 * 16000) - int(2.0 * 16000)
    assert np.array_equal(seg, full[32000:68000]) and seg.dtype == np.float32
    with pytest.raises(RuntimeError):
        D.load_audio(str(kaldi / "audio" / "missing.wav"))
    (kaldi / "audio" / "fake.wav").write_bytes(b"ID3 not a wav")
    with pytest.raises(RuntimeError):
        D.load_audio(str(kaldi / "audio" / "fake.wav"))
    batches = list(D.to_audio_batches([str(kaldi / "k2")], batch_size=2, output_ids=True))
    assert [len(b) for b in batches] == [2, 1] and batches[0][0][1] == "rec-s1"


def test_label_cleanup_and_tokenizer(gold):
    assert D.remove_special_words("bonjour <noise>  le   monde") == "bonjour le monde"
    assert D.remove_special_words("j ' ai") == "j'ai"
    assert D.remove_special_words("j'ai", glue_apostrophe=False) == "j' ai"
    assert D.remove_special_words(None) == ""
    z = gold("greedy.npz")
    tok = D.CharTokenizer([str(v) for v in z["vocab"]])
    assert tok.pad_token_id == 0
    for ids, want in zip(z["ids"], z["text"]):  # Wav2Vec2CTCTokenizer.batch_decode golden
        assert tok.decode(ids) == " ".join(str(want).split())
    assert tok.encode("ab c") == [5, 6, 4, 7] and tok.decode(tok.encode("ab c"), group_tokens=False) == "ab c"


def test_collation_conventions(gold):
    z = gold("features.npz")
    assert (D.pad_labels([[5, 6, 7, 8], [9], [10, 11]]) == z["labels_padded"]).all()
    x, lens = D.pad_waves([np.ones(5, np.float32), np.ones(3, np.float32)])
    assert x.shape == (2, 5) and (x[1, 3:] == 0).all() and list(lens) == [5, 3]
    assert (conv_out_lengths(Wav2Vec2Config(), z["len_T"]) == z["len_F"]).all()


def test_specaugment_indices_match_hf(gold):
    z = gold("specaug.npz")
    for i, (B, S) in enumerate(((4, 499), (3, 499), (2, 49))):
        m = compute_mask_indices((B, S), 0.05, 10, None if i == 0 else list(z[f"lens{i}"]), 2, rng=np.random.RandomState(100 + i))
        assert (m == z[f"mask{i}"]).all()


def test_schedule_wer_naming():
    # lr logged after steps 1 and 2 in the reference's own golden (tests/expected/train_transformers/trainer_state.json:12-13,27-28)
    assert abs(linear_warmup_lr(1e-4, 1, 500, 10000) - 2e-7) < 1e-15
    assert abs(linear_warmup_lr(1e-4, 2, 500, 10000) - 4e-7) < 1e-15
    assert linear_warmup_lr(1e-4, 500, 500, 1000) == 1e-4 and linear_warmup_lr(1e-4, 750, 500, 1000) == 0.5e-4
    assert word_error_rate(["a b c", "d"], ["a x c", "d e"]) == 2 / 4


def test_output_folder_names_equal_the_reference_tests_golden_strings(gold_json, tmp_path):
    """The reference's own test asserts the folder names its train script creates
    (tests/unittests/test_train_transformers.py:23-24,55-56).  Same command line through this build's parser and
    ``train_folder_name`` (wav2vec_train.py:210-239) must give the same strings -- md5 of the data paths included."""
    from ssak_amd.naming import train_folder_name
    for case in gold_json("host_strings.json")["output_dir"]:
        root = tmp_path / "ref"
        for rel in (case["script"], case["train"], case["valid"]):
            (root / rel).parent.mkdir(parents=True, exist_ok=True)
        argv = [str(root / case["train"]), str(root / case["valid"]), "--base_model", case["base_model"]] + \
            [str(root / f) if f.startswith("tests/") else f for f in case["flags"]]
        a = build_parser().parse_args(argv)
        a.online = a.online or a.data_augment
        assert train_folder_name(vars(a), str(root / case["script"]), untrained=True) == case["untrained"]
        assert train_folder_name(vars(a), str(root / case["script"])) == case["output"]
    a = build_parser().parse_args(["tr", "va", "--base_model", "/x/m", "--debug", "--no_freeze"])
    assert train_folder_name(vars(a), "/s.py") == "hf_DEBUG_md-15_md-1_bm-_x_m_lr-0.0001_bs-8_wd-0_ad-0.1_hd-0.05_fpd-0_ld-0.1_mtp-0.05_s-69_adamwt_nofreeze"


def test_host_string_helpers_equal_the_reference(gold_json):
    """remove_special_words / hashmd5 / remove_commonprefix / args_to_str against strings produced by importing the reference's
    modules (oracle/gen_golden_host.py; ssak/utils/text_basic.py:91-125, misc.py:42-46,76-92, train_utils.py:4-38)."""
    from ssak_amd import naming as N
    z = gold_json("host_strings.json")
    for text, kw, want in z["remove_special_words"]:
        assert D.remove_special_words(text, **kw) == want, (text, kw)
    for obj, want in z["hashmd5_tuple"]:
        assert N.hashmd5(tuple(obj)) == want
    for paths, stop, want in z["remove_commonprefix"]:
        assert N.strip_common_prefix(paths, stop) == want
    for d, want in z["args_to_str"]:
        assert N.hparams_to_str(d) == want
    for d, want in z["args_to_str_sorted"]:
        assert N.hparams_to_str(d, sort=True) == want


def test_wavscp_lines_of_the_reference_fixtures_resolve(gold_json, tmp_path, monkeypatch):
    """Every wav.scp line of the reference's test Kaldi folders (tests/data/kaldi/{minimal,small,complete}: `id sox
    $DATAPATH/... |`, one quoted path with a space) and the other forms ssak/utils/kaldi.py:13-35 accepts (plain path, flac
    pipe, absolute sox) resolve to the expected audio path, with $VAR expansion; the weighted list file expands to
    (folder, weight) pairs (ssak/utils/dataset.py:165-192)."""
    z = gold_json("host_strings.json")
    monkeypatch.setenv("DATAPATH", "/corpus")
    for name, case in z["wavscp"].items():
        f = tmp_path / f"{name}.scp"
        f.write_text("\n".join(case["lines"]) + "\n")
        got = D.parse_kaldi_wavscp(str(f))
        assert got == {k: v.replace("$DATAPATH", "/corpus") for k, v in case["expected"].items()}, name
    (tmp_path / "bad.scp").write_text("x ffmpeg -i a.mp3 - |\n")
    with pytest.raises(RuntimeError):
        D.parse_kaldi_wavscp(str(tmp_path / "bad.scp"))
    for sub in ("small", "minimal"):
        (tmp_path / "kaldi" / sub).mkdir(parents=True)
    monkeypatch.setenv("DATAPATH", str(tmp_path))
    (tmp_path / "list.txt").write_text(z["list_file"])
    assert D.expand_kaldi_paths(str(tmp_path / "list.txt")) == [(str(tmp_path / "kaldi/small"), 1.0), (str(tmp_path / "kaldi/minimal"), 2.0)]


def test_batching_and_sharding():
    rng = np.random.RandomState(0)
    lens = list(rng.randint(16000, 240000, 103))
    b = D.length_grouped_batches(lens, 8, np.random.RandomState(1))
    assert sorted(i for x in b for i in x) == list(range(103))
    assert all(lens[x[0]] >= lens[x[-1]] for x in b)
    shards = [D.shard_batch(b[0], r, 4) for r in range(4)]
    assert sum(shards, []) == b[0] and all(len(s) == 2 for s in shards)


def test_synthetic_batch_is_ctc_feasible():
    w, lab = synth_batch(4, 16000, seed=1)
    assert w.shape == (4, 16000) and w.dtype == np.float32 and np.abs(w).max() <= 1.0
    n = (lab >= 0).sum(-1)
    assert n.min() >= 60 and n.max() <= 120 and lab.max() < len(VOCAB) and (2 * n + 1 <= 499).all()


def test_align_host_mirror_matches_oracle(gold):
    """The pure-Python tail of the alignment path (merge_repeats / merge_words / vocabulary helpers) against the oracle's
    restatement of ssak/utils/align_transcriptions.py:140-172 on the golden path."""
    from oracle import align_ref
    from ssak_amd import align
    from ssak_amd.data import CharTokenizer
    z = gold("align.npz")
    tok = z["base_tokens"].tolist()
    path_o = [align_ref.Point(int(a), int(b), float(c)) for a, b, c in zip(z["base_path_token"], z["base_path_time"], z["base_path_score"])]
    path_m = [align.Point(p.token_index, p.time_index, p.score) for p in path_o]
    transcript = "".join("ab cd"[t % 5] for t in tok)
    so, sm = align_ref.merge_repeats(transcript, path_o), align.merge_repeats(transcript, path_m)
    assert [(s.label, s.start, s.end, s.score) for s in so] == [(s.label, s.start, s.end, s.score) for s in sm]
    wo, wm = align_ref.merge_words(so), align.merge_words(sm)
    assert [(s.label, s.start, s.end, s.score) for s in wo] == [(s.label, s.start, s.end, s.score) for s in wm]
    t = CharTokenizer(["<pad>", "<s>", "</s>", "<unk>", "|", "a", "b", "É"])
    labels, blank = align.get_model_vocab((None, t))
    assert labels[4] == " " and blank == 0
    d = {c: i for i, c in enumerate(labels)}
    assert align.loose_get_char_index(d, "A", 4) == 5 and align.loose_get_char_index(d, "é", 4) == 7
    assert align.loose_get_char_index(d, "z", 4) == 4 and align.loose_get_char_index(d, "z", None) is None


def test_newbob_scheduler_mirror():
    """ssak_amd.sb_head.NewBobScheduler (the recipe's lr_annealing_* objects, yaml :124-135) against the oracle's list form."""
    from oracle import sb_head_ref as S
    from ssak_amd.sb_head import NewBobScheduler
    vals = [12.0, 11.0, 10.99, 10.98, 9.0, 9.5, 9.4, 0.0, 0.0]
    for factor, patient in ((0.8, 0), (0.9, 0), (0.5, 2)):
        sch = NewBobScheduler(1.0, factor, 0.0025, patient)
        got = []
        for v in vals:
            old, new = sch(v)
            assert old == (got[-1] if got else 1.0)
            got.append(new)
        assert got == pytest.approx(S.new_bob(vals, 1.0, factor, 0.0025, patient))
    sch2 = NewBobScheduler(1.0, 0.8)
    sch2.load_state_dict(sch.state_dict())
    assert sch2.hyperparam_value == sch.hyperparam_value and sch2.metric_values == sch.metric_values


_SB_YAML = """lr: 1.0
lr_wav2vec: 0.0001
batch_size: 32
freeze_wav2vec: True
seed: 1234
__set_seed: !apply:torch.manual_seed [!ref <seed>]
train: !PLACEHOLDER
valid: !PLACEHOLDER
output_folder_prefix: ''
output_folder: !ref <output_folder_prefix>run_lr<lr>_bs<batch_size>
save_folder: !ref <output_folder>/save
model_opt_class: !name:torch.optim.Adadelta
    lr: !ref <lr>
    rho: 0.95
    eps: 1.e-8
lr_annealing_model: !new:speechbrain.nnet.schedulers.NewBobScheduler
    initial_value: !ref <lr>
    annealing_factor: 0.8
    patient: 0
"""


def test_speechbrain_cli_yaml_and_overrides(tmp_path):
    """ssak_amd.train_speechbrain reads the recipe's yaml without hyperpyyaml: tagged nodes as plain mappings, `!ref <key>`
    resolved (typed when alone, interpolated inside strings), `!PLACEHOLDER` filled from --key=value / --key value, types of
    overridden entries kept, --gpus dropped (ssak/train/speechbrain/wav2vec_train.py:499-529)."""
    from ssak_amd.train_speechbrain import CharVocab, load_hparams, pad_tokens, parse_argv
    y = tmp_path / "h.yaml"
    y.write_text(_SB_YAML)
    f, o = parse_argv([str(y), "--train=/a", "--valid", "/b", "--lr=0.5", "--freeze_wav2vec=False", "--gpus", "0", "--debug"])
    assert f == str(y) and o == {"train": "/a", "valid": "/b", "lr": "0.5", "freeze_wav2vec": "False", "debug": "true"}
    hp = load_hparams(f, o)
    assert hp["lr"] == 0.5 and hp["freeze_wav2vec"] is False and hp["batch_size"] == 32 and hp["train"] == "/a"
    assert hp["output_folder"] == "run_lr0.5_bs32" and hp["save_folder"] == "run_lr0.5_bs32/save"
    assert hp["model_opt_class"] == {"lr": 0.5, "rho": 0.95, "eps": 1e-8}
    assert hp["lr_annealing_model"]["initial_value"] == 0.5 and hp["lr_annealing_model"]["annealing_factor"] == 0.8
    with pytest.raises(SystemExit):
        load_hparams(f, {})  # train / valid are mandatory
    with pytest.raises(SystemExit):
        parse_argv(["--train=/a"])  # no yaml file
    v = CharVocab.from_texts(["ab c", "ca"])
    assert v.symbols == ["<blank>", " ", "a", "b", "c"] and v.encode("a cz") == [2, 1, 4]
    t, tl = pad_tokens([[2, 3, 4, 2], [3]])
    assert t.tolist() == [[2, 3, 4, 2], [3, 0, 0, 0]] and tl.tolist() == [1.0, 0.25]


def test_shard_batch_keeps_every_utterance():
    """Uneven contiguous shards (the short last batch of an epoch is trained, HF dataloader_drop_last=False): the first
    len % world ranks take one more, nothing is dropped, a rank may be empty."""
    for n in range(0, 12):
        for world in (1, 2, 3, 8):
            idx = list(range(100, 100 + n))
            shards = [D.shard_batch(idx, r, world) for r in range(world)]
            assert sum(shards, []) == idx
            sizes = [len(s) for s in shards]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_best_model_tracker_and_checkpoint_rotation(tmp_path):
    """metric_for_best_model="wer" (lower is better) + EarlyStoppingCallback + save_total_limit=2 as HF Trainer sequences them
    (wav2vec_train.py:368-372,392; docker/transformers_modified/trainer.py:2224-2238,2735-2783): a tie is no improvement for
    the patience counter and does not move the best checkpoint; the best and the newest checkpoint survive the rotation."""
    import json
    from ssak_amd import train as T
    state = {}
    tr = T.BestModelTracker(state, patience=3)
    out = tmp_path / "run"
    wers = [0.9, 0.5, 0.5, 0.7, 0.4, 0.6, 0.6, 0.6]
    stops, kept = [], []
    for i, w in enumerate(wers, 1):
        ck = out / f"checkpoint-{10 * i}"
        ck.mkdir(parents=True)
        stops.append(tr.after_evaluation(w, str(ck)))
        (ck / "trainer_state.json").write_text(json.dumps(state))
        T.rotate_checkpoints(str(out), state["best_model_checkpoint"])
        kept.append(sorted(int(d.split("-")[1]) for d in os.listdir(out)))
    assert stops == [False] * 7 + [True]
    assert kept == [[10], [10, 20], [20, 30], [20, 40], [40, 50], [50, 60], [50, 70], [50, 80]]
    assert state["best_metric"] == 0.4 and state["best_model_checkpoint"].endswith("checkpoint-50")
    assert state["early_stopping_patience_counter"] == 3
    assert T.EARLY_STOPPING_PATIENCE == 15  # the reference's patience


def test_alignment_tool_text_normalisation_matches_the_reference(gold_json):
    """The transcript clean-up of ssak_amd.tools.align_audio_transcript (tools/align_audio_transcript.py:78-118 in the
    reference) against strings produced by IMPORTING the reference's ssak.utils.text_basic and evaluating its
    punctuation-spacing tables (oracle/gen_golden_host.py): typographic quotes, composed accents, ellipsis, special words,
    French / default spacing, ligatures and punctuation stripping at word level."""
    from ssak_amd.tools import align_audio_transcript as T
    z = gold_json("host_strings.json")
    assert len(z["align_text_normalization"]) >= 30
    for text, lang, want in z["align_text_normalization"]:
        assert T.custom_text_normalization(text, lang=lang) == want, (text, lang)
    for word, lig, punc, want in z["align_word_normalization"]:
        got = T.custom_word_normalization(word, "fr", remove_digits=False, remove_punc=punc, remove_ligatures=lig, remove_etset=False)
        assert got == " ".join(want.split()), (word, lig, punc)


def test_alignment_tool_cut_decision_pure_function():
    """cut_at_word_boundaries against oracle/align_tool_ref.cut_lines (the reference's add_segment loop restated) on random word
    timings: same pieces, same printed times, plain and refine modes, punctuation-only words, pieces longer than max_duration."""
    from oracle import align_ref as AR
    from oracle import align_tool_ref as OT
    from ssak_amd.align import Segment
    from ssak_amd.tools import align_audio_transcript as T
    rng = np.random.default_rng(3)
    for trial in range(60):
        n = int(rng.integers(1, 40))
        F = int(rng.integers(50, 1500))
        cuts = np.sort(rng.integers(0, F, 2 * n))
        words = ["w%d" % i if rng.random() > 0.1 else "," for i in range(n)]
        if words[0] == ",":
            words[0] = "w0"
        segs = [(w, int(cuts[2 * i]), int(cuts[2 * i + 1]) + 1, float(rng.random())) for i, w in enumerate(words)]
        audio_len = F * 320 + int(rng.integers(0, 320))
        for refine in (None, 0.25):
            maxd = float(rng.choice([0.5, 2.0, 4.0, 30.0]))
            got = T.cut_at_word_boundaries([Segment(*s) for s in segs], words, F, audio_len, 16000, maxd, refine)
            want = OT.cut_lines("u", "rec", "spk", 1.5, words, [AR.Segment(*s) for s in segs], F, audio_len, 16000, maxd, refine)
            lines = [f"u_cut{i:02} rec {1.5 + a:.3f} {1.5 + b:.3f}\n" for i, a, b, _ in got if b > a]
            assert lines == want["segments"], (trial, refine)
            assert [f"u_cut{i:02} {t}\n" for i, a, b, t in got if b > a] == want["text"]


def test_length_grouped_batches_with_a_frame_budget():
    """ssak_amd.data.length_grouped_batches(frame_budget=...): the reference's shuffle / mega-batch / sort (HF LengthGroupedSampler,
    docker/transformers_modified/trainer.py:758-775) with slices of a constant PADDED length instead of a constant count: every
    utterance exactly once, count x longest <= budget (or a single utterance longer than the budget), the count capped, the
    count mode untouched by the new argument."""
    import numpy as np
    from ssak_amd.data import length_grouped_batches
    rng = np.random.default_rng(0)
    L = (np.exp(rng.uniform(0, np.log(15), 1000)) * 16000).astype(int).tolist()
    ref = length_grouped_batches(L, 16, np.random.RandomState(3))
    assert ref == length_grouped_batches(L, 16, np.random.RandomState(3), frame_budget=None)
    assert all(len(b) == 16 for b in ref[:-1]) and sorted(i for b in ref for i in b) == list(range(1000))
    budget = 160 * 16000
    got = length_grouped_batches(L, 16, np.random.RandomState(3), frame_budget=budget)
    assert sorted(i for b in got for i in b) == list(range(1000))
    for b in got:
        assert L[b[0]] == max(L[i] for i in b)  # sorted inside a mega-batch: the first one is the longest
        assert len(b) * L[b[0]] <= budget or len(b) == 1
        assert 1 <= len(b) <= 8 * 16
    assert max(len(b) for b in got) > 3 * min(len(b) for b in got)  # short utterances travel in larger batches
    tight = length_grouped_batches(L, 16, np.random.RandomState(3), frame_budget=budget, max_batch=20)
    assert max(len(b) for b in tight) == 20
