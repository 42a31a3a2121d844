"""Secondary workloads of BASELINE.json (configs[3], configs[4]) and the audio-ingest path, as functions: bench.py puts their
lines into `secondary[]` of its one JSON line; tools/bench_whisper.py / bench_xlsr.py / bench_ingest.py print them alone.
Synthetic data and seeded random-init weights of the named architectures (no network for datasets or checkpoints)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0


def _w2v2_state(model, seed=69):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, (off, n, shape) in model.layout.items():
        if name.endswith("layer_norm.weight"):
            t = torch.ones(shape)
        elif name.endswith(".bias"):
            t = torch.zeros(shape)
        elif name.endswith("masked_spec_embed"):
            t = torch.rand(shape, generator=g)
        elif ".conv.weight" in name or name.endswith("original1"):
            t = torch.randn(shape, generator=g) * (2.0 / (shape[1] * shape[2])) ** 0.5
        else:
            t = torch.randn(shape, generator=g) * 0.02
        sd[name] = t
    v = sd["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    sd["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return sd


def whisper_step(B=8, steps=10, warmup=2, device="cuda:0"):
    """BASELINE configs[3]: Whisper-small encoder + CTC head train step on 30 s windows.  Step = log-mel (a13, HIP kernel) ->
    encoder forward -> CTC -> backward -> clip + AdamW.  Also times the log-mel front end alone (HIP events)."""
    from ssak_amd import hip
    from ssak_amd.synth import synth_wave
    from ssak_amd.trainer import AdamW
    from ssak_amd.whisper import WhisperCTCConfig, WhisperEncoderForCTC
    cfg = WhisperCTCConfig()
    model = WhisperEncoderForCTC(cfg, device=device).train()
    g = torch.Generator().manual_seed(0)
    sd = {}
    for n, (off, cnt, shape) in model.layout.items():
        if n.endswith("embed_positions.weight"):
            length, channels = shape  # Whisper's fixed sinusoidal table
            inv = torch.exp(-(np.log(10000.0) / (channels // 2 - 1)) * torch.arange(channels // 2))
            t = torch.arange(length).view(-1, 1) * inv.view(1, -1)
            sd[n] = torch.cat([t.sin(), t.cos()], dim=1)
        elif "layer_norm.weight" in n:
            sd[n] = torch.ones(shape)
        elif n.endswith(".bias"):
            sd[n] = torch.zeros(shape)
        else:
            sd[n] = torch.randn(shape, generator=g) * 0.02
    model.load_state_dict(sd)
    opt = AdamW(model, lr=1e-4, warmup_steps=500)
    rng = np.random.default_rng(0)
    wav = torch.tensor(np.stack([synth_wave(rng, 480000) for _ in range(B)])).to(device)
    labels = torch.randint(1, cfg.vocab_size, (B, 200)).to(device)

    def step():
        mel = hip.logmel_whisper(wav)
        out = model(mel, labels=labels)
        model.backward()
        opt.step()
        return out.loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # the log-mel front end alone, timed by the library's own launch-timing slot (HIP events on the launch stream, like every
    # kernel class of the headline's survey): 1.92 MB of waveform in, 0.96 MB of features out per window
    hip.logmel_whisper(wav)
    torch.cuda.synchronize()
    hip.prof_enable(1)
    hip.prof_collect()
    for _ in range(10):
        hip.logmel_whisper(wav)
    torch.cuda.synchronize()
    hip.prof_enable(0)
    lm_row = [r for r in hip.prof_collect() if "log-mel" in r[0] and r[1] > 0][0]
    lm_us = lm_row[2] * 1e3 / lm_row[1]
    lm_bytes = lm_row[3] / lm_row[1]
    gf = 1032.0  # ~3 x 344.16 GF per 30 s window, nothing frozen
    tf = gf * B / dt / 1e3
    del model, opt
    return {"workload": "Whisper-small encoder + CTC head train step incl. the log-mel kernel, bf16, synthetic 30 s windows "
                        "(BASELINE.json configs[3], 1 GPU)",
            "value": round(B / dt, 2), "unit": "windows/sec", "audio_sec_per_sec": round(30 * B / dt, 1), "ms_per_step": round(dt * 1e3, 3),
            "batch": B, "steps": steps, "whole_step_tflops": round(tf, 1), "whole_step_frac": round(tf / PEAK_BF16_TFLOPS, 4),
            "logmel": {"kernel": lm_row[0], "bound": "hbm", "launches": lm_row[1], "us_per_window": round(lm_us / B, 2),
                       "us_per_call": round(lm_us, 1), "algorithmic_mb_per_window": round(lm_bytes / B / 1e6, 3),
                       "achieved_gbs": round(lm_bytes / (lm_us * 1e-6) / 1e9, 1), "peak_gbs": 8000.0,
                       "frac": round(lm_bytes / (lm_us * 1e-6) / 1e9 / 8000.0, 4), "share_of_step": round(lm_us * 1e-6 / dt, 4),
                       "timed_with": "ssak_prof_* slot: HIP events around every ssak_logmel_whisper call on the launch stream"},
            "loss": round(float(loss.item()), 4)}


def whisper_sweep(batches=(8, 16, 32), steps=10, device="cuda:0"):
    """"B in {8, 16, 32}; report best" for the Whisper window step -- the best line, with every batch's figures
    under `sweep` (the log-mel slot is the best line's)."""
    rows = []
    for B in batches:
        try:
            rows.append(whisper_step(B=B, steps=steps if B <= 16 else max(4, steps // 2), device=device))
        except Exception as e:  # (a batch that does not fit must not take the line down)
            rows.append({"batch": B, "error": repr(e)})
        torch.cuda.empty_cache()
    ok = [r for r in rows if "value" in r]
    best = dict(max(ok, key=lambda r: r["value"]))
    best["sweep"] = [{k: r.get(k) for k in ("batch", "value", "ms_per_step", "whole_step_frac", "error") if k in r} for r in rows]
    return best


def xlsr_bucketed(B=16, N=320, device="cuda:0", frame_budget=None, model_trainer=None):
    """BASELINE configs[4] on one GPU: Wav2Vec2-large-XLSR CTC train step over mixed-length utterances (durations log-uniform in
    [1 s, 15 s], wav2vec_train.py:149-150), batched like HF's LengthGroupedSampler (docker/transformers_modified/trainer.py:758-775),
    right-padded with lengths.  One pass over all batches; algorithmic FLOPs counted on the real (unpadded) lengths."""
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.data import length_grouped_batches
    from ssak_amd.model import Wav2Vec2ForCTC, conv_out_lengths
    from ssak_amd.synth import synth_text, synth_wave, text_to_ids
    from ssak_amd.trainer import AdamW, Trainer
    cfg = Wav2Vec2Config(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                         feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)

    def train_gflop(T: int) -> float:
        L, cin, fe = T, 1, 0.0
        for c, k, s in zip(cfg.conv_dim, cfg.conv_kernel, cfg.conv_stride):
            L = (L - k) // s + 1
            fe += 2.0 * L * c * cin * k
            cin = c
        H, I, F = cfg.hidden_size, cfg.intermediate_size, L
        rest = 2.0 * F * cin * H + 2.0 * F * H * (H // cfg.num_conv_pos_embedding_groups) * cfg.num_conv_pos_embeddings
        rest += cfg.num_hidden_layers * (2.0 * F * (4 * H * H + 2 * H * I) + 4.0 * F * F * H) + 2.0 * F * H * cfg.vocab_size
        return (fe + 3.0 * rest) / 1e9

    assert abs(train_gflop(160000) - 1053.50) < 1.0, train_gflop(160000)  # XLSR-large @10 s
    if model_trainer is None:
        model = Wav2Vec2ForCTC(cfg, device=device, freeze_feature_encoder=True, seed=69).train()
        model.load_state_dict(_w2v2_state(model))
        opt = AdamW(model, lr=1e-4, warmup_steps=500)
        trainer = Trainer(model, opt)
    else:
        model, trainer = model_trainer
    rng = np.random.default_rng(1234)
    durs = np.exp(rng.uniform(np.log(1.0), np.log(15.0), N))
    nsamp = (durs * 16000).astype(np.int64)
    if frame_budget is None:
        batches = [b for b in length_grouped_batches(nsamp.tolist(), B, np.random.RandomState(0)) if len(b) == B]
    else:  # constant padded length per step (ssak_amd.data.length_grouped_batches: frame_budget, in samples here)
        batches = length_grouped_batches(nsamp.tolist(), B, np.random.RandomState(0), frame_budget=frame_budget)
    dev_batches = []
    for idx in batches:
        T = (int(max(nsamp[i] for i in idx)) + 7) // 8 * 8
        wav = np.zeros((len(idx), T), np.float32)
        ids = []
        for r, i in enumerate(idx):
            wav[r, :nsamp[i]] = synth_wave(rng, int(nsamp[i]))
            fl = int(conv_out_lengths(cfg, np.array([nsamp[i]]))[0])
            n = max(1, min(int(durs[i] * 8), (fl - 1) // 2))  # ~8 characters per second, always feasible for CTC
            ids.append(text_to_ids(synth_text(rng, n, n)))
        Lm = max(len(x) for x in ids)
        lab = np.full((len(idx), Lm), -100, np.int64)
        for r, x in enumerate(ids):
            lab[r, :len(x)] = x
        dev_batches.append((torch.tensor(wav).to(device), torch.tensor(nsamp[idx].astype(np.int32)).to(device), torch.tensor(lab).to(device)))
    order = sorted(range(len(dev_batches)), key=lambda j: -dev_batches[j][0].shape[1])
    for j in (order[0], order[-1]):  # warm-up on the longest batch (sizes the workspace once) and one short one
        trainer.train_step(*dev_batches[j])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for w, l, y in dev_batches:
        loss = trainer.train_step(w, l, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    used = [i for b in batches for i in b]
    utts = len(used)
    audio = float(sum(durs[i] for i in used))
    padded = float(sum(dev_batches[j][0].shape[1] * dev_batches[j][0].shape[0] for j in range(len(dev_batches)))) / 16000.0
    tf = sum(train_gflop(int(nsamp[i])) for i in used) / dt / 1e3
    return {"workload": "Wav2Vec2-large-XLSR CTC train step, bf16, durations log-uniform 1-15 s, length-grouped batches, attention "
                        "mask (BASELINE.json configs[4] on 1 GPU)",
            "value": round(utts / dt, 2), "unit": "utterances/sec", "audio_sec_per_sec": round(audio / dt, 1),
            "batch": B if frame_budget is None else None,
            "batching": (f"constant count: {B} utterances per step (HF LengthGroupedSampler, the reference's mode)" if frame_budget is None else
                         f"constant padded length: count x longest <= {frame_budget / 16000:.0f} s of audio per step "
                         f"({min(len(b) for b in batches)}-{max(len(b) for b in batches)} utterances; ssak_amd.data.length_grouped_batches frame_budget)"),
            "steps": len(dev_batches), "ms_per_step": round(dt / len(dev_batches) * 1e3, 3), "padding_overhead": round(padded / audio - 1.0, 4),
            "whole_step_tflops": round(tf, 1), "whole_step_frac": round(tf / PEAK_BF16_TFLOPS, 4), "loss": round(float(loss.item()), 4)}


def xlsr_sweep(batches=(16, 32, 64), budgets_s=(320.0,), N=640, device="cuda:0"):
    """SURVEY.md section 8d for configs[4]: the bucketed epoch at B in {16, 32, 64} (constant count, the reference's batching) and
    with a constant padded length per step; the best line, every configuration's figures under `sweep`.  One model for all."""
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer
    cfg = Wav2Vec2Config(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                         feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)
    model = Wav2Vec2ForCTC(cfg, device=device, freeze_feature_encoder=True, seed=69).train()
    model.load_state_dict(_w2v2_state(model))
    trainer = Trainer(model, AdamW(model, lr=1e-4, warmup_steps=500))
    rows = []
    for B in batches:
        rows.append(xlsr_bucketed(B=B, N=N, device=device, model_trainer=(model, trainer)))
    for sec in budgets_s:
        rows.append(xlsr_bucketed(B=16, N=N, device=device, frame_budget=sec * 16000, model_trainer=(model, trainer)))
    best = dict(max(rows, key=lambda r: r["audio_sec_per_sec"]))
    best["sweep"] = [{k: r[k] for k in ("batching", "value", "audio_sec_per_sec", "ms_per_step", "steps", "padding_overhead", "whole_step_frac")} for r in rows]
    del model, trainer
    return best


def online_steps(B=32, steps=20, warmup=5, n_files=4096, device="cuda:0", keep_dir=None):
    """The headline step fed from FILES: SURVEY.md section 8d's Kaldi folder (4 096 PCM16 mono 16 kHz WAV files of 10 s, wav.scp / text
    / utt2dur) through the loop of `python -m ssak_amd.train --online` (ssak_amd/train.py: load_kaldi -> length-grouped batch plan
    -> BatchPrefetcher(DeviceIngest) -> Trainer.train_step on the normalised device batch), `steps` timed steps after `warmup`.
    Reference path: ssak/utils/dataset.py:498-645, ssak/utils/audio.py:24-105, 6 loader workers at wav2vec_train.py:360."""
    import shutil
    import tempfile
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.data import CharTokenizer, length_grouped_batches, load_kaldi, pad_labels, remove_special_words
    from ssak_amd.ingest import BatchPrefetcher, DeviceIngest
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.synth import VOCAB, write_kaldi_folder
    from ssak_amd.trainer import AdamW, Trainer
    d = keep_dir or tempfile.mkdtemp(prefix="ssak_kaldi_")
    t_gen = time.perf_counter()
    if not os.path.exists(os.path.join(d, "wav.scp")):
        write_kaldi_folder(d, n_files, threads=min(16, len(os.sched_getaffinity(0))))
    t_gen = time.perf_counter() - t_gen
    utts = load_kaldi(d, 1.0, 15.0)
    tok = CharTokenizer(VOCAB)
    labels = [tok.encode(remove_special_words(u.text)) for u in utts]
    lens = [int(u.duration * 16000) for u in utts]
    plan = length_grouped_batches(lens, B, np.random.RandomState(69))[:warmup + steps]
    model = Wav2Vec2ForCTC(Wav2Vec2Config(), device=device, freeze_feature_encoder=True, seed=69).train()
    model.load_state_dict(_w2v2_state(model))
    trainer = Trainer(model, AdamW(model, lr=1e-4, weight_decay=0.0, max_grad_norm=1.0, warmup_steps=500, total_steps=100000))
    ingest = DeviceIngest(16000, device)
    feed = iter(BatchPrefetcher(ingest, [[(utts[i].path, utts[i].start or None, utts[i].end or None) for i in m] for m in plan], depth=3,
                                labels=[pad_labels([labels[i] for i in m]) for m in plan]))
    t0 = None
    for k, m in enumerate(plan):
        if k == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        x, ln, lab = next(feed)
        loss = trainer.train_step(x, ln, lab, raw=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = sum(len(m) for m in plan[warmup:])
    if keep_dir is None:
        shutil.rmtree(d, ignore_errors=True)
    del model, trainer
    return {"path": f"Kaldi folder of {n_files} x 10 s PCM16 WAV files -> load_kaldi -> length-grouped batches of {B} -> BatchPrefetcher "
                    f"({ingest.readers} native reader threads (ssak_read_ranges: pread into a pinned ring), one H2D copy per batch) -> device decode + normalise -> "
                    "the headline train step (the loop of `ssak_amd.train --online`)",
            "value": round(n / dt, 2), "unit": "utterances/sec", "steps": len(plan) - warmup, "warmup": warmup, "ms_per_step": round(dt / (len(plan) - warmup) * 1e3, 3),
            "batch": B, "folder_written_in_s": round(t_gen, 1), "final_loss": round(float(loss.item()), 4)}


def ingest_rate(N=1024, sr=16000, nch=1, B=32):
    """10 s PCM16 WAV files -> normalised device batches of B through the prefetching device ingest
    (ssak_amd/ingest.py: file read, PCM decode + mono mix + resample on the device, waveform normalisation)."""
    import tempfile
    import wave as wavmod
    from ssak_amd.ingest import BatchPrefetcher, DeviceIngest
    d = tempfile.mkdtemp()
    rng = np.random.default_rng(0)
    items = []
    for i in range(N):
        p = os.path.join(d, f"u{i}.wav")
        pcm = (rng.standard_normal(10 * sr * nch) * 3000).astype("<i2")
        with wavmod.open(p, "wb") as f:
            f.setnchannels(nch)
            f.setsampwidth(2)
            f.setframerate(sr)
            f.writeframes(pcm.tobytes())
        items.append((p, None, None))
    batches = [items[i:i + B] for i in range(0, N, B)]
    ing = DeviceIngest(16000)
    readers = ing.readers
    for w, l in BatchPrefetcher(ing, batches[:2]):
        pass
    torch.cuda.synchronize()
    # cold pass first: the files were written a moment ago and sit in the page cache -- fdatasync + posix_fadvise(DONTNEED) on each
    # (ssak_drop_file_cache) sends the next read to the storage device.  (On a memory-backed file system the advice is a no-op and the
    # two figures coincide: `fs_type` says what /tmp is.)
    import ctypes
    from ssak_amd import hip
    from ssak_amd.ingest import clear_wav_cache
    c_paths = (ctypes.c_char_p * N)(*[os.fsencode(p) for p, _, _ in items])
    not_dropped = hip.lib.ssak_drop_file_cache(c_paths, N)
    clear_wav_cache()
    t0 = time.perf_counter()
    for w, l in BatchPrefetcher(ing, batches, depth=3):
        pass
    torch.cuda.synchronize()
    dt_cold = time.perf_counter() - t0
    fs_type = "?"
    try:
        best = ""
        for line in open("/proc/mounts"):
            f = line.split()
            if d.startswith(f[1]) and len(f[1]) > len(best):
                best, fs_type = f[1], f[2]
    except OSError:
        pass
    t0 = time.perf_counter()
    for w, l in BatchPrefetcher(ing, batches, depth=3):
        pass
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for p, _, _ in items:
        os.unlink(p)
    os.rmdir(d)
    return {"path": "Kaldi-style wav.scp entries -> PCM16 WAV read -> device decode / mono / resample -> ssak_wave_normalize -> batches of "
                    f"{B} (ssak_amd/ingest.py: {readers} native reader threads pread into a pinned ring (ssak_read_ranges), prefetch depth 3)",
            "utterances_per_sec": round(N / dt, 1), "files": N, "reader_threads": readers,
            "warm": {"utterances_per_sec": round(N / dt, 1), "mb_per_sec": round(N * 10 * sr * nch * 2 / dt / 1e6, 1), "what": "second pass: page-cache reads"},
            "cold": {"utterances_per_sec": round(N / dt_cold, 1), "mb_per_sec": round(N * 10 * sr * nch * 2 / dt_cold / 1e6, 1),
                     "what": "first pass after fdatasync + posix_fadvise(DONTNEED) on every file (ssak_drop_file_cache) and with the header cache "
                             "cleared: reads come from the storage device", "files_not_dropped": int(not_dropped), "fs_type": fs_type},
            "source": f"{sr} Hz x {nch} ch, 10 s"}


# ---------------------------------------------------------------------------------------------------------------------------------
# The ssak/infer entry point and the f-rows (forced alignment, evaluation metric) on the driver's clock: secondary[2..4].
GF_FORWARD_PER_UTT = 148.16  # Wav2Vec2-base forward, 10 s utterance


def _cpu_cores():
    sys.path.insert(0, ROOT)
    import bench
    return bench.host_cores()


def infer_line(B=32, steps=20, warmup=3, device="cuda:0", cpu_batch=4, cpu_steps=3):
    """The `ssak/infer` hot path (ssak/infer/transformers_infer.py:190-269: processor -> model(...).logits -> argmax / batch_decode,
    :84-85) at B x 10 s: ssak_wave_normalize -> Wav2Vec2-base eval forward -> ssak_ctc_greedy_decode, inputs resident in HBM.
    Beside it the CPU restatement's eval-mode forward (oracle/w2v2_ref.py, eager torch fp32 on all host cores) + argmax + collapse
    on a bounded sample."""
    from ssak_amd import hip
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.synth import synth_batch
    cfg = Wav2Vec2Config()
    model = Wav2Vec2ForCTC(cfg, device=device, seed=69).eval()
    model.load_state_dict(_w2v2_state(model))
    waves_np = synth_batch(B, 160000, seed=1)[0]
    waves = torch.tensor(waves_np).to(device)

    def step():
        x = hip.wave_normalize(waves, None)
        out = model(x)
        return hip.ctc_greedy_decode(out.logits.contiguous(), None, cfg.pad_token_id)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ids, n = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # latency of ONE utterance (the reference's `compute_logits(model, audio)` for a single file, ssak/infer/general.py:76-97)
    one = waves[:1].contiguous()
    for _ in range(3):
        hip.ctc_greedy_decode(model(hip.wave_normalize(one, None)).logits.contiguous(), None, cfg.pad_token_id)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        hip.ctc_greedy_decode(model(hip.wave_normalize(one, None)).logits.contiguous(), None, cfg.pad_token_id)
    torch.cuda.synchronize()
    lat = (time.perf_counter() - t1) / 10
    tf = GF_FORWARD_PER_UTT * B / dt / 1e3
    out = {"workload": "ssak/infer path: ssak_wave_normalize -> Wav2Vec2-base eval forward (bf16) -> ssak_ctc_greedy_decode, "
                       f"B = {B} x 10 s utterances resident in HBM (transformers_infer.py:190-269, :84-85)",
           "value": round(B / dt, 1), "unit": "utterances/sec", "batch": B, "steps": steps, "ms_per_batch": round(dt * 1e3, 3),
           "real_time_factor": round(dt / (10.0 * B), 7), "audio_sec_per_sec": round(10.0 * B / dt, 1),
           "single_utterance_latency_ms": round(lat * 1e3, 3),
           "whole_forward_tflops": round(tf, 1), "whole_forward_frac": round(tf / PEAK_BF16_TFLOPS, 4),
           "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4),
                        "algorithmic_gflop_per_utt": GF_FORWARD_PER_UTT}}
    del model
    torch.cuda.empty_cache()
    try:
        from oracle import w2v2_ref as R
        cores = _cpu_cores()
        torch.set_num_threads(cores)
        oc = R.W2V2Config.base()
        p = R.init_params(oc, 69)
        x = torch.tensor(R.zero_mean_unit_var_norm(list(waves_np[:cpu_batch])))
        ts = []
        with torch.no_grad():
            for k in range(1 + cpu_steps):
                t2 = time.perf_counter()
                _, logits = R.forward(p, oc, x, None, None, train=False)
                pred = logits.argmax(-1).numpy()
                for row in pred:  # greedy collapse: repeats merged, blanks dropped (batch_decode's group_tokens)
                    keep = np.concatenate([[True], row[1:] != row[:-1]]) & (row != oc.pad_token_id)
                    row[keep]
                ts.append(time.perf_counter() - t2)
        med = sorted(ts[1:])[len(ts[1:]) // 2]
        out["cpu_baseline"] = {"value": round(cpu_batch / med, 3), "unit": "utterances/sec", "cores": cores, "kind": "port",
                               "sample": f"1 warm-up + {cpu_steps} eval forwards of oracle/w2v2_ref.py (eager torch fp32) + argmax + collapse, "
                                         f"B = {cpu_batch} x 10 s, median {med:.2f} s"}
    except Exception as e:  # the baseline leg must not take the line down
        out["cpu_baseline"] = {"error": repr(e)}
    return out


def align_line(n=32, F=1500, L=300, V=32, reps=20, device="cuda:0", cpu_utts=2):
    """Forced alignment (ssak/utils/align_transcriptions.py:27-123 called per utterance from :294-402):
    ssak_ctc_forced_align_batch on n x (F frames, L tokens), emissions and tokens resident in HBM, HIP events on the launch stream.
    Beside it the reference's host loop (a Python loop of torch CPU ops over the frames + the backtrack walk), timed on a few
    utterances of the same batch."""
    from ssak_amd import hip
    g = torch.Generator().manual_seed(F + L)
    em = torch.log_softmax(torch.randn(n, F, V, generator=g) * 2.0, -1)
    tok = torch.randint(1, V, (n, L), generator=g, dtype=torch.int32)
    em_d, tok_d = em.to(device), tok.to(device)
    fl = torch.full((n,), F, dtype=torch.int32, device=device)
    tl = torch.full((n,), L, dtype=torch.int32, device=device)
    path_token = torch.empty((n, F), dtype=torch.int32, device=device)
    path_logp = torch.empty((n, F), dtype=torch.float32, device=device)
    info = torch.empty((n, 2), dtype=torch.int32, device=device)
    ws = torch.empty(hip.lib.ssak_ctc_align_batch_workspace_bytes(n, F, L), dtype=torch.uint8, device=device)

    def launch():
        hip.check(hip.lib.ssak_ctc_forced_align_batch(hip.ptr(em_d), hip.ptr(fl), hip.ptr(tok_d), hip.ptr(tl), n, F, V, L, 0, None, None,
                                                      hip.ptr(path_token), hip.ptr(path_logp), hip.ptr(info), hip.ptr(ws), ws.numel(), hip.stream()))

    with torch.cuda.device(device):
        for _ in range(3):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            launch()
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    cnt = info.cpu().numpy()[:, 0]
    algo_bytes = n * (F * V * 4 + L * 4 + F * 8)  # emissions in, tokens in, path (token, log-prob) out
    out = {"workload": f"ssak_ctc_forced_align_batch: {n} utterances x ({F} frames, {L} tokens, V = {V}), Viterbi trellis + backtrack in "
                       "one launch, one workgroup per utterance (align_transcriptions.py:27-123)",
           "value": round(n / (us * 1e-6), 1), "unit": "utterances/sec", "us_per_launch": round(us, 1), "us_per_utterance": round(us / n, 2),
           "us_per_frame": round(us / F, 4), "aligned": int((cnt >= 0).sum()), "utterances": n,
           "roofline": {"bound": "latency", "note": "F sequential lattice steps per utterance; one workgroup (one CU) per utterance",
                        "achieved": round(n / (us * 1e-6) / min(n, 256), 1), "unit": "utterances/s/CU", "peak": None, "frac": None,
                        "cus_busy": min(n, 256), "cus": 256, "hbm_gbs": round(algo_bytes / (us * 1e-6) / 1e9, 2),
                        "algorithmic_mb_per_launch": round(algo_bytes / 1e6, 2)},
           "timed_with": "HIP events on the launch stream around the launches"}
    try:
        from oracle import align_ref
        from oracle.gen_golden_align import backtrack_torch, trellis_torch
        torch.set_num_threads(_cpu_cores())
        ts, same = [], True
        pt_h, info_h = path_token.cpu().numpy(), info.cpu().numpy()
        for i in range(cpu_utts):
            t0 = time.perf_counter()
            tr = trellis_torch(em[i], tok[i].tolist(), 0)
            path = backtrack_torch(tr, em[i], tok[i].tolist(), 0)
            ts.append(time.perf_counter() - t0)
            c, first = int(info_h[i, 0]), int(info_h[i, 1])
            same = same and path is not None and c == len(path) and first == path[0][1] and [p[0] for p in path] == pt_h[i, first:first + c].tolist()
        med = sorted(ts)[len(ts) // 2]
        out["cpu_baseline"] = {"value": round(1.0 / med, 3), "unit": "utterances/sec", "cores": _cpu_cores(), "kind": "port",
                               "ms_per_utterance": round(med * 1e3, 1), "device_path_identical": bool(same),
                               "sample": f"{cpu_utts} utterances of the same batch through the reference's per-frame loop of torch CPU ops "
                                         "(get_trellis + backtrack as oracle/gen_golden_align.py restates them; that file pins them to the "
                                         "reference's own functions bit for bit)"}
        del align_ref
    except Exception as e:
        out["cpu_baseline"] = {"error": repr(e)}
    return out


def wer_line(B=32, F=499, reps=50, device="cuda:0"):
    """Evaluation metric on the device (compute_metrics of wav2vec_train.py:107-125 pulls the logits to the host,
    argmaxes in numpy and scores strings): ssak_ctc_greedy_decode + ssak_ctc_wer on an eval batch of logits resident in HBM; only
    two integers per utterance would cross to the host.  The CPU restatement (oracle/wer_ref.py) on the same batch beside it."""
    from ssak_amd import hip
    from ssak_amd.metrics import token_classes
    from ssak_amd.synth import VOCAB, synth_text, text_to_ids
    rng = np.random.default_rng(7)
    V = len(VOCAB)
    refs = [text_to_ids(synth_text(rng, 60, 120)) for _ in range(B)]
    Lm = max(len(r) for r in refs)
    labels = np.full((B, Lm), -100, np.int64)
    logits = rng.standard_normal((B, F, V)).astype(np.float32)
    for b, r in enumerate(refs):
        labels[b, :len(r)] = r
        hyp = list(r)
        for k in rng.choice(len(hyp), max(1, len(hyp) // 10), replace=False):  # ~10 % character errors
            hyp[k] = int(rng.integers(5, V))
        pos = np.sort(rng.choice(F // 2, len(hyp), replace=False)) * 2  # (a blank frame between any two emitted tokens)
        logits[b, :, 0] += 6.0
        logits[b, pos, 0] -= 6.0
        logits[b, pos, hyp] += 8.0
    lg, lab = torch.tensor(logits).to(device), torch.tensor(labels).to(device)
    cls = token_classes(VOCAB).to(device)

    def step():
        ids, n = hip.ctc_greedy_decode(lg, None, 0)
        return hip.ctc_wer(ids, n, lab, cls)

    with torch.cuda.device(device):
        for _ in range(3):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            edits, nref = step()
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    e_h, r_h = edits.cpu().numpy(), nref.cpu().numpy()
    algo_bytes = B * (F * V * 4 + Lm * 4 + 8)
    out = {"workload": f"ssak_ctc_greedy_decode + ssak_ctc_wer on an eval batch: {B} x [{F}, {V}] fp32 logits in HBM, {Lm}-token references "
                       "(compute_metrics, wav2vec_train.py:107-125)",
           "value": round(B / (us * 1e-6), 1), "unit": "utterances/sec", "us_per_batch": round(us, 1), "wer": round(float(e_h.sum()) / max(1, int(r_h.sum())), 4),
           "roofline": {"bound": "hbm", "achieved": round(algo_bytes / (us * 1e-6) / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(algo_bytes / (us * 1e-6) / 1e9 / 8000.0, 5), "algorithmic_mb_per_batch": round(algo_bytes / 1e6, 2),
                        "note": "two launches of a few microseconds on a 2 MB batch: launch-latency-bound, not bandwidth-bound"},
           "timed_with": "HIP events on the launch stream around (decode, wer) pairs"}
    try:
        from oracle import wer_ref
        t0 = time.perf_counter()
        ce, cr, cw = wer_ref.compute_metrics(logits.argmax(-1), labels, VOCAB, 0)
        dc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(B / dc, 1), "unit": "utterances/sec", "cores": 1, "kind": "port", "ms_per_batch": round(dc * 1e3, 2),
                               "counts_identical": bool(np.array_equal(ce, e_h) and np.array_equal(cr, r_h)),
                               "sample": "the same batch through oracle/wer_ref.py (numpy argmax + the reference's string pipeline + word-level edit distance)"}
    except Exception as e:
        out["cpu_baseline"] = {"error": repr(e)}
    return out
