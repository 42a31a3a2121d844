#!/usr/bin/env python
"""Headline benchmark: utterances/sec of one full Wav2Vec2-base CTC train step on N MI355X.

Metric and config are BASELINE.json's: 16 kHz x 10 s utterances, Wav2Vec2-base (Wav2Vec2Config() defaults),
bf16 compute, feature encoder frozen, CTC reduction "mean", the train script's regularisers on
(ssak/train/transformers/wav2vec_train.py:161-165,313-329), AdamW + clip.  A step = waveform normalise ->
forward -> CTC loss+grad -> backward -> [RCCL all-reduce] -> grad-norm clip + AdamW, on a synthetic batch that is
already resident in HBM.  One process per GPU; for N > 1 launch under torch.distributed.run (weak scaling:
the per-GPU batch is fixed).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md section 8d, algorithmic GFLOP per 10 s utterance (2 x MAC; no recompute, no padding, no optimizer):
GF_FEATURE_ENCODER = 49.08          # forward only (frozen)
GF_FIXED_FWD = 0.39 + 4.72 + 0.02   # feature projection + positional conv + lm_head
GF_LAYER_FWD = 7.829                # one encoder layer (QKVO 2.36, attention 0.77, FFN 4.71)
GF_PER_UTT_TRAIN = 346.32           # = 49.08 + 3 x (5.13 + 12 x 7.829): all 12 layers kept
PEAK_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (spec)
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak (spec); ~6.3 TB/s is what a streaming kernel achieves


def gf_per_utt(kept_layers: float) -> float:
    """Algorithmic work of one train step with `kept_layers` encoder layers surviving LayerDrop (forward + 2 x backward)."""
    return GF_FEATURE_ENCODER + 3.0 * (GF_FIXED_FWD + kept_layers * GF_LAYER_FWD)


def host_cores() -> int:
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota (the reference uses
    every core it sees, ssak/utils/env.py:86-90; oversubscribing a quota-limited box only slows it down)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, int(os.environ.get("SSAK_CPU_THREADS", "64"))))


def cpu_baseline(warmup: int = 3, timed_steps: int = 10, budget_s: float = 420.0):
    """The CPU restatement (oracle: eager torch fp32 on all host cores as ssak/utils/env.py:86-90 does, gradient checkpointing
    on as wav2vec_train.py:329 enables it) timed on a bounded sample of the same workload with SURVEY.md section 8d's protocol:
    B = 8, 3 warm-up + 10 timed full train steps, median (min / max beside it).  `budget_s` only guards a pathologically slow
    host: the timed steps stop early (never below 5) when the whole leg would run past it."""
    from oracle import w2v2_ref as R
    from ssak_amd.synth import synth_batch
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = R.W2V2Config.base()
    p = {n: t.clone().requires_grad_(not R.is_feature_encoder_param(n)) for n, t in R.init_params(cfg, 69).items()}
    opt = torch.optim.AdamW([t for t in p.values() if t.requires_grad], lr=1e-4, weight_decay=0.0)
    B = 8
    waves, labels = synth_batch(B, 160000, seed=99)
    x = torch.tensor(R.zero_mean_unit_var_norm(list(waves)))
    lab = torch.tensor(labels)
    rs = np.random.RandomState(0)
    times = []
    t_all = time.time()
    while len(times) < warmup + timed_steps and (len(times) < warmup + 5 or (time.time() - t_all) + times[-1] < budget_s):
        t0 = time.time()
        mask = torch.tensor(R.compute_mask_indices((B, 499), cfg.mask_time_prob, cfg.mask_time_length, None, 2, rng=rs))
        keep = rs.rand(cfg.num_hidden_layers) >= cfg.layerdrop
        loss, _ = R.forward(p, cfg, x, None, lab, train=True, mask_time_indices=mask, layer_keep=keep, gradient_checkpointing=True)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([t for t in p.values() if t.requires_grad], 1.0)
        opt.step()
        times.append(time.time() - t0)
    timed = sorted(times[warmup:])
    med = timed[len(timed) // 2]
    return {"value": round(B / med, 4), "unit": "utterances/sec", "cores": cores, "kind": "port",
            "min": round(B / timed[-1], 4), "median": round(B / med, 4), "max": round(B / timed[0], 4), "timed_steps": len(timed),
            "warmup_steps": warmup,
            "sample": f"{warmup} warm-up + {len(timed)} timed full train steps (fwd + bwd with per-layer gradient checkpointing + clip + AdamW) "
                      f"of oracle/w2v2_ref.py, eager torch fp32, B={B} x 10 s, median step {med:.2f} s "
                      f"(fastest {timed[0]:.2f} s, slowest {timed[-1]:.2f} s)"}


def build_stamp():
    """Commit / source hash / sha256 of the library this process loaded (tools/stamp.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stamp
    return stamp.current()


def traffic_of(kernel_name: str, build: dict):
    """HBM bytes per launch of `kernel_name` from the newest committed rocprofv3 --pmc passes -- but only if they were taken on
    THIS build (profiles/rNN_hbm_traffic.json carries the stamp of the build it measured: library sha256 and the sha256 of the
    kernel sources): a number from other sources is refused, not reported."""
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            tj = json.load(open(fn))
            rel = os.path.relpath(fn, ROOT)
            st = tj.get("stamp", {})
            same_lib = st.get("lib_sha256") == build["lib_sha256"]
            # (hipcc's output is not bit-reproducible across incremental builds: the identity that matters is the kernel SOURCES)
            same_src = st.get("source_sha256") == build["source_sha256"] and not build.get("sources_modified_since_commit")
            if not (same_lib or same_src):
                return None, f"refused: {rel} was measured on another build of the library (neither its library nor its source stamp matches the loaded libssak_hip.so)"
            return (tj["kernels"][kernel_name]["hbm_bytes_per_launch"],
                    f"{rel} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes of: {tj['command']}; "
                    f"{'same library' if same_lib else 'same kernel sources, library rebuilt'})")
        except (OSError, KeyError, ValueError):
            return None, None
    return None, None


# The fused attention kernels are bound by VALU issue, not by the matrix pipe: their ceiling follows from
# the wave-instruction count per score element.  One SIMD issues a 64-lane VALU instruction every 4 cycles: 1024 SIMDs x 16
# lanes per cycle x 2.4 GHz lane-instructions per second; a score element carries 4 x 64 flops forward (S, O) and 8 x 64 backward.
VALU_LANE_RATE = 1024 * 16 * 2.4e9
ATTN_VALU_FALLBACK = {"fwd": (13.2, "estimate from the instruction stream (no PMC pass of this build)"),
                      "bwd": (27.0, "estimate from the instruction stream (no PMC pass of this build)")}


def attention_valu_per_element(build: dict, elements_per_launch: float):
    """VALU wave-instructions x 64 lanes per score element of the attention kernels, from the newest committed SQ_INSTS_VALU pass
    of THIS build (profiles/rNN_pmc_sq.json, stamped like the traffic file); an estimate that says so otherwise."""
    import glob
    out = dict(ATTN_VALU_FALLBACK)
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq.json")), reverse=True):
        try:
            tj = json.load(open(fn))
            st = tj.get("stamp", {})
            same = st.get("lib_sha256") == build["lib_sha256"] or (st.get("source_sha256") == build["source_sha256"] and not build.get("sources_modified_since_commit"))
            if not same:
                break
            k = tj["kernels"]
            pick = lambda pre: sum(v["SQ_INSTS_VALU_per_launch"] for n, v in k.items() if n.startswith(pre) and "<true" in n)
            rel = os.path.relpath(fn, ROOT)
            if pick("attn_fwd_kernel"):
                out["fwd"] = (round(pick("attn_fwd_kernel") * 64 / elements_per_launch, 2), f"{rel}: SQ_INSTS_VALU x 64 / score elements per launch")
            if pick("attn_bwd_dq_kernel") and pick("attn_bwd_dkv_kernel"):
                out["bwd"] = (round((pick("attn_bwd_dq_kernel") + pick("attn_bwd_dkv_kernel")) * 64 / elements_per_launch, 2),
                              f"{rel}: SQ_INSTS_VALU x 64 / score elements per launch, dQ + dK/dV kernels")
        except (OSError, KeyError, ValueError):
            pass
        break
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU per step")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads (Whisper-small, XLSR-large bucketed, ingest)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the per-GPU batch sweep (8, 16) and the un-folded normalisation run after the timed region")
    ap.add_argument("--long-steps", type=int, default=100, help="extra untimed-by-the-driver run after the timed region (0 = skip)")
    args = ap.parse_args()
    build = build_stamp()  # before anything touches the GPU: it spawns git (no child processes under a profiler's preload later)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py --gpus N>1 must be launched under torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    # rehearsal switches for a one-GPU box (never set by the driver): all ranks on one card, gloo instead of RCCL
    if os.environ.get("SSAK_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("SSAK_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)

    from ssak_amd import hip
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.synth import synth_batch
    from ssak_amd.trainer import AdamW, Trainer

    cfg = Wav2Vec2Config()  # wav2vec2-base + the train script's regularisers
    model = Wav2Vec2ForCTC(cfg, device=dev, freeze_feature_encoder=True, seed=69).train()
    # seeded random init of the architecture (no network for checkpoints): same scales as HF's _init_weights
    g = torch.Generator(device="cpu").manual_seed(69)
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
    model.load_state_dict(sd)
    opt = AdamW(model, lr=1e-4, weight_decay=0.0, max_grad_norm=1.0, warmup_steps=500, total_steps=100000)
    trainer = Trainer(model, opt, measure_stall=True)
    if os.environ.get("SSAK_TILE_ORDER") == "1":  # development switch: ticket tile order without a process group
        model.set_option(hip.W2V2_OPT_DYNAMIC_TILES, 1)
    if os.environ.get("SSAK_BENCH_POSCONV_GEMM") == "1":  # A/B switch: the positional convolution as the Toeplitz GEMM of rounds 1-2
        model.set_option(hip.W2V2_OPT_POSCONV_DIRECT, 0)
    if os.environ.get("SSAK_BENCH_FRAGMENTS") == "1":  # A/B switch: the opt-in B-direct GEMM form with fragment-ordered weight copies
        model.set_option(hip.W2V2_OPT_FRAGMENT_WEIGHTS, 1)
    trainer.broadcast_parameters()

    T = int(round(args.seconds * 16000))
    B = args.batch
    waves_np, labels_np = synth_batch(B, T, seed=1234 + rank)
    waves = torch.tensor(waves_np).to(dev)
    labels = torch.tensor(labels_np).to(dev)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Roofline leg: HIP events around GEMM launches, recorded by the library on the launch stream.  Bracketing EVERY GEMM of
    # a step costs ~3 % of it (an event pair keeps consecutive kernels from overlapping head to tail), so the survey of all
    # kernel classes runs inside two of the warm-up steps and the timed region brackets only the dominant slot.
    # Order of the W warm-up steps: [cold steps: two when W >= 5, one when W = 4] -> the survey steps -> the remaining plain steps.  The plain steps
    # come LAST, right before the timed region: reading the survey's events back takes the host a while, the GPU idles
    # meanwhile and drops its clocks, and a timed region that starts from that state pays 20-50 ms of ramp-up (measured: 20
    # timed steps read 16.9-17.9 ms per step right after the read-back, 14.9-15.0 ms with two plain steps in between).
    # launch timing is per stream: under data parallelism the clip + AdamW tail runs on the trainer's side stream, so the
    # switch is set and the slots are read on both (one GPU: the tail runs in line and there is only the compute stream)
    def prof_enable(mode):
        hip.prof_enable(mode)
        if trainer.opt_stream is not None:
            with torch.cuda.stream(trainer.opt_stream):
                hip.prof_enable(mode)

    def prof_collect():
        rows = hip.prof_collect()
        if trainer.opt_stream is not None:
            with torch.cuda.stream(trainer.opt_stream):
                side = hip.prof_collect()
            merged = {r[0]: r for r in rows}
            for r in side:
                if r[0] in merged:
                    a = merged[r[0]]
                    merged[r[0]] = (a[0], a[1] + r[1], a[2] + r[2], a[3] + r[3], a[4])
            names = {r[0] for r in rows}
            # (the compute stream's list keeps its order -- `dom` below indexes it; slots only the side stream recorded follow)
            rows = [merged[r[0]] for r in rows] + [r for r in side if r[0] not in names]
        return rows

    n_cold = 2 if args.warmup >= 5 else (1 if args.warmup >= 4 else 0)  # first launches load code objects, size workspaces, allocate copies
    n_survey = min(2, args.warmup - n_cold)
    for _ in range(n_cold):
        trainer.train_step(waves, None, labels)
    sync()
    survey, dom, dom_kernel, dom_slots = None, -1, None, []
    if n_survey:
        prof_enable(1)
        prof_collect()
        ts = time.perf_counter()
        for _ in range(n_survey):
            trainer.train_step(waves, None, labels)
        sync()
        survey_dt = time.perf_counter() - ts
        prof_enable(0)
        survey = prof_collect()
        # the dominant KERNEL = the instantiation with the largest summed time (a persistent GEMM instantiation serves several
        # products and has one slot per (N, K): its slots are added up); inside the timed region only its LARGEST slot is
        # bracketed -- an event pair around each of its ~55 launches per step would cost ~2 % of the headline -- and all of its
        # slots are bracketed in a few steps run right after the timed region (roofline pass)
        base = lambda n: n.split(" (N = ")[0]
        by_kernel = {}
        for i, pr in enumerate(survey):
            if pr[1] > 0:
                by_kernel.setdefault(base(pr[0]), []).append(i)
        dom_kernel = max(by_kernel, key=lambda k: sum(survey[i][2] for i in by_kernel[k]))
        dom_slots = by_kernel[dom_kernel]
        dom = max(dom_slots, key=lambda i: survey[i][2])
    prof_enable(2 + dom if dom >= 0 else 1)
    if os.environ.get("SSAK_BENCH_NO_PROF") == "1":  # development switch: cost of the events
        prof_enable(0)
    for _ in range(args.warmup - n_cold - n_survey):
        trainer.train_step(waves, None, labels)
    sync()
    prof_collect()  # (drops what the plain warm-up steps recorded for the dominant slot)
    # no Python garbage collection inside the timed region: a generation-2 pass over the survey's objects stopped the host for
    # ~40 ms in the first timed step of every other run (SSAK_BENCH_STEP_TRACE=1: host issue 41.9 ms, device 57.1 ms for that step)
    import gc
    gc.collect()
    gc.disable()
    sync()
    fw0, kl0 = model.train_forwards, model.kept_layers
    trace = [] if os.environ.get("SSAK_BENCH_STEP_TRACE") == "1" else None  # development switch: per-step device times on stderr
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if trace is not None:
            trace.append((torch.cuda.Event(enable_timing=True), time.perf_counter()))
            trace[-1][0].record()
        loss = trainer.train_step(waves, None, labels)
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    if trace:
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        torch.cuda.synchronize()
        evs = [e for e, _ in trace] + [end]
        print("per-step device ms:", " ".join(f"{evs[i].elapsed_time(evs[i + 1]):.1f}" for i in range(len(trace))), file=sys.stderr)
        print("host issue ms:", " ".join(f"{(trace[i + 1][1] - trace[i][1]) * 1e3:.1f}" for i in range(len(trace) - 1)), file=sys.stderr)
    prof_enable(0)
    prof = prof_collect()
    kept_avg = (model.kept_layers - kl0) / max(1, model.train_forwards - fw0)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    final_loss = float(loss.item())
    # exposed part of the exchange + optimizer tail (side stream): how long a forward waits for the previous update, sampled
    # on a few extra untimed steps (reading the events synchronises)
    stalls = []
    for _ in range(3):
        trainer.train_step(waves, None, labels)
        trainer.train_step(waves, None, labels)
        stalls.append(trainer.stall_ms())
    sync()
    # roofline pass: every launch of the dominant kernel (all its products) bracketed, a few steps right after the timed region
    roof_pass, roof_steps = None, 0
    if dom_slots:
        roof_steps = 6
        hip.prof_enable_slots(dom_slots)
        if trainer.opt_stream is not None:
            with torch.cuda.stream(trainer.opt_stream):
                hip.prof_enable_slots(dom_slots)
        prof_collect()
        for _ in range(roof_steps):
            trainer.train_step(waves, None, labels)
        sync()
        prof_enable(0)
        roof_pass = [r for r in prof_collect() if r[1] > 0]
    # a longer run of the same step after the driver's timed region (profiler markers off): box noise and LayerDrop's
    # step-to-step work differences average out over it
    long_run = None
    if args.long_steps > 0:
        prof_enable(0)
        t1 = time.perf_counter()
        for _ in range(args.long_steps):
            trainer.train_step(waves, None, labels)
        sync()
        dl = time.perf_counter() - t1
        if world > 1:
            tl = torch.tensor([dl], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tl, op=torch.distributed.ReduceOp.MAX)
            dl = float(tl.item())
        long_run = {"steps": args.long_steps, "value": round(B * world * args.long_steps / dl, 2), "ms_per_step": round(dl / args.long_steps * 1e3, 3),
                    "note": "same step, run after the timed region, no event markers"}

    # "per-GPU batch B in {8, 16, 32}; report best" -- the other batch sizes through the same model / trainer,
    # a short timed run each (the reference's default is --batch_size 8, wav2vec_train.py:158); the headline stays the --batch line
    sweep = []
    if world == 1 and not args.no_sweep and not args.no_secondary:  # (profiler passes run --no-secondary: only the headline's launches)
        for Bs in (8, 16):
            if Bs == B:
                continue
            kl1, fw1 = model.kept_layers, model.train_forwards
            w_s, l_s = waves[:Bs].contiguous(), labels[:Bs].contiguous()
            for _ in range(3):
                trainer.train_step(w_s, None, l_s)
            sync()
            n_s = 30
            ts = time.perf_counter()
            for _ in range(n_s):
                trainer.train_step(w_s, None, l_s)
            sync()
            ds = time.perf_counter() - ts
            kept_s = (model.kept_layers - kl1) / max(1, model.train_forwards - fw1)
            tf_s = gf_per_utt(kept_s) * 1e9 * Bs * n_s / ds / 1e12
            sweep.append({"per_gpu_batch": Bs, "value": round(Bs * n_s / ds, 2), "ms_per_step": round(ds / n_s * 1e3, 3), "steps": n_s,
                          "whole_step_tflops": round(tf_s, 1), "whole_step_frac": round(tf_s / PEAK_BF16_TFLOPS, 4)})
        # and the un-folded step (ssak_wave_normalize as its own pass, what a padded batch takes) beside the folded one: alternating
        # blocks of ten steps on the same clock state (the boxes drift by a per cent over a second; a single pair of runs read as noise)
        t_fold = t_own = 0.0
        for _ in range(2):
            trainer.train_step(waves, None, labels, fold_norm=False)
        for _ in range(3):
            for own in (False, True):
                sync()
                ts = time.perf_counter()
                for _ in range(10):
                    trainer.train_step(waves, None, labels, fold_norm=False if own else None)
                sync()
                if own:
                    t_own += time.perf_counter() - ts
                else:
                    t_fold += time.perf_counter() - ts
        folded_ms, unfolded_ms = t_fold / 30 * 1e3, t_own / 30 * 1e3
    if rank == 0:
        utts = B * world * args.steps
        value = utts / dt

        attn_valu = attention_valu_per_element(build, float(B) * cfg.num_attention_heads * model.num_frames(T) ** 2)

        def entry(p, steps):
            name, launches, ms, work, bound = p
            e = {"kernel": name, "bound": bound, "launches_per_step": round(launches / steps, 2),
                 "us_per_step": round(ms * 1e3 / steps, 1), "avg_launch_us": round(ms * 1e3 / launches, 2)}
            if name.startswith("attn_"):
                # VALU-issue-bound: the ceiling is what the VALU pipes allow at this kernel's instruction count
                which = "fwd" if name.startswith("attn_fwd") else "bwd"
                v, src = attn_valu[which]
                ceil_tf = (256.0 if which == "fwd" else 512.0) * VALU_LANE_RATE / v / 1e12
                tf = work / (ms * 1e-3) / 1e12
                e.update(bound="valu", achieved=round(tf, 1), peak=round(ceil_tf, 1), unit="TFLOP/s", valu_per_score_element=v,
                         valu_source=src, mfma_frac=round(tf / PEAK_BF16_TFLOPS, 4), algorithmic_gflop_per_step=round(work / steps / 1e9, 1))
            elif name.startswith("ctc_"):
                # latency-bound (F sequential frames per utterance, one 2-wave workgroup per utterance): utterances per second per
                # busy CU and the occupancy, with the log-prob traffic beside them
                n_utt = B * launches
                e.update(achieved=round(n_utt / (ms * 1e-3) / min(B, 256), 1), peak=None, unit="utterances/s/CU",
                         occupancy={"workgroups": B, "waves_per_workgroup": 2, "cus_busy": min(B, 256), "cus": 256,
                                    "waves_per_busy_cu": 2, "of_waves_per_cu": 32},
                         hbm_gbs=round(work / (ms * 1e-3) / 1e9, 1), algorithmic_mb_per_step=round(work / steps / 1e6, 1))
                e["frac"] = None
                return e
            elif bound == "mfma":
                e.update(achieved=round(work / (ms * 1e-3) / 1e12, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                         algorithmic_gflop_per_step=round(work / steps / 1e9, 1))
            else:
                e.update(achieved=round(work / (ms * 1e-3) / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                         algorithmic_mb_per_step=round(work / steps / 1e6, 1))
            e["frac"] = round(e["achieved"] / e["peak"], 4)
            return e

        # dominant kernel = the instantiation with the largest summed duration in the survey; its largest product is bracketed
        # inside the timed region, all of its products in the roofline pass right after it
        prof = [p for p in prof if p[1] > 0]
        prof.sort(key=lambda p: -p[2])
        roof = None
        if prof or roof_pass:
            if roof_pass:
                rp_ms, rp_work, rp_launches = sum(r[2] for r in roof_pass), sum(r[3] for r in roof_pass), sum(r[1] for r in roof_pass)
                roof = entry((dom_kernel, rp_launches, rp_ms, rp_work, roof_pass[0][4]), roof_steps)
                roof["products"] = [entry(r, roof_steps) for r in sorted(roof_pass, key=lambda r: -r[2])]
                roof["timed_with"] = (f"HIP events around every launch of this kernel (all its products) in {roof_steps} steps run right after the "
                                      "timed region; bracketing them inside it would cost ~2 % of the headline (an event pair keeps kernels from "
                                      "overlapping head to tail)")
                if prof:
                    roof["timed_region_slot"] = dict(entry(prof[0], args.steps), note="the kernel's largest product, bracketed INSIDE the timed region")
            else:
                roof = entry(prof[0], args.steps)
                roof["timed_with"] = "HIP events around every launch of this slot inside the timed region"
            name = roof["kernel"]
            if survey and dom_slots:
                roof["share_of_step"] = round(sum(survey[i][2] for i in dom_slots) * 1e-3 / survey_dt, 4)
                roof["dominant_by"] = "largest summed kernel time in the survey steps (every launch of the step bracketed)"
            traffic, traffic_src = traffic_of(name, build)
            roof.update(traffic=traffic, traffic_source=traffic_src,
                        kept_layers_per_step=round(kept_avg, 3),
                        whole_step_tflops=round(gf_per_utt(kept_avg) * 1e9 * B * args.steps / dt / 1e12, 1),
                        whole_step_frac=round(gf_per_utt(kept_avg) * 1e9 * B * args.steps / dt / 1e12 / PEAK_BF16_TFLOPS, 4),
                        whole_step_note="algorithmic GFLOP of the layers that actually ran (LayerDrop 0.1 drops ~1.2 of 12 per step)")
            if survey:  # every kernel class, from the last warm-up steps (all launches bracketed: costs a few % of those steps)
                sv = sorted((p for p in survey if p[1] > 0), key=lambda p: -p[2])
                gm = [p for p in sv if p[0].startswith("gemm")]
                sv_ms = sum(p[2] for p in sv)
                roof["kernels"] = [dict(entry(p, n_survey), share_of_step=round(p[2] * 1e-3 / survey_dt, 4)) for p in sv]
                if gm:
                    g_tf = sum(p[3] for p in gm) / (sum(p[2] for p in gm) * 1e-3) / 1e12
                    roof["primary"] = {"what": "every GEMM launch of the train step (all instantiations; time-weighted = total algorithmic "
                                               "FLOPs / total GEMM time)", "bound": "mfma", "achieved": round(g_tf, 1), "peak": PEAK_BF16_TFLOPS,
                                       "unit": "TFLOP/s", "frac": round(g_tf / PEAK_BF16_TFLOPS, 4),
                                       "share_of_step": round(sum(p[2] for p in gm) * 1e-3 / survey_dt, 3),
                                       "source": f"last {n_survey} warm-up steps, every launch bracketed by HIP events"}
                roof["survey"] = {"source": f"last {n_survey} warm-up steps, every launch bracketed by HIP events",
                                  "all_gemm_tflops": round(sum(p[3] for p in gm) / (sum(p[2] for p in gm) * 1e-3) / 1e12, 1),
                                  "gemm_share_of_step": round(sum(p[2] for p in gm) * 1e-3 / survey_dt, 3),
                                  "bracketed_share_of_step": round(sv_ms * 1e-3 / survey_dt, 3)}
        out = {"metric": "utterances/sec (16 kHz, 10 s) Wav2Vec2-base CTC train step", "value": round(value, 2),
               "unit": "utterances/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic, HBM-resident batch (the timed loop re-feeds one device batch; the "
               "file -> device path is measured beside it: ingest)",
               "config": {"workload": "Wav2Vec2-base CTC fine-tune step, bf16, synthetic 10 s @16 kHz utterances "
                                      "(BASELINE.json configs[1]; DP over xGMI for n_gpus>1 = configs[2])",
                          "per_gpu_batch": B, "global_batch": B * world, "samples_per_utt": T, "frames": model.num_frames(T),
                          "frozen_feature_encoder": True, "regularisers": "script defaults (dropout/layerdrop/specaugment on)",
                          "parallelism": f"dp{world}", "final_loss": round(final_loss, 4),
                          "kept_layers_per_step": round(kept_avg, 3)},
               "roofline": roof, "build": build,
               "optimizer_tail": {"stream": "side" if trainer.opt_stream is not None else "compute",
                                  "exposed_us_per_step": None if np.isnan(np.median(stalls)) else round(1e3 * float(np.median(stalls)), 1),
                                  "bucket_wait_us": trainer.bucket_wait_us(),
                                  "note": "exposed_us_per_step: wait of the forward at its first trainable-parameter read (after the frozen "
                                          "conv stack) for the exchange tail + clip + AdamW of the previous step; bucket_wait_us: per "
                                          "gradient bucket (announcement order), time the optimizer stream sat at its all-reduce "
                                          "(null without a process group)"}}
        if world > 1:
            out["exchange"] = {"backend": backend, "rccl_ranks": world, "collective": "sum all-reduce per gradient bucket, "
                               "issued from the engine's grad-ready callback while the backward runs",
                               "grad_dtype": trainer.grad_exchange_dtype,
                               "bucket_bytes": [c * (2 if trainer.grad_exchange_dtype == "bf16" else 4) for _, c in trainer.bucket_log],
                               "payload_bytes_per_step": sum(c for _, c in trainer.bucket_log) * (2 if trainer.grad_exchange_dtype == "bf16" else 4)}
        out["long_run"] = long_run
        if sweep:
            tf_h = gf_per_utt(kept_avg) * 1e9 * B * args.steps / dt / 1e12
            rows = sorted(sweep + [{"per_gpu_batch": B, "value": round(value, 2), "ms_per_step": round(dt / args.steps * 1e3, 3), "steps": args.steps,
                                    "whole_step_tflops": round(tf_h, 1), "whole_step_frac": round(tf_h / PEAK_BF16_TFLOPS, 4)}], key=lambda r: r["per_gpu_batch"])
            out["headline_sweep"] = {"rows": rows, "best_per_gpu_batch": max(rows, key=lambda r: r["value"])["per_gpu_batch"],
                                     "note": f"SURVEY.md 8d: B in {{8, 16, 32}}, report best; `value` is the B = {B} line (the timed region), the others are "
                                             "30-step runs of the same model right after it"}
            out["normalisation"] = {"folded_ms_per_step": round(folded_ms, 3), "own_pass_ms_per_step": round(unfolded_ms, 3),
                                    "note": "the headline batch is un-padded (every utterance fills T), so ssak_amd.train and this bench fold the waveform "
                                            "normalisation (a1) into conv0's GroupNorm statistics; padded / ragged batches and the layer-norm topology run "
                                            "ssak_wave_normalize as a pass of its own (both figures: three alternating blocks of ten steps each, after the timed region)"}
        if world == 1 and not args.no_secondary:
            # BASELINE.json configs[3] / configs[4] and the ingest path on the same clock as the headline (their models are built after
            # the headline's timed region; its buffers are released first)
            del trainer, opt, model
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import side_benches
            sec = []
            # secondary[0..1]: BASELINE configs[3] / [4]; [2]: the ssak/infer path; [3]: forced alignment (f1); [4]: evaluation metric (f3)
            for fn, kw in ((side_benches.whisper_sweep, dict()), (side_benches.xlsr_sweep, dict()), (side_benches.infer_line, dict()),
                           (side_benches.align_line, dict()), (side_benches.wer_line, dict())):
                try:
                    sec.append(fn(**kw))
                except Exception as e:  # a secondary line must not take the headline down with it
                    sec.append({"workload": fn.__name__, "error": repr(e)})
                torch.cuda.empty_cache()
            out["secondary"] = sec
            # the log-mel front end (a13) as a line of roofline.kernels[]: it is not part of the headline step, so its slot comes
            # from the Whisper window step above (same library, same process) and says so
            lm = next((e.get("logmel") for e in sec if isinstance(e, dict) and e.get("logmel")), None)
            if lm and out.get("roofline") and out["roofline"].get("kernels") is not None:
                out["roofline"]["kernels"].append({
                    "kernel": lm["kernel"], "bound": "hbm", "launches_per_step": 1.0, "us_per_step": lm["us_per_call"],
                    "avg_launch_us": lm["us_per_call"], "achieved": lm["achieved_gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "algorithmic_mb_per_step": round(lm["algorithmic_mb_per_window"] * sec[0].get("batch", 8), 2), "frac": lm["frac"],
                    "share_of_step": lm["share_of_step"],
                    "workload": f"secondary[0]: Whisper-small window step, B = {sec[0].get('batch', 8)} (per ssak_logmel_whisper call; share = of THAT step)"})
            try:
                out["ingest"] = side_benches.ingest_rate()
            except Exception as e:
                out["ingest"] = {"error": repr(e)}
            # the headline step fed from the section-8d Kaldi folder through `train.py --online`'s loop
            try:
                out["online"] = side_benches.online_steps(B=B)
                out["online"]["vs_resident_batch"] = round(out["online"]["value"] / value, 4)
            except Exception as e:
                out["online"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
