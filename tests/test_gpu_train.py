"""GPU: the training path end to end -- matched loss against the fp32 oracle over many optimizer steps, vocabulary
padding, and the Kaldi-folder train / infer command lines of the reference on a synthetic corpus."""
import dataclasses
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-12))


def _cfg(Wav2Vec2Config, oc):
    d = dataclasses.asdict(oc)
    d.pop("initializer_range")
    return Wav2Vec2Config(**d)


def test_matched_loss_50_steps_vs_fp32_oracle():
    """BASELINE.md "matched loss": 50 optimizer steps (AdamW lr 1e-3 after a 5-step warm-up, clip 1.0) from the same
    seeded init on the same batch, stochastic ops off: the bf16 HIP loss curve stays within 2e-2 relative of the fp32
    oracle curve (eager torch + torch.optim.AdamW) at every step."""
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer, linear_warmup_lr
    oc = R.W2V2Config.tiny().deterministic()
    p0 = R.init_params(oc, 13)
    rng = np.random.default_rng(0)
    x = R.zero_mean_unit_var_norm([rng.standard_normal(8000).astype(np.float32) for _ in range(4)])
    labels = R.pad_labels([list(rng.integers(1, 32, n)) for n in (6, 4, 7, 5)])
    steps, lr, warm = 50, 1e-3, 5
    # oracle curve
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
    # HIP curve
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    model.load_state_dict(p0)
    tr = Trainer(model, AdamW(model, lr=lr, warmup_steps=warm, total_steps=1000, max_grad_norm=1.0))
    xd, ld = torch.tensor(x).cuda(), torch.tensor(labels).cuda()
    got = [float(tr.train_step(xd, None, ld, raw=False).item()) for _ in range(steps)]
    rel = max(abs(a - b) / abs(b) for a, b in zip(got, ref))
    print("matched loss: first", got[0], ref[0], "last", got[-1], ref[-1], "max rel", rel)
    assert ref[-1] < 0.8 * ref[0]          # the oracle actually learns on this batch
    assert rel < 2e-2


@pytest.mark.parametrize("skip", [True, False])
def test_optimizer_step_of_layerdrop_skipped_layers(skip):
    """A layer LayerDrop skips has no gradient.  ``AdamW(skip_unused_layers=True)`` = torch >= 2.0 defaults (``.grad`` is None: the
    layer's weights, moments and per-parameter step count are left alone); the default = a zero gradient that takes the step
    (``zero_grad(set_to_none=False)`` / DistributedDataParallel).  Both against the fp32 oracle + ``torch.optim.AdamW`` over eight
    steps with explicit decisions, compared on the weights of a layer that is skipped in some steps."""
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer, linear_warmup_lr
    oc = R.W2V2Config.tiny(num_hidden_layers=3).deterministic()
    p0 = R.init_params(oc, 21)
    rng = np.random.default_rng(5)
    x = R.zero_mean_unit_var_norm([rng.standard_normal(8000).astype(np.float32) for _ in range(4)])
    labels = R.pad_labels([list(rng.integers(1, 32, n)) for n in (6, 4, 7, 5)])
    keeps = [[1, 1, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1], [1, 0, 0], [1, 1, 1], [1, 0, 1], [1, 1, 1]]
    lr, warm = 1e-3, 2
    names = R.trainable_names(oc)
    q = {n: (t.clone().requires_grad_(n in names)) for n, t in p0.items()}
    opt = torch.optim.AdamW([q[n] for n in names], lr=lr, weight_decay=0.0)
    for s, keep in enumerate(keeps):
        for g in opt.param_groups:
            g["lr"] = linear_warmup_lr(lr, s, warm, 1000)
        loss, _ = R.forward(q, oc, torch.tensor(x), None, torch.tensor(labels), layer_keep=keep)
        opt.zero_grad(set_to_none=skip)
        loss.backward()
        if not skip:  # (the first time a layer is skipped before it ever had a gradient, torch has nothing to zero)
            for n in names:
                if q[n].grad is None:
                    q[n].grad = torch.zeros_like(q[n])
        torch.nn.utils.clip_grad_norm_([q[n] for n in names if q[n].grad is not None], 1.0)
        opt.step()
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc), exact=True).train()  # fp32-exact engine: the two semantics differ by ~40 %
    model.load_state_dict(p0)                                             # of layer 1's movement, the engines by ~1e-3
    opt2 = AdamW(model, lr=lr, warmup_steps=warm, total_steps=1000, max_grad_norm=1.0, skip_unused_layers=skip)
    xd, ld = torch.tensor(x).cuda(), torch.tensor(labels).cuda()
    for keep in keeps:
        model(xd, labels=ld, layer_keep=keep)
        model.backward()
        opt2.step(layer_keep=model.last_layer_keep)
    sd = model.state_dict()
    worst, which = 0.0, None
    for n in names:
        if ".encoder.layers." not in n or not n.endswith(".weight") or p0[n].dim() != 2:
            continue
        a, b = sd[n].float().cpu(), q[n].detach()
        d = float((a - b).norm() / (b - p0[n]).norm())
        if d > worst:
            worst, which = d, n
    print("skip" if skip else "zero-gradient", "semantics: worst (engine - torch) / (torch's movement) over the layers' matrices", worst, which)
    assert worst < 3e-2, (worst, which)


def test_vocab_not_multiple_of_8():
    """A 29-symbol tokenizer: the head is padded to 32 inert classes; logits / loss / head gradient match the oracle."""
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    oc = dataclasses.replace(R.W2V2Config.tiny().deterministic(), vocab_size=29)
    p = R.init_params(oc, 4)
    rng = np.random.default_rng(3)
    x = R.zero_mean_unit_var_norm([rng.standard_normal(8000).astype(np.float32) for _ in range(2)])
    labels = R.pad_labels([[28, 3, 5], [1, 2]])
    loss, logits, grads = R.loss_and_grads(p, oc, torch.tensor(x), None, torch.tensor(labels))
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    model.load_state_dict(p)
    out = model(torch.tensor(x), labels=torch.tensor(labels))
    assert out.logits.shape[-1] == 29
    assert (out.logits.cpu() - logits).norm() / logits.norm() < 2e-2
    assert abs(out.loss.item() - loss.item()) < 2e-2 * loss.item()
    model.backward()
    g = model.grad("lm_head.weight").cpu()
    assert (g[:29] - grads["lm_head.weight"]).norm() / grads["lm_head.weight"].norm() < 6e-2
    assert float(g[29:].abs().max()) == 0.0 and model.state_dict()["lm_head.weight"].shape == (29, 64)
    with pytest.raises(ValueError):
        model(torch.tensor(x), labels=torch.tensor([[29], [1]]))


def test_train_and_infer_cli_on_kaldi_folder(tmp_path):
    """The reference's command lines (wav2vec_train.py TRAIN VALID --flags, transformers_infer DATA --model) on a
    synthetic Kaldi corpus: output folder layout (init_eval.json, checkpoint-N/trainer_state.json, final/), a loss that
    goes down, and inference that writes one `id transcript` line per utterance."""
    from ssak_amd import data as D
    from ssak_amd.checkpoint import save_pretrained
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.synth import VOCAB, synth_text, synth_wave
    from oracle import w2v2_ref as R
    rng = np.random.default_rng(0)
    kd = tmp_path / "kaldi"
    (kd / "audio").mkdir(parents=True)
    with open(kd / "wav.scp", "w") as fw, open(kd / "text", "w") as ft, open(kd / "utt2dur", "w") as fd:
        for i in range(8):
            n = int(rng.integers(16000, 24000))
            D.write_wav(str(kd / "audio" / f"u{i}.wav"), synth_wave(rng, n))
            fw.write(f"utt{i} sox {kd}/audio/u{i}.wav -t wav -r 16k -b 16 -c 1 - |\n")
            ft.write(f"utt{i} {synth_text(rng, 3, 6)} <noise>\n")
            fd.write(f"utt{i} {n / 16000:.3f}\n")
    oc = dataclasses.replace(R.W2V2Config.tiny(), layerdrop=0.0)
    base = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc))
    base.load_state_dict(R.init_params(oc, 1))
    save_pretrained(base, D.CharTokenizer(VOCAB), str(tmp_path / "base"))
    del base
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "ssak_amd.train", str(kd), str(kd), "--base_model", str(tmp_path / "base"),
                        "--batch_size", "4", "--num_epochs", "20", "--eval_steps", "20", "--learning_rate", "3e-3",
                        "--min_duration", "0", "--output_dir", str(tmp_path / "out")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    # two folders (wav2vec_train.py:238-239,397-410): the run's, and the "untrained" one that caches the initial evaluation
    outs = sorted(d for d in os.listdir(tmp_path / "out") if "_adamwt" in d)
    cache = sorted(d for d in os.listdir(tmp_path / "out") if "_adamwt" not in d)
    assert len(outs) == 1 and outs[0].startswith("hf_") and "_bs-4_" in outs[0] and outs[0].endswith("_s-69_adamwt")
    assert len(cache) == 1 and outs[0].startswith(cache[0] + "_lr-0.003") and (tmp_path / "out" / cache[0] / "init_eval.json").exists()
    run = tmp_path / "out" / outs[0]
    assert "ssak_amd" in open(run / "README.txt").read()
    assert (run / "init_eval.json").exists() and (run / "final" / "model.safetensors").exists()
    st = json.load(open(run / "checkpoint-40" / "trainer_state.json"))
    train_losses = [e["loss"] for e in st["log_history"] if "loss" in e]
    evals = [e["eval_loss"] for e in st["log_history"] if "eval_loss" in e]
    assert st["global_step"] == 40 and len(train_losses) == 2 and train_losses[1] < train_losses[0]
    assert evals[-1] < json.load(open(run / "init_eval.json"))["eval_loss"]
    # resume (wav2vec_train.py:245,415): an interrupted copy of the run -- checkpoint-20 only -- continues to the same end:
    # same batches from the seed, same optimizer moments and step, same second-interval loss
    import shutil
    shutil.copytree(tmp_path / "out", tmp_path / "out_resume")
    run_r = tmp_path / "out_resume" / outs[0]
    shutil.rmtree(run_r / "checkpoint-40")
    shutil.rmtree(run_r / "final")
    r = subprocess.run([sys.executable, "-m", "ssak_amd.train", str(kd), str(kd), "--base_model", str(tmp_path / "base"),
                        "--batch_size", "4", "--num_epochs", "20", "--eval_steps", "20", "--learning_rate", "3e-3",
                        "--min_duration", "0", "--output_dir", str(tmp_path / "out_resume")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "resuming from" in r.stdout and "checkpoint-20" in r.stdout
    st_r = json.load(open(run_r / "checkpoint-40" / "trainer_state.json"))
    losses_r = [e["loss"] for e in st_r["log_history"] if "loss" in e]
    assert st_r["global_step"] == 40 and len(losses_r) == 2 and losses_r[0] == train_losses[0]
    assert abs(losses_r[1] - train_losses[1]) < 1e-3 * abs(train_losses[1])
    assert (run_r / "final" / "model.safetensors").exists()
    # --online: the same training with audio read on the fly and decoded / normalised on the device (ssak_amd.ingest)
    # must follow the same loss trajectory (same seed, same batches; dropout is off in this config)
    r = subprocess.run([sys.executable, "-m", "ssak_amd.train", str(kd), str(kd), "--base_model", str(tmp_path / "base"),
                        "--batch_size", "4", "--num_epochs", "20", "--eval_steps", "20", "--learning_rate", "3e-3", "--online",
                        "--min_duration", "0", "--output_dir", str(tmp_path / "out_online")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    run2 = tmp_path / "out_online" / [d for d in os.listdir(tmp_path / "out_online") if d.endswith("_adamwt_online")][0]
    st2 = json.load(open(run2 / "checkpoint-40" / "trainer_state.json"))
    losses2 = [e["loss"] for e in st2["log_history"] if "loss" in e]
    assert len(losses2) == 2 and losses2[1] < losses2[0]
    r = subprocess.run([sys.executable, "-m", "ssak_amd.infer", str(kd), "--model", str(run / "final"), "--use_ids",
                        "--batch_size", "3", "--output", str(tmp_path / "hyp.txt")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = open(tmp_path / "hyp.txt").read().splitlines()
    assert len(lines) == 8 and sorted(l.split()[0] for l in lines) == [f"utt{i}" for i in range(8)]


@pytest.mark.parametrize("topology", ["base", "xlsr"])
def test_dropout_masks_replayed_consistently(topology):
    """All dropout sites on (feat_proj, encoder input, attention, hidden x2, activation, final) with a fixed step seed:
    the loss is a deterministic function of the parameters, and a step of size eps along the analytic gradient g must
    change it by eps*|g| to first order.  A mask regenerated differently in the backward at any site shrinks the
    measured slope by roughly the keep probability of that site; tolerance 12 %."""
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    kw = dict(attention_dropout=0.25, hidden_dropout=0.25, activation_dropout=0.25, feat_proj_dropout=0.25, final_dropout=0.25,
              layerdrop=0.0, mask_time_prob=0.0)
    if topology == "xlsr":
        kw.update(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)
    oc = R.W2V2Config.tiny(**kw)
    p = R.init_params(oc, 21)
    rng = np.random.default_rng(8)
    x = torch.tensor(R.zero_mean_unit_var_norm([rng.standard_normal(8000).astype(np.float32) for _ in range(4)]))
    labels = torch.tensor(R.pad_labels([list(rng.integers(1, 32, n)) for n in (6, 4, 7, 5)]))
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    model.load_state_dict(p)

    def loss_at(with_grad=False):
        model._step_seed = 12345  # the same masks on every call
        out = model(x, labels=labels)
        if with_grad:
            model.backward()
        return float(out.loss.item())

    l0 = loss_at(True)
    assert loss_at() == l0
    n = model.num_trainable
    g = model.grads[:n].clone()
    gn = float(g.norm())
    base = model.params[:n].clone()
    slopes = []
    for eps in (0.02, 0.04):
        model.params[:n] = base + eps * g / gn
        model.sync_weights(full=True)
        lp = loss_at()
        model.params[:n] = base - eps * g / gn
        model.sync_weights(full=True)
        lm = loss_at()
        slopes.append((lp - lm) / (2 * eps))
    print(topology, "analytic |g|", gn, "measured slopes", slopes)
    assert abs(slopes[0] - gn) < 0.12 * gn and abs(slopes[1] - gn) < 0.12 * gn


# ------------------------------------------------------------------ data parallelism on the real trainer
def _dp_problem():
    from oracle import w2v2_ref as R
    oc = R.W2V2Config.tiny().deterministic()
    rng = np.random.default_rng(3)
    x = R.zero_mean_unit_var_norm([rng.standard_normal(9000).astype(np.float32) for _ in range(8)])
    labels = R.pad_labels([list(rng.integers(1, 32, 5)) for _ in range(8)])  # equal target lengths: shard means average exactly
    return oc, R.init_params(oc, 21), x, labels


def _dp_worker(rank, world, port, out, exchange="fp32"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SSAK_DP_GRAD_DTYPE=exchange)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)  # both ranks share the one card of the test box
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.data import shard_batch
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer
    oc, p0, x, labels = _dp_problem()
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    model.load_state_dict(p0)
    tr = Trainer(model, AdamW(model, lr=1e-3, warmup_steps=2, total_steps=100, max_grad_norm=1.0))
    tr.broadcast_parameters()
    mine = shard_batch(list(range(8)), rank, world)
    xd, ld = torch.tensor(x[mine]).cuda(), torch.tensor(labels[mine]).cuda()
    losses = [float(tr.train_step(xd, None, ld, raw=False).item()) for _ in range(4)]
    if rank == 0:
        torch.save({"params": model.params[:model.num_trainable].cpu(), "losses": losses}, out)  # (rank-0 shard losses)
    torch.distributed.destroy_process_group()


class _GlooStandInComm:
    """Stand-in for hip.Comm (RCCL through the C ABI) with the same contract -- ``all_reduce(tensor, offset, count, stream_)``
    ordered on the GIVEN stream, in-place sum over ranks -- carried by gloo through the host.  Everything else of
    Trainer(exchange="c") is the real thing: the private exchange stream, the `ready` event behind the kernels that produced the
    gradient range, the `done` event the optimizer stream waits for, the drain before torch.distributed's own collectives."""

    def __init__(self):
        self.calls = []

    def all_reduce(self, t, offset=0, count=None, stream_=None):
        n = t.numel() - offset if count is None else count
        st = torch.cuda.ExternalStream(stream_.value)
        self.calls.append((int(offset), int(n), int(stream_.value)))
        with torch.cuda.stream(st):
            host = t.view(-1)[offset:offset + n].float().cpu()  # (on `st`: behind whatever the caller ordered it behind)
            torch.distributed.all_reduce(host)
            t.view(-1)[offset:offset + n].copy_(host.to(t.dtype), non_blocking=True)

    def close(self):
        self.calls.append("closed")


def _dp_c_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.data import shard_batch
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer
    oc, p0, x, labels = _dp_problem()
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    model.load_state_dict(p0)
    comm = _GlooStandInComm()
    tr = Trainer(model, AdamW(model, lr=1e-3, warmup_steps=2, total_steps=100, max_grad_norm=1.0), exchange="c", comm=comm, measure_stall=True)
    tr.broadcast_parameters()
    mine = shard_batch(list(range(8)), rank, world)
    xd, ld = torch.tensor(x[mine]).cuda(), torch.tensor(labels[mine]).cuda()
    losses = []
    for step in range(4):
        losses.append(float(tr.train_step(xd, None, ld, raw=False).item()))
        if step == 1:  # a torch.distributed collective between two steps, as train.evaluate issues one: the exchange is drained first
            tr.drain_exchange()
            probe = torch.ones(1, device="cuda")
            torch.distributed.all_reduce(probe)
            assert float(probe.item()) == world
    ranges = [(o, c) for o, c in model.announced_grad_ranges()]
    per_step = [c for c in comm.calls if c != "closed"]
    assert len(per_step) == 4 * len(ranges) and [(o, c) for o, c, _ in per_step[:len(ranges)]] == ranges
    assert len({s for _, _, s in per_step}) == 1 and per_step[0][2] != torch.cuda.current_stream().cuda_stream  # its own stream
    waits = tr.bucket_wait_us()
    assert waits is not None and len(waits) == len(ranges)
    tr.close()
    tr.close()  # idempotent
    assert comm.calls[-1] == "closed" and comm.calls.count("closed") == 1
    if rank == 0:
        torch.save({"params": model.params[:model.num_trainable].cpu(), "losses": losses}, out)
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_dp2_exchange_c_event_ordering_with_a_stand_in_communicator(tmp_path):
    """Trainer(exchange="c") -- the library's own communicator on a private stream -- on two ranks with gloo standing in for RCCL
    behind the communicator's interface: the buckets arrive in the engine's announced order on ONE stream that is not the compute
    stream, each behind the kernels that produced its range and ahead of the optimizer tail that consumes it (a mis-ordered event
    reduces stale gradients and the parameters leave the single-process run), a torch.distributed collective between steps
    follows a drained exchange, and close() destroys the communicator once.  What stays unexercised is RCCL itself."""
    import socket
    import torch.multiprocessing as mp
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dpc.pt")
    mp.spawn(_dp_c_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    oc, p0, x, labels = _dp_problem()
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    model.load_state_dict(p0)
    tr = Trainer(model, AdamW(model, lr=1e-3, warmup_steps=2, total_steps=100, max_grad_norm=1.0))
    xd, ld = torch.tensor(x).cuda(), torch.tensor(labels).cuda()
    for _ in range(4):
        tr.train_step(xd, None, ld, raw=False)
    ref = model.params[:model.num_trainable].cpu()
    start = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc))
    start.load_state_dict(p0)
    p_init = start.params[:start.num_trainable].cpu()
    du_ref, du_dp = ref - p_init, got["params"] - p_init
    rel = float((du_dp - du_ref).norm() / du_ref.norm())
    assert du_ref.abs().max() > 1e-3 and rel < 0.1, rel


@pytest.mark.timeout(600)
@pytest.mark.parametrize("exchange", ["fp32", "bf16"])
def test_dp2_trainer_equals_single_process(tmp_path, exchange):
    """(exchange = "bf16": the optional half-width gradient exchange -- buckets rounded to bf16 for the all-reduce.)
    Two ranks of the REAL trainer (engine grad-ready callbacks -> bucketed async all-reduce -> fused clip + AdamW with the
    1/world scale) on half the batch each reproduce the single-process run on the whole batch: same parameters after 4
    steps within bf16-engine noise.  gloo stands in for RCCL so that both ranks fit on the one GPU of the test box."""
    import socket
    import torch.multiprocessing as mp
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dp.pt")
    mp.spawn(_dp_worker, args=(2, port, out, exchange), nprocs=2, join=True)
    got = torch.load(out)
    oc, p0, x, labels = _dp_problem()
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    model.load_state_dict(p0)
    tr = Trainer(model, AdamW(model, lr=1e-3, warmup_steps=2, total_steps=100, max_grad_norm=1.0))
    xd, ld = torch.tensor(x).cuda(), torch.tensor(labels).cuda()
    for _ in range(4):
        tr.train_step(xd, None, ld, raw=False)
    ref = model.params[:model.num_trainable].cpu()
    start = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc))
    start.load_state_dict(p0)
    p_init = start.params[:start.num_trainable].cpu()
    # the update both runs made, compared as a whole: AdamW's m / sqrt(v) turns bf16 noise on near-zero gradients into
    # +-lr moves, so single parameters may differ by a step while the update vectors agree
    du_ref, du_dp = ref - p_init, got["params"] - p_init
    rel = float((du_dp - du_ref).norm() / du_ref.norm())
    assert du_ref.abs().max() > 1e-3 and rel < 0.1, rel


def _dp_uneven_problem():
    from oracle import w2v2_ref as R
    oc = R.W2V2Config.tiny().deterministic()
    rng = np.random.default_rng(13)
    x = R.zero_mean_unit_var_norm([rng.standard_normal(9000).astype(np.float32) for _ in range(7)])
    labels = R.pad_labels([list(rng.integers(1, 32, n)) for n in (5, 3, 8, 4, 6, 2, 7)])  # unequal target lengths on purpose
    # 7 utterances in batches of 4: a full batch, the short last batch of the epoch (3 -> shards of 2 and 1), a batch of ONE
    # (rank 1's shard is empty) and a batch of 5 (3 + 2)
    batches = [[0, 1, 2, 3], [4, 5, 6], [0], [1, 2, 3, 4, 5]]
    return oc, R.init_params(oc, 22), x, labels, batches


def _dp_uneven_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.data import shard_batch
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer
    oc, p0, x, labels, batches = _dp_uneven_problem()
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    model.load_state_dict(p0)
    tr = Trainer(model, AdamW(model, lr=1e-3, warmup_steps=2, total_steps=100, max_grad_norm=1.0))
    tr.broadcast_parameters()
    sizes = []
    for whole in batches:
        mine = shard_batch(whole, rank, world)
        sizes.append(len(mine))
        if mine:
            tr.train_step(torch.tensor(x[mine]).cuda(), None, torch.tensor(labels[mine]).cuda(), raw=False, global_count=len(whole))
        else:
            tr.train_step(None, None, None, global_count=len(whole))
    torch.save({"params": model.params[:model.num_trainable].cpu(), "sizes": sizes}, f"{out}.{rank}")
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_dp2_uneven_shards_weighted_by_utterance_count(tmp_path):
    """SURVEY.md section 8e "with unequal shard sizes weight by utterance count": HF trains on the short last batch of an epoch
    (dataloader_drop_last=False, docker/transformers_modified/trainer.py:834).  Two ranks over 7 utterances in batches of 4
    -- shards of 2+2, 2+1, 1+0 (an EMPTY shard: zeros through the same collectives) and 3+2, unequal target lengths -- end
    where the single-process run on the whole batches ends, and both ranks hold the same parameters."""
    import socket
    import torch.multiprocessing as mp
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dp_uneven.pt")
    mp.spawn(_dp_uneven_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["sizes"] == [2, 2, 1, 3] and r1["sizes"] == [2, 1, 0, 2]
    assert torch.equal(r0["params"], r1["params"])
    oc, p0, x, labels, batches = _dp_uneven_problem()
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    model.load_state_dict(p0)
    tr = Trainer(model, AdamW(model, lr=1e-3, warmup_steps=2, total_steps=100, max_grad_norm=1.0))
    for whole in batches:
        tr.train_step(torch.tensor(x[whole]).cuda(), None, torch.tensor(labels[whole]).cuda(), raw=False)
    ref = model.params[:model.num_trainable].cpu()
    start = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc))
    start.load_state_dict(p0)
    p_init = start.params[:start.num_trainable].cpu()
    du_ref, du_dp = ref - p_init, r0["params"] - p_init
    rel = float((du_dp - du_ref).norm() / du_ref.norm())
    # an unweighted mean of the shard means (what equal-shard code would do) lands ~0.3 away on this problem
    assert du_ref.abs().max() > 1e-3 and rel < 0.1, rel


def test_train_cli_best_model_early_stopping_and_rotation(tmp_path, monkeypatch):
    """What decides the reference's output model (wav2vec_train.py:368-372,392,419-420): metric_for_best_model="wer",
    load_best_model_at_end, save_total_limit=2, EarlyStoppingCallback.  ssak_amd.train.main() runs in process with the
    evaluation's WER scripted to 0.5, 0.4, 0.6, 0.7 (the last evaluations are WORSE) and the patience set to 2: training
    stops after the 4th evaluation although steps remain, the 2nd checkpoint (best) and the 4th (newest) survive the
    rotation, and final/ holds the best checkpoint's weights bit for bit -- not the last ones."""
    from safetensors.numpy import load_file
    from oracle import w2v2_ref as R
    from ssak_amd import data as D
    from ssak_amd import train as T
    from ssak_amd.checkpoint import save_pretrained
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.synth import VOCAB, synth_text, synth_wave
    rng = np.random.default_rng(0)
    kd = tmp_path / "kaldi"
    (kd / "audio").mkdir(parents=True)
    with open(kd / "wav.scp", "w") as fw, open(kd / "text", "w") as ft, open(kd / "utt2dur", "w") as fd:
        for i in range(7):  # 7 utterances, batches of 4: every epoch ends in a short batch of 3 (trained, not dropped)
            n = int(rng.integers(16000, 24000))
            D.write_wav(str(kd / "audio" / f"u{i}.wav"), synth_wave(rng, n))
            fw.write(f"utt{i} {kd}/audio/u{i}.wav\n")
            ft.write(f"utt{i} {synth_text(rng, 3, 6)}\n")
            fd.write(f"utt{i} {n / 16000:.3f}\n")
    oc = dataclasses.replace(R.W2V2Config.tiny(), layerdrop=0.0)
    base = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc))
    base.load_state_dict(R.init_params(oc, 1))
    save_pretrained(base, D.CharTokenizer(VOCAB), str(tmp_path / "base"))
    del base
    script = iter([0.5, 0.4, 0.6, 0.7, 0.1, 0.1, 0.1, 0.1])
    real_eval = T.evaluate

    def scripted(model, tok, waves, labels, batch_size, rank=0, world=1):
        m = real_eval(model, tok, waves, labels, batch_size, rank, world)
        m["eval_wer"] = next(script)
        return m

    monkeypatch.setattr(T, "evaluate", scripted)
    monkeypatch.setattr(T, "EARLY_STOPPING_PATIENCE", 2)
    T.main([str(kd), str(kd), "--base_model", str(tmp_path / "base"), "--batch_size", "4", "--num_epochs", "20", "--eval_steps", "5",
            "--learning_rate", "3e-3", "--min_duration", "0", "--disable_first_eval", "--output_dir", str(tmp_path / "out")])
    run = tmp_path / "out" / [d for d in os.listdir(tmp_path / "out") if "_adamwt" in d][0]
    cks = sorted(d for d in os.listdir(run) if d.startswith("checkpoint-"))
    assert cks == ["checkpoint-10", "checkpoint-20"], cks  # best (2nd evaluation) + newest (4th); 35 steps were planned
    st = json.load(open(run / "checkpoint-20" / "trainer_state.json"))
    assert st["global_step"] == 20 and st["max_steps"] == 35 and st["best_metric"] == 0.4
    assert st["best_model_checkpoint"].endswith("checkpoint-10") and st["early_stopping_patience_counter"] == 2
    # epochs count the short batch as a step: 2 steps per epoch
    assert [e["epoch"] for e in st["log_history"] if "loss" in e] == [2.5, 5.0, 7.5, 10.0]
    final, best, last = (load_file(str(run / d / "model.safetensors")) for d in ("final", "checkpoint-10", "checkpoint-20"))
    assert all(np.array_equal(final[k], best[k]) for k in best)
    assert any(not np.array_equal(final[k], last[k]) for k in last)
    assert os.path.exists(run / "checkpoint-20" / "rng-rank0.json")


@pytest.mark.timeout(900)
def test_train_cli_two_rank_resume_restores_each_ranks_regulariser_streams(tmp_path):
    """A resumed data-parallel run continues the uninterrupted one: the trainer seeds rank r's dropout / SpecAugment / LayerDrop
    streams with seed + r, every rank writes its own rng-rank{r}.json into a checkpoint, and on resume every rank reads ITS
    file (round-2 advice: rank 0's state used to be loaded everywhere, after which all ranks drew identical masks).  Two ranks
    over gloo on the one card: the streams at step 10 of a run resumed from checkpoint-5 equal those of the uninterrupted run,
    rank by rank, and differ between the ranks."""
    import shutil
    import socket
    from oracle import w2v2_ref as R
    from ssak_amd import data as D
    from ssak_amd.checkpoint import save_pretrained
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.synth import VOCAB, synth_text, synth_wave
    rng = np.random.default_rng(2)
    kd = tmp_path / "kaldi"
    (kd / "audio").mkdir(parents=True)
    with open(kd / "wav.scp", "w") as fw, open(kd / "text", "w") as ft, open(kd / "utt2dur", "w") as fd:
        for i in range(9):  # 9 utterances in batches of 4 over 2 ranks: 2+2, 2+2, 1+0
            n = int(rng.integers(16000, 24000))
            D.write_wav(str(kd / "audio" / f"u{i}.wav"), synth_wave(rng, n))
            fw.write(f"utt{i}\t{kd}/audio/u{i}.wav\n")
            ft.write(f"utt{i} {synth_text(rng, 3, 6)}\n")
            fd.write(f"utt{i} {n / 16000:.3f}\n")
    oc = R.W2V2Config.tiny()  # dropouts, LayerDrop and SpecAugment at the script's defaults
    base = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc))
    base.load_state_dict(R.init_params(oc, 1))
    save_pretrained(base, D.CharTokenizer(VOCAB), str(tmp_path / "base"))
    del base
    torch.cuda.synchronize()

    def run(out):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, PYTHONPATH=ROOT, SSAK_DIST_BACKEND="gloo")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), "-m", "ssak_amd.train", str(kd), str(kd), "--base_model", str(tmp_path / "base"),
                            "--batch_size", "4", "--num_epochs", "4", "--eval_steps", "5", "--learning_rate", "1e-3", "--min_duration", "0",
                            "--disable_first_eval", "--output_dir", str(out)], env=env, capture_output=True, text=True, timeout=800)
        assert r.returncode == 0, r.stderr[-3000:]
        return r.stdout

    run(tmp_path / "out")
    name = [d for d in os.listdir(tmp_path / "out") if "_adamwt" in d][0]
    full = tmp_path / "out" / name
    assert json.load(open(full / "checkpoint-9" / "trainer_state.json"))["global_step"] == 9  # round(4 * 9 / 4)
    shutil.copytree(tmp_path / "out", tmp_path / "out_resume")
    res = tmp_path / "out_resume" / name
    shutil.rmtree(res / "checkpoint-9")
    shutil.rmtree(res / "final")
    assert "resuming from" in run(tmp_path / "out_resume")
    states = {}
    for tag, folder in (("full", full), ("resumed", res)):
        for r in (0, 1):
            states[tag, r] = json.load(open(folder / "checkpoint-9" / f"rng-rank{r}.json"))
    assert states["full", 0] == states["resumed", 0] and states["full", 1] == states["resumed", 1]
    assert states["full", 0]["step_seed"] != states["full", 1]["step_seed"]
    assert states["full", 0]["host_rng"] != states["full", 1]["host_rng"]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("topology", ["layer", "group"])
def test_train_cli_sharded_evaluation_equals_single_process(tmp_path, topology):
    """Every rank evaluates its contiguous shard of each validation batch (the reference's per_device_eval_batch_size =
    batch_size // num_devices, ssak/train/transformers/wav2vec_train.py:357) and one all-reduce of (edits, words, loss sum,
    utterances) gives the metrics: the initial evaluation of two ranks (gloo, one card) equals the single-process one -- word
    error counts exactly, the loss to fp32 summation order -- on 11 utterances in batches of 4 (shards 2+2, 2+2, 2+1).  Both
    topologies: the layer-norm (XLSR) model runs with the attention mask; the group-norm (base) model runs WITHOUT one, so its
    logits depend on how far an utterance was padded -- every shard is therefore padded to the longest utterance of the GLOBAL
    batch, as the reference's collator pads before DataParallel scatters (wav2vec_train.py:79-100), and the metrics do not depend
    on the world size."""
    import socket
    from oracle import w2v2_ref as R
    from ssak_amd import data as D
    from ssak_amd.checkpoint import save_pretrained
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.synth import VOCAB, synth_text, synth_wave
    rng = np.random.default_rng(5)
    kd = tmp_path / "kaldi"
    (kd / "audio").mkdir(parents=True)
    with open(kd / "wav.scp", "w") as fw, open(kd / "text", "w") as ft, open(kd / "utt2dur", "w") as fd:
        for i in range(11):
            n = int(rng.integers(16000, 30000))
            D.write_wav(str(kd / "audio" / f"u{i}.wav"), synth_wave(rng, n))
            fw.write(f"utt{i}\t{kd}/audio/u{i}.wav\n")
            ft.write(f"utt{i} {synth_text(rng, 3, 8)}\n")
            fd.write(f"utt{i} {n / 16000:.3f}\n")
    oc = (R.W2V2Config.tiny(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True) if topology == "layer" else R.W2V2Config.tiny()).deterministic()
    base = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc))
    base.load_state_dict(R.init_params(oc, 3))
    save_pretrained(base, D.CharTokenizer(VOCAB), str(tmp_path / "base"))
    del base
    torch.cuda.synchronize()

    def run(out, nproc):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, PYTHONPATH=ROOT, SSAK_DIST_BACKEND="gloo")
        launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                    "--master-port", str(port), "-m"] if nproc > 1 else [sys.executable, "-m"]
        r = subprocess.run(launcher + ["ssak_amd.train", str(kd), str(kd), "--base_model", str(tmp_path / "base"), "--batch_size", "4",
                                       "--num_epochs", "1", "--eval_steps", "100", "--learning_rate", "1e-4", "--min_duration", "0",
                                       "--output_dir", str(out)], env=env, capture_output=True, text=True, timeout=800)
        assert r.returncode == 0, r.stderr[-3000:]
        name = [d for d in os.listdir(out) if "_adamwt" in d][0]
        run_dir = os.path.join(out, name)
        st = json.load(open(os.path.join(run_dir, "checkpoint-3", "trainer_state.json")))
        return json.load(open(os.path.join(run_dir, "init_eval.json"))), [e for e in st["log_history"] if "eval_wer" in e][-1]

    one_init, one_last = run(tmp_path / "one", 1)
    two_init, two_last = run(tmp_path / "two", 2)
    assert one_init["eval_wer"] == two_init["eval_wer"] and one_init["eval_wer"] > 0
    assert abs(one_init["eval_loss"] - two_init["eval_loss"]) <= 1e-4 * abs(one_init["eval_loss"])
    # after three data-parallel steps both runs hold (nearly) the same weights: the metrics of the final evaluation agree too
    assert abs(one_last["eval_loss"] - two_last["eval_loss"]) <= 2e-2 * abs(one_last["eval_loss"])


@pytest.mark.timeout(900)
def test_bench_two_rank_rehearsal_reports_the_exchange(tmp_path):
    """bench.py under torch.distributed.run with two ranks (both on the one card, gloo standing in for RCCL -- the driver's
    N > 1 runs use one GPU per rank over RCCL): the JSON line carries what the N > 1 readings need -- the exposed optimizer /
    exchange tail and, per gradient bucket, how long the optimizer stream waited for its all-reduce -- non-null, with the
    bucket list covering the 360.8 MB payload of the base model."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT, SSAK_BENCH_SHARE_GPU="1", SSAK_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "3",
                        "--batch", "4"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8
    tail = d["optimizer_tail"]
    assert tail["stream"] == "side" and tail["exposed_us_per_step"] is not None and tail["exposed_us_per_step"] >= 0
    ex = d["exchange"]
    assert ex["payload_bytes_per_step"] == 4 * 90_195_872 and len(ex["bucket_bytes"]) == 15
    waits = tail["bucket_wait_us"]
    assert waits is not None and len(waits) == len(ex["bucket_bytes"]) and all(w >= 0 for w in waits)
    assert d["roofline"]["primary"]["frac"] > 0 and d["build"]["lib_sha256"]


def test_train_step_is_bitwise_reproducible():
    """No GLOBAL float atomics on the path (the CTC gradient sums posteriors in wave-private LDS bins with LDS atomics, whose
    order is fixed within a wave-instruction): two runs from the same seed (dropout, LayerDrop and SpecAugment ON) end in bit-identical
    parameters and losses -- bias gradients, the SpecAugment embedding gradient and the clip norm are all fixed-order sums."""
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer
    oc = R.W2V2Config.tiny()
    oc = dataclasses.replace(oc, mask_time_prob=0.2)
    p0 = R.init_params(oc, 5)
    rng = np.random.default_rng(1)
    x = torch.tensor(R.zero_mean_unit_var_norm([rng.standard_normal(12000).astype(np.float32) for _ in range(6)])).cuda()
    labels = torch.tensor(R.pad_labels([list(rng.integers(1, 32, n)) for n in (6, 4, 7, 5, 3, 8)])).cuda()

    def run():
        model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc), seed=123).train()
        model.load_state_dict(p0)
        tr = Trainer(model, AdamW(model, lr=1e-3, warmup_steps=1, total_steps=100, max_grad_norm=1.0))
        losses = [tr.train_step(x, None, labels, raw=False).item() for _ in range(5)]
        return losses, model.params.clone()

    (l1, p1), (l2, p2) = run(), run()
    assert l1 == l2
    assert torch.equal(p1, p2)


def _dp_layerdrop_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer
    oc = dataclasses.replace(R.W2V2Config.tiny(), layerdrop=0.5)
    model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc), seed=100 + rank).train()  # per-rank LayerDrop / dropout draws
    model.load_state_dict(R.init_params(oc, 21))
    tr = Trainer(model, AdamW(model, lr=1e-3, warmup_steps=1, total_steps=100, max_grad_norm=1.0))
    tr.broadcast_parameters()
    rng = np.random.default_rng(50 + rank)
    x = torch.tensor(R.zero_mean_unit_var_norm([rng.standard_normal(9000).astype(np.float32) for _ in range(4)])).cuda()
    labels = torch.tensor(R.pad_labels([list(rng.integers(1, 32, 5)) for _ in range(4)])).cuda()
    for _ in range(6):
        tr.train_step(x, None, labels, raw=False)
    torch.save(model.params[:model.num_trainable].cpu(), os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_dp2_ranks_stay_identical_with_per_rank_layerdrop(tmp_path):
    """Ranks that drop DIFFERENT layers still pair the same gradient ranges in every collective (announcements stay in layer
    order whatever a rank kept or dropped): after six steps both ranks hold bit-identical parameters."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_layerdrop_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a, b)


def test_torch_autograd_loop_matches_native_trainer():
    """The reference's loop shape -- loss.backward(), clip_grad_norm_, torch.optim.AdamW.step(), zero_grad() on a torch host
    loop (ssak_amd.autograd) -- against the fused native trainer from the same start: same losses, same parameters after three
    steps (both clip at 1.0; Adam's sign-like first steps are compared on the update as a whole).  And the CTC autograd
    function against torch's own F.ctc_loss gradient."""
    from oracle import w2v2_ref as R
    from ssak_amd.autograd import TorchWav2Vec2ForCTC, ctc_loss
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer
    oc = R.W2V2Config.tiny().deterministic()
    p0 = R.init_params(oc, 3)
    rng = np.random.default_rng(1)
    x = torch.tensor(R.zero_mean_unit_var_norm([rng.standard_normal(9000).astype(np.float32) for _ in range(4)])).cuda()
    labels = torch.tensor(R.pad_labels([list(rng.integers(1, 32, n)) for n in (6, 4, 7, 5)])).cuda()
    native = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    native.load_state_dict(p0)
    tr = Trainer(native, AdamW(native, lr=1e-3, warmup_steps=0, total_steps=1 << 40, max_grad_norm=1.0))
    tm = TorchWav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc)).train()
    tm.load_state_dict(p0)
    opt = torch.optim.AdamW(tm.parameters(), lr=1e-3, weight_decay=0.0)
    p_init = native.params.clone()
    for _ in range(3):
        l_native = tr.train_step(x, None, labels, raw=False).item()
        out = tm(x, labels=labels)
        out.loss.backward()
        assert float(tm.params.grad[tm.num_trainable:].abs().max()) == 0.0  # frozen feature encoder
        torch.nn.utils.clip_grad_norm_(tm.parameters(), 1.0)
        opt.step()
        tm.zero_grad()
        assert abs(out.loss.item() - l_native) < 2e-3 * abs(l_native)
    n = native.num_trainable
    du_n, du_t = (native.params - p_init)[:n], (tm.params.detach() - p_init)[:n]
    assert float(du_n.abs().max()) > 1e-3 and float((du_t - du_n).norm() / du_n.norm()) < 0.05
    # evaluation through the torch-visible model uses the updated weights (shadow refreshed on the version change)
    tm.eval()
    native.eval()
    assert rel_l2(tm(x).logits.cpu(), native(x).logits.cpu()) < 2e-2
    # CTC as an autograd function on logits with history
    lg = torch.randn(3, 40, 16, device="cuda", requires_grad=True)
    w = torch.randn(16, 16, device="cuda", requires_grad=True)
    tl = torch.tensor([[1, 2, 3, -1], [4, 5, -1, -1], [6, 7, 8, 9]], device="cuda")
    loss, nll = ctc_loss(lg @ w, None, tl)
    (loss * 2.0).backward()
    lg2, w2 = lg.detach().cpu().requires_grad_(True), w.detach().cpu().requires_grad_(True)
    lp = torch.log_softmax(lg2 @ w2, -1).transpose(0, 1)
    ref = torch.nn.functional.ctc_loss(lp, torch.tensor([1, 2, 3, 4, 5, 6, 7, 8, 9]), torch.full((3,), 40), torch.tensor([3, 2, 4]),
                                       blank=0, reduction="mean", zero_infinity=True)
    (ref * 2.0).backward()
    assert abs(loss.item() - ref.item()) < 1e-4 * abs(ref.item())
    assert rel_l2(lg.grad.cpu(), lg2.grad) < 1e-3 and rel_l2(w.grad.cpu(), w2.grad) < 1e-3


def test_optimizer_side_stream_equals_inline():
    """The optimizer tail on its own HIP stream (the data-parallel default: norm partials per bucket, clip + AdamW after the last
    one, the next forward waiting for the update only at the feature projection, ssak_w2v2_set_param_event) gives the same
    parameters, bit for bit, as the in-line tail after several steps -- forward / backward of step n+1 never see a half-applied
    update, and host reads of model.params right after train_step are ordered behind it."""
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    from ssak_amd.trainer import AdamW, Trainer
    oc = R.W2V2Config.tiny()  # regularisers ON: masks, LayerDrop and SpecAugment draws must line up too
    p0 = R.init_params(oc, 8)
    rng = np.random.default_rng(2)
    x = torch.tensor(R.zero_mean_unit_var_norm([rng.standard_normal(9000).astype(np.float32) for _ in range(4)])).cuda()
    labels = torch.tensor(R.pad_labels([list(rng.integers(1, 32, 5)) for _ in range(4)])).cuda()
    outs = []
    for side in (False, True):
        model = Wav2Vec2ForCTC(_cfg(Wav2Vec2Config, oc), seed=4).train()
        model.load_state_dict(p0)
        tr = Trainer(model, AdamW(model, lr=1e-3, warmup_steps=1, total_steps=100, weight_decay=0.01), optimizer_stream=side,
                     measure_stall=side)
        losses = [tr.train_step(x, None, labels, raw=False) for _ in range(5)]
        params = model.params.clone()  # no explicit synchronisation: the property orders this read behind the update
        outs.append((torch.stack(losses).cpu(), params.cpu(), tr.opt.grad_norm()))
        if side:
            assert tr.opt_stream is not None and tr.stall_ms() >= 0.0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
