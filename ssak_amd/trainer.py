"""Train step around the HIP engine: forward -> CTC -> backward -> (RCCL all-reduce) -> clip + AdamW.

Mirrors the inner loop the reference gets from HF ``Trainer`` (docker/transformers_modified/trainer.py:1755-1855,
2504-2548) with the settings of ssak/train/transformers/wav2vec_train.py:353-384: ``optim="adamw_torch"``,
lr 1e-4, weight_decay 0.0, warm-up 500 steps then linear decay, max_grad_norm 1.0, gradient accumulation 1.

Data parallelism is one process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI): utterances are
independent through forward/CTC/backward, so the only exchange is the sum all-reduce of the flat fp32 gradient buffer
(the reference's single-process ``nn.DataParallel`` gathers to GPU 0 instead, trainer.py:1345-1346).  The buffer is
reduced in buckets -- one per encoder layer as its gradients become final, plus one for the rest -- so the exchange
overlaps the remaining backward; xGMI is point-to-point, so a few 28 MB buckets keep every link busy without the
latency of many small collectives.
"""
from __future__ import annotations

import os

import torch

from . import hip
from .model import Wav2Vec2ForCTC


def linear_warmup_lr(base_lr: float, step: int, warmup_steps: int, total_steps: int) -> float:
    """lr for 0-based optimizer step ``step`` (transformers.get_linear_schedule_with_warmup)."""
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    return base_lr * max(0.0, (total_steps - step) / max(1, total_steps - warmup_steps))


def hf_decays(name: str, config) -> bool:
    """HF Trainer's weight-decay parameter group (docker/transformers_modified/trainer.py:1013-1024): every parameter that
    is not inside an ``nn.LayerNorm`` module and has no "bias" in its name.  The conv0 normalisation of the group-norm
    ("base") feature encoder is an ``nn.GroupNorm`` -- not in ``ALL_LAYERNORM_LAYERS`` -- so its weight IS decayed."""
    if "bias" in name:
        return False
    if "layer_norm" in name:
        group_norm0 = (getattr(config, "feat_extract_norm", "") == "group"
                       and name.startswith("wav2vec2.feature_extractor.conv_layers.0.layer_norm"))
        return group_norm0
    return True


def decay_ranges(model):
    """[(offset, count, decays)] covering [0, num_trainable) of the flat buffers, adjacent parameters of one group merged.
    The engine lays the trainable matrices out first and the vectors after them, so the default (frozen feature encoder)
    model is two ranges."""
    items = sorted((off, name) for name, (off, _, _) in model.layout.items() if off < model.num_trainable)
    out = []
    for i, (off, name) in enumerate(items):
        end = items[i + 1][0] if i + 1 < len(items) else model.num_trainable
        d = hf_decays(name, model.config)
        if out and out[-1][2] == d:
            out[-1] = (out[-1][0], end - out[-1][0], d)
        else:
            out.append((off, end - off, d))
    return out


class AdamW:
    """Flat-buffer AdamW with fused global-norm clipping (kernels: ssak_grad_sumsq / ssak_adamw_step) and HF Trainer's
    two weight-decay groups (matrices decayed, biases / LayerNorm affines not)."""

    def __init__(self, model: Wav2Vec2ForCTC, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 max_grad_norm=1.0, warmup_steps=500, total_steps=100000, skip_unused_layers: bool = False):
        """``skip_unused_layers``: what ``torch.optim.AdamW`` does under torch >= 2.0 defaults for an encoder layer LayerDrop skipped
        in this step (its ``.grad`` is None): weights, both moments and the layer's own step count (bias correction) stay as they
        are.  Off (the default): the skipped layer takes the step with a zero gradient, as under ``zero_grad(set_to_none=False)`` and
        under DistributedDataParallel whenever another rank ran the layer."""
        self.skip_unused_layers = bool(skip_unused_layers)
        self._segments = None      # [(offset, count, layer or -1, decays)] covering [0, n): built on first use
        self.layer_steps = None    # optimizer steps each encoder layer has taken (skip_unused_layers)
        self.model = model
        self.n = model.num_trainable
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm, self.warmup_steps, self.total_steps = max_grad_norm, warmup_steps, total_steps
        self.exp_avg = torch.zeros(self.n, dtype=torch.float32, device=model.device)
        self.exp_avg_sq = torch.zeros(self.n, dtype=torch.float32, device=model.device)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=model.device)
        self._sumsq_ws = torch.empty(1024, dtype=torch.float32, device=model.device)  # per-workgroup partials of the norm
        self.step_count = 0
        self.ranges = decay_ranges(model)

    def current_lr(self) -> float:
        return linear_warmup_lr(self.lr, self.step_count, self.warmup_steps, self.total_steps)

    def _update_ranges(self):
        """One sweep when nothing is decayed (the reference's default weight_decay 0.0), else one per group range."""
        if self.weight_decay == 0.0:
            return [(0, self.n, 0.0)]
        return [(off, cnt, self.weight_decay if d else 0.0) for off, cnt, d in self.ranges]

    def add_sumsq(self, offset: int, count: int, first: bool):
        """gnorm_sq (+)= sum(grads[offset:offset+count]^2) on the current stream: the norm partial of one bucket."""
        m = self.model
        fn = hip.lib.ssak_grad_sumsq if first else hip.lib.ssak_grad_sumsq_add
        hip.check(fn(hip.ptr(m.grads[offset:]), count, hip.ptr(self.gnorm_sq), hip.ptr(self._sumsq_ws),
                     self._sumsq_ws.numel() * 4, hip.stream()))

    def _layer_segments(self):
        """[(offset, count, encoder layer or -1, decays)] covering the trainable range, adjacent parameters of one layer and
        decay group merged."""
        if self._segments is None:
            import re
            m = self.model
            items = sorted((off, n, name) for name, (off, n, _) in m.layout.items() if off < m.num_trainable)
            segs = []
            for off, n, name in items:
                mt = re.search(r"\.layers\.(\d+)\.", name)
                layer = int(mt.group(1)) if mt and ".encoder." in name else -1
                d = hf_decays(name, m.config)
                if segs and segs[-1][2] == layer and segs[-1][3] == d and segs[-1][0] + segs[-1][1] == off:
                    segs[-1] = (segs[-1][0], segs[-1][1] + n, layer, d)
                else:
                    segs.append((off, n, layer, d))
            self._segments = segs
            self.layer_steps = [0] * (1 + max([s[2] for s in segs] + [-1]))
        return self._segments

    def step(self, grad_scale: float = 1.0, norm_done: bool = False, layer_keep=None):
        """Clip + update on the CURRENT stream.  ``norm_done``: gnorm_sq already holds the sum of squares (bucket partials).
        ``layer_keep``: this step's LayerDrop decisions (read only with ``skip_unused_layers``)."""
        m = self.model
        lr = self.current_lr()
        self.step_count += 1
        with torch.cuda.device(m.device):
            st = hip.stream()
            if not norm_done:
                self.add_sumsq(0, self.n, first=True)
            if self.skip_unused_layers:
                # one launch per (layer, decay group) segment: a skipped layer is left alone, a kept one is bias-corrected by
                # ITS step count (torch keeps `step` per parameter)
                segs = self._layer_segments()
                keep = [True] * len(self.layer_steps) if layer_keep is None else [bool(k) for k in layer_keep]
                for l, k in enumerate(keep):
                    self.layer_steps[l] += int(k)
                ranges = [(off, cnt, self.weight_decay if d else 0.0, self.step_count if layer < 0 else self.layer_steps[layer])
                          for off, cnt, layer, d in segs if layer < 0 or keep[layer]]
            else:
                ranges = [(off, cnt, wd, self.step_count) for off, cnt, wd in self._update_ranges()]
            for off, cnt, wd, stepno in ranges:
                hip.check(hip.lib.ssak_adamw_step(hip.ptr(m.params[off:]), hip.ptr(m.grads[off:]), hip.ptr(self.exp_avg[off:]),
                                                  hip.ptr(self.exp_avg_sq[off:]), hip.ptr(m.shadow[off:]), cnt,
                                                  hip.ptr(self.gnorm_sq), self.max_grad_norm, grad_scale, lr, self.betas[0],
                                                  self.betas[1], self.eps, wd, stepno, st))
        m.sync_weights(full=False)  # the weight-normed positional-conv layouts follow the updated (g, v)

    def grad_norm(self, grad_scale: float = 1.0) -> float:
        self.model.wait_params()
        return float(self.gnorm_sq.sqrt().item()) * grad_scale

    def state_dict(self):
        self.model.wait_params()
        sd = {"exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(), "step": self.step_count}
        if self.skip_unused_layers and self.layer_steps is not None:
            sd["layer_steps"] = list(self.layer_steps)
        return sd

    def load_state_dict(self, sd):
        self.model.wait_params()
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        if self.skip_unused_layers:
            self._layer_segments()
            self.layer_steps = list(sd.get("layer_steps", [self.step_count] * len(self.layer_steps)))


class _EventWork:
    """``wait()`` of an exchange launched through the C ABI: the current stream waits for the collective's event."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


def C_stream(s: torch.cuda.Stream):
    import ctypes
    return ctypes.c_void_p(s.cuda_stream)


class Trainer:
    """One optimizer step per call; data-parallel when a process group is initialised.

    Under data parallelism the optimizer tail runs on a SIDE STREAM (``optimizer_stream``; environment SSAK_OPT_STREAM=0/1): as each
    gradient bucket's all-reduce completes, its share of the clip norm is summed there (the norm partials ride with the
    buckets); after the last one the clip coefficient is known and AdamW sweeps the buffers, still on the side stream.  The
    compute stream meanwhile starts the next step -- waveform normalisation and the (frozen) conv feature encoder, a third of
    the forward, read no trainable parameter -- and waits for the update only at the feature projection
    (``ssak_w2v2_set_param_event``).  The exchange tail (last bucket) and the optimizer are hidden under that work."""

    def __init__(self, model: Wav2Vec2ForCTC, optimizer: AdamW, normalize_on_device: bool = True,
                 grad_exchange_dtype: str | None = None, optimizer_stream: bool | None = None, measure_stall: bool = False,
                 per_rank_seed: bool = True, exchange: str | None = None, comm=None):
        """``grad_exchange_dtype``: "fp32" (default; or environment SSAK_DP_GRAD_DTYPE) or "bf16" -- the gradient buckets are
        rounded to bf16 for the all-reduce and widened again before the clip + update: half the bytes over xGMI (180 MB
        instead of 361 MB per step for the base model) at bf16 rounding of the exchanged sums.
        ``exchange``: "torch" (default; or environment SSAK_DP_EXCHANGE) = torch.distributed.all_reduce (backend nccl = RCCL), or
        "c" = the library's own ``ssak_allreduce`` (include/ssak_hip.h: RCCL through the C ABI, the path a host without
        torch.distributed takes; torch.distributed then only carries the 128-byte communicator id).  ``comm``: a communicator object
        to use for exchange "c" instead of creating ``hip.Comm`` (``all_reduce(tensor, offset, count, stream_)`` asynchronous on the
        given stream, ``close()``): tests drive the event ordering of this path with a stand-in for RCCL."""
        self.model, self.opt = model, optimizer
        self.exchange = exchange or os.environ.get("SSAK_DP_EXCHANGE", "torch")
        if self.exchange not in ("torch", "c"):
            raise ValueError("exchange must be 'torch' or 'c'")
        self._comm = None
        self.grad_exchange_dtype = grad_exchange_dtype or os.environ.get("SSAK_DP_GRAD_DTYPE", "fp32")
        if self.grad_exchange_dtype not in ("fp32", "bf16"):
            raise ValueError("grad_exchange_dtype must be 'fp32' or 'bf16'")
        self._g16 = None
        self.dist = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.world = torch.distributed.get_world_size() if self.dist else 1
        self.normalize_on_device = normalize_on_device
        # group-norm ("base") models run without attention mask, layer-norm (XLSR) models with it
        self.use_mask = model.config.feat_extract_norm == "layer"
        self._fold_norm = hasattr(model, "can_fold_normalisation") and model.can_fold_normalisation() and os.environ.get("SSAK_FOLD_NORM", "1") != "0"
        if self.dist and self.world > 1 and per_rank_seed:
            # independent replicas draw independent regularisers: dropout masks, SpecAugment spans and LayerDrop decisions
            # come from seed + rank (identical seeds would apply one mask pattern to every shard of the global batch)
            model.reseed(model._seed + torch.distributed.get_rank())
        self._works = []  # (work, offset, count) of this step's bucket all-reduces, in announcement order
        self._bucket_ev = []  # per bucket: events around its wait on the optimizer stream (measure_stall)
        self.bucket_log = []  # [(offset, count)] of the last step (bench: bucket sizes)
        if optimizer_stream is None:
            # default: on under data parallelism (it hides the exchange tail + the update under the next step's conv stack),
            # off on one GPU, where there is no exchange to hide and the HBM-bound AdamW sweep only competes with the equally
            # HBM-bound conv0 of the next step (measured: 17.40 vs 17.35 ms in line, profiles/r02_ab_optimizer_stream.log; the
            # small conv0 statistics kernels also queue behind AdamW's 4096 workgroups).  SSAK_OPT_STREAM=0/1 overrides.
            env = os.environ.get("SSAK_OPT_STREAM")
            optimizer_stream = (env != "0") if env is not None else (self.dist and self.world > 1)
        self.opt_stream = None
        self._stall = None
        if optimizer_stream:
            with torch.cuda.device(model.device):
                self.opt_stream = torch.cuda.Stream()
                self._ready = torch.cuda.Event()
                if measure_stall:
                    self._stall = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    for ev in self._stall:
                        ev.record()  # (created lazily by torch: the engine needs the raw handles now)
                self._ready.record(self.opt_stream)
            model.set_param_event(self._ready, *(self._stall or (None, None)))
        if self.dist and self.world > 1:
            # RCCL's kernels take CUs away from the persistent GEMMs for a while: this handle's products draw their tiles
            # from tickets (include/ssak_hip.h: ssak_gemm_desc.dynamic_tiles)
            model.set_option(hip.W2V2_OPT_DYNAMIC_TILES, 1)
        if self.dist:
            # bucketed exchange: one async sum all-reduce per announced gradient range (a layer's matrices = 28 MB
            # for base), issued while the rest of the backward is still running; RCCL runs them on its own stream
            model.set_grad_ready_callback(self._on_grads_ready)
            if self.exchange == "c":
                with torch.cuda.device(model.device):
                    if comm is not None:
                        self._comm = comm
                    else:
                        uid = [hip.Comm.unique_id() if torch.distributed.get_rank() == 0 else None]
                        torch.distributed.broadcast_object_list(uid, src=0)
                        self._comm = hip.Comm(self.world, torch.distributed.get_rank(), uid[0])
                    self._xstream = torch.cuda.Stream()  # the collectives' own stream: they overlap the rest of the backward

    def drain_exchange(self):
        """Exchange "c" keeps a PRIVATE communicator whose collectives run on the trainer's own stream.  Two communicators with
        collectives in flight at once deadlock unless every rank launches them in the same order, so before anything goes
        through torch.distributed's communicator on this device (evaluation sums, broadcasts, barriers) the current stream --
        the one torch's collective is ordered behind -- waits for the exchange stream and for the optimizer tail that consumes
        it.  A no-op for exchange "torch"."""
        if self._comm is None:
            return
        with torch.cuda.device(self.model.device):
            cur = torch.cuda.current_stream()
            cur.wait_stream(self._xstream)
            if self.opt_stream is not None:
                cur.wait_stream(self.opt_stream)

    def close(self):
        """Destroys the private communicator of exchange "c" (ncclCommDestroy) after its stream has drained; idempotent."""
        if self._comm is not None:
            with torch.cuda.device(self.model.device):
                self._xstream.synchronize()
                if self.opt_stream is not None:
                    self.opt_stream.synchronize()
            self._comm.close()
            self._comm = None

    def _on_grads_ready(self, offset: int, count: int):
        m = self.model
        if self.grad_exchange_dtype == "bf16":
            if self._g16 is None:
                self._g16 = torch.empty(m.num_trainable, dtype=torch.bfloat16, device=m.device)
            # ranges start at multiples of 8 elements for every supported topology; the cast runs on the compute stream,
            # behind the kernels that produced the range
            hip.check(hip.lib.ssak_cast_f32_bf16(hip.ptr(m.grads[offset:]), hip.ptr(self._g16[offset:]), count, hip.stream()))
            buf = self._g16
        else:
            buf = m.grads
        if self._comm is not None:
            # ssak_allreduce on the exchange stream, behind the kernels that produced the range; `work.wait()` = the waiting
            # stream waits for the collective's event (what torch's async work object does)
            with torch.cuda.device(m.device):
                ready = torch.cuda.Event()
                ready.record()
                self._xstream.wait_event(ready)
                self._comm.all_reduce(buf, offset, count, stream_=C_stream(self._xstream))
                done = torch.cuda.Event()
                done.record(self._xstream)
            work = _EventWork(done)
        else:
            work = torch.distributed.all_reduce(buf[offset:offset + count], async_op=True)
        self._works.append((work, offset, count))

    def broadcast_parameters(self):
        if self.dist:
            self.drain_exchange()
            self.model.wait_params()
            torch.distributed.broadcast(self.model.params, src=0)
            self.model.sync_weights(full=True)

    def stall_ms(self) -> float:
        """Exposed part of the previous step's exchange + optimizer tail: how long the last forward's compute stream sat at
        the parameter-ready wait (needs ``measure_stall=True``; synchronises)."""
        if self._stall is None:
            return float("nan")
        self._stall[1].synchronize()
        return self._stall[0].elapsed_time(self._stall[1])

    def bucket_wait_us(self):
        """Per gradient bucket of the LAST step, in announcement order: microseconds the optimizer stream spent waiting for the
        bucket's all-reduce (needs ``measure_stall=True`` and the side stream; synchronises).  The first entries absorb the
        backward still running, the last one is the exposed exchange tail.  None without a process group."""
        n = len(self.bucket_log)
        if self._stall is None or self.opt_stream is None or n == 0 or len(self._bucket_ev) < n:
            return None
        self._bucket_ev[n - 1][1].synchronize()
        return [round(1e3 * self._bucket_ev[i][0].elapsed_time(self._bucket_ev[i][1]), 1) for i in range(n)]

    def _tail(self, norm_from_buckets: bool, compute_stream=None):
        """Norm partials per reduced bucket, clip + AdamW, positional-conv layouts: on the current stream."""
        m = self.model
        timed = self._stall is not None and self.opt_stream is not None
        for i, (work, off, cnt) in enumerate(self._works):
            if timed:  # how long the optimizer stream sits at this bucket's collective (bench.py: optimizer_tail.bucket_wait_us)
                while len(self._bucket_ev) <= i:
                    self._bucket_ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
                self._bucket_ev[i][0].record()
            work.wait()  # the current stream waits for this bucket's collective; the mean over ranks is folded into the optimizer
            if timed:
                self._bucket_ev[i][1].record()
            if self.grad_exchange_dtype == "bf16":
                hip.check(hip.lib.ssak_cast_bf16_f32(hip.ptr(self._g16[off:]), hip.ptr(m.grads[off:]), cnt, hip.stream()))
            if norm_from_buckets:
                self.opt.add_sumsq(off, cnt, first=(i == 0))
        self.bucket_log = [(off, cnt) for _, off, cnt in self._works]
        have_norm = norm_from_buckets and bool(self._works)
        self._works.clear()
        if compute_stream is not None:
            # everything of the backward (without a process group: the gradients themselves) precedes the update
            torch.cuda.current_stream().wait_stream(compute_stream)
        # (per-layer skipping follows ONE process's LayerDrop decisions: under a process group a layer another rank ran has a gradient)
        self.opt.step(grad_scale=1.0 / self.world, norm_done=have_norm, layer_keep=self.model.last_layer_keep if self.world == 1 else None)

    def train_step(self, waves: torch.Tensor, lengths, labels: torch.Tensor, raw: bool = True, global_count: int | None = None,
                   fold_norm: bool | None = None):
        """waves [B,T] fp32 on the device (raw samples when ``raw``: normalised here, a1), lengths [B] or None,
        labels [B,L] (-100 padding).  Returns the (local) loss tensor; no host synchronisation.

        ``fold_norm``: None = the trainer's default (SSAK_FOLD_NORM, on when the model allows it); False = keep the separate
        ssak_wave_normalize pass for this call.  The fold applies only to un-padded batches (``lengths`` None: every utterance
        fills the batch's T) of the group-norm topology with a frozen feature encoder; everything else normalises in its own pass.

        ``global_count`` (data parallel): utterances of the GLOBAL batch when the ranks' shards differ in size (the short last
        batch of an epoch, data.shard_batch).  The CTC loss is a mean over utterances, so rank r's gradient enters the sum
        all-reduce weighted by n_r * world / global_count (the optimizer divides the sum by world): the update equals the
        single-process update on the whole batch.  A rank whose shard is empty (``waves`` None or
        zero rows) contributes zeros to the same sequence of collectives."""
        m = self.model
        n_local = 0 if waves is None else int(waves.shape[0])
        scale = 1.0
        if global_count is not None and self.world > 1:
            scale = n_local * self.world / float(global_count)
        if n_local == 0:
            if not (self.dist and self.world > 1):
                raise ValueError("train_step: empty batch")
            loss = self._empty_step()
        else:
            # raw full-length utterances into the group-norm model: the normalisation (a1) rides in conv0's GroupNorm statistics,
            # no pass of its own (model.can_fold_normalisation; ragged batches -- lengths given -- and the layer-norm topology
            # keep ssak_wave_normalize)
            fold = raw and lengths is None and self._fold_norm and fold_norm is not False
            x = waves if (fold or not raw) else hip.wave_normalize(waves, lengths)
            out = m(x, lengths=lengths if self.use_mask else None, labels=labels, **({"raw_input": True} if fold else {}))
            m.backward(grad_scale=scale)  # with a process group: announces finished gradient ranges -> bucketed all-reduces overlap it
            loss = out.loss
        if self.opt_stream is None:
            self._tail(norm_from_buckets=False)
            return loss
        with torch.cuda.device(m.device):
            cur = torch.cuda.current_stream()
            with torch.cuda.stream(self.opt_stream):
                self._tail(norm_from_buckets=True, compute_stream=cur)
                self._ready.record(self.opt_stream)
        return loss

    def _empty_step(self):
        """This rank has no utterance of the global batch: zero gradients through the same bucket sequence the engine's
        backward announces (ssak_w2v2_grad_ranges lists it from the configuration alone), so every rank still pairs the k-th
        collective with the same range."""
        m = self.model
        m.grads[:m.num_trainable].zero_()
        for off, cnt in m.announced_grad_ranges():
            self._on_grads_ready(off, cnt)
        return torch.zeros(1, dtype=torch.float32, device=m.device)  # (the shape of CTCOutput.loss)
