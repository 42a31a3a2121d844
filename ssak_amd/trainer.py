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


class AdamW:
    """Flat-buffer AdamW with fused global-norm clipping (kernels: ssak_grad_sumsq / ssak_adamw_step)."""

    def __init__(self, model: Wav2Vec2ForCTC, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 max_grad_norm=1.0, warmup_steps=500, total_steps=100000):
        self.model = model
        self.n = model.num_trainable
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm, self.warmup_steps, self.total_steps = max_grad_norm, warmup_steps, total_steps
        self.exp_avg = torch.zeros(self.n, dtype=torch.float32, device=model.device)
        self.exp_avg_sq = torch.zeros(self.n, dtype=torch.float32, device=model.device)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=model.device)
        self._sumsq_ws = torch.empty(1024, dtype=torch.float32, device=model.device)  # per-workgroup partials of the norm
        self.step_count = 0

    def current_lr(self) -> float:
        return linear_warmup_lr(self.lr, self.step_count, self.warmup_steps, self.total_steps)

    def step(self, grad_scale: float = 1.0):
        m = self.model
        lr = self.current_lr()
        self.step_count += 1
        with torch.cuda.device(m.device):
            st = hip.stream()
            hip.check(hip.lib.ssak_grad_sumsq(hip.ptr(m.grads), self.n, hip.ptr(self.gnorm_sq), hip.ptr(self._sumsq_ws),
                                              self._sumsq_ws.numel() * 4, st))
            hip.check(hip.lib.ssak_adamw_step(hip.ptr(m.params), hip.ptr(m.grads), hip.ptr(self.exp_avg),
                                              hip.ptr(self.exp_avg_sq), hip.ptr(m.shadow), self.n, hip.ptr(self.gnorm_sq),
                                              self.max_grad_norm, grad_scale, lr, self.betas[0], self.betas[1], self.eps,
                                              self.weight_decay, self.step_count, st))
        m.sync_weights(full=False)  # the weight-normed positional-conv layouts follow the updated (g, v)

    def grad_norm(self, grad_scale: float = 1.0) -> float:
        return float(self.gnorm_sq.sqrt().item()) * grad_scale

    def state_dict(self):
        return {"exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(), "step": self.step_count}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])


class Trainer:
    """One optimizer step per call; data-parallel when a process group is initialised."""

    def __init__(self, model: Wav2Vec2ForCTC, optimizer: AdamW, normalize_on_device: bool = True,
                 grad_exchange_dtype: str | None = None):
        """``grad_exchange_dtype``: "fp32" (default; or environment SSAK_DP_GRAD_DTYPE) or "bf16" -- the gradient buckets are
        rounded to bf16 for the all-reduce and widened again before the clip + update: half the bytes over xGMI (180 MB
        instead of 361 MB per step for the base model) at bf16 rounding of the exchanged sums."""
        self.model, self.opt = model, optimizer
        self.grad_exchange_dtype = grad_exchange_dtype or os.environ.get("SSAK_DP_GRAD_DTYPE", "fp32")
        if self.grad_exchange_dtype not in ("fp32", "bf16"):
            raise ValueError("grad_exchange_dtype must be 'fp32' or 'bf16'")
        self._g16 = None
        self.dist = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.world = torch.distributed.get_world_size() if self.dist else 1
        self.normalize_on_device = normalize_on_device
        # group-norm ("base") models run without attention mask, layer-norm (XLSR) models with it
        self.use_mask = model.config.feat_extract_norm == "layer"
        self._works = []
        if self.dist and self.world > 1:
            # RCCL's kernels take CUs away from the persistent GEMMs for a while: draw tiles from tickets (include/ssak_hip.h)
            hip.check(hip.lib.ssak_gemm_tile_order(1))
        if self.dist:
            # bucketed exchange: one async sum all-reduce per announced gradient range (a layer's matrices = 28 MB
            # for base), issued while the rest of the backward is still running; RCCL runs them on its own stream
            model.set_grad_ready_callback(self._on_grads_ready)

    def _on_grads_ready(self, offset: int, count: int):
        m = self.model
        if self.grad_exchange_dtype == "bf16":
            if self._g16 is None:
                self._g16 = torch.empty(m.num_trainable, dtype=torch.bfloat16, device=m.device)
            # ranges start at multiples of 8 elements for every supported topology; the cast runs on the compute stream,
            # behind the kernels that produced the range
            hip.check(hip.lib.ssak_cast_f32_bf16(hip.ptr(m.grads[offset:]), hip.ptr(self._g16[offset:]), count, hip.stream()))
            self._works.append(torch.distributed.all_reduce(self._g16[offset:offset + count], async_op=True))
            return
        self._works.append(torch.distributed.all_reduce(m.grads[offset:offset + count], async_op=True))

    def broadcast_parameters(self):
        if self.dist:
            torch.distributed.broadcast(self.model.params, src=0)
            self.model.sync_weights(full=True)

    def train_step(self, waves: torch.Tensor, lengths, labels: torch.Tensor, raw: bool = True):
        """waves [B,T] fp32 on the device (raw samples when ``raw``: normalised here, a1), lengths [B] or None,
        labels [B,L] (-100 padding).  Returns the (local) loss tensor; no host synchronisation."""
        m = self.model
        x = hip.wave_normalize(waves, lengths) if raw else waves
        out = m(x, lengths=lengths if self.use_mask else None, labels=labels)
        m.backward()  # with a process group: announces finished gradient ranges -> bucketed all-reduces overlap it
        for w in self._works:
            w.wait()  # the compute stream waits for the reduced buckets; mean over ranks is folded into the optimizer
        if self._works and self.grad_exchange_dtype == "bf16":
            hip.check(hip.lib.ssak_cast_bf16_f32(hip.ptr(self._g16), hip.ptr(m.grads), m.num_trainable, hip.stream()))
        self._works.clear()
        self.opt.step(grad_scale=1.0 / self.world)
        return out.loss
