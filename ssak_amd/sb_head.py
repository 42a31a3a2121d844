"""The SpeechBrain fine-tuning recipe on the device: counterpart of ``ssak/train/speechbrain/wav2vec_train.py``.

What the recipe's ``Trainer(sb.core.Brain)`` does per batch (:39-137) with the modules of
``ssak/train/speechbrain/fr/hyperparameters_wav2vec_finetune_cv-fr.yaml`` (:87-137)::

    feats  = modules.wav2vec2(wavs)              # layer_norm(wav) -> Wav2Vec2Model -> layer_norm(out)   (frozen by default)
    x      = modules.enc(feats)                  # 3 x [Linear(1024) -> BatchNorm1d -> LeakyReLU -> Dropout(0.15)]
    logits = modules.ctc_lin(x)                  # Linear(76)
    loss   = ctc_cost(log_softmax(logits), tokens, wav_lens, tokens_lens)
    loss.backward(); clip_grad_norm_(all, 5.0); Adadelta(lr 1.0, rho 0.95, eps 1e-8).step() [+ Adam(lr 1e-4) on wav2vec2]
    after each validation: NewBob annealing of both learning rates on the validation loss (:181-194)

Here every tensor operation is a kernel of ``libssak_hip.so``: the Linears are ``ssak_gemm_bf16`` (bf16 operands, fp32
accumulation, fp32 master weights), BatchNorm + LeakyReLU + dropout one fused pass (``ssak_batchnorm_act_*``), the two
``F.layer_norm(x, x.shape[1:])`` of the wav2vec2 wrapper ``ssak_utt_norm_*``, log-softmax + CTC ``ssak_ctc_loss_fwd_bwd``,
the optimizers ``ssak_adadelta_step`` / ``ssak_adamw_step`` behind one joint clip coefficient.  torch supplies buffers, the
stream and the process group.  speechbrain itself is not installed in this image: the module semantics restated here are
those of torch.nn.Linear / BatchNorm1d / LeakyReLU / Dropout / F.ctc_loss / torch.optim.Adadelta, which speechbrain wraps
(oracle/sb_head_ref.py states them in plain torch and the tests compare against it).

Out of scope (host-side data augmentation and bookkeeping of the recipe): ``TimeDomainSpecAugment`` (yaml :82-85), the
checkpointer / train loggers (:139-152), the dataio pipeline (:300-494).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import hip
from .model import Wav2Vec2ForCTC

TRAIN, VALID, TEST = "train", "valid", "test"  # sb.Stage


class NewBobScheduler:
    """Anneal a value when the relative improvement of the tracked metric falls below a threshold
    (speechbrain.nnet.schedulers.NewBobScheduler as configured at yaml :124-135; called at wav2vec_train.py:182-187)."""

    def __init__(self, initial_value: float, annealing_factor: float = 0.5, improvement_threshold: float = 0.0025,
                 patient: int = 0):
        self.hyperparam_value = initial_value
        self.annealing_factor = annealing_factor
        self.improvement_threshold = improvement_threshold
        self.patient = patient
        self.metric_values = []
        self.current_patient = patient

    def __call__(self, metric_value: float):
        """-> (value used so far, value to use from now on)"""
        old = new = self.hyperparam_value
        if self.metric_values:
            prev = self.metric_values[-1]
            improvement = 0.0 if prev == 0 else (prev - metric_value) / prev
            if improvement < self.improvement_threshold:
                if self.current_patient == 0:
                    new *= self.annealing_factor
                    self.current_patient = self.patient
                else:
                    self.current_patient -= 1
        self.metric_values.append(metric_value)
        self.hyperparam_value = new
        return old, new

    def state_dict(self):
        return {"hyperparam_value": self.hyperparam_value, "metric_values": list(self.metric_values),
                "current_patient": self.current_patient}

    def load_state_dict(self, sd):
        self.hyperparam_value = sd["hyperparam_value"]
        self.metric_values = list(sd["metric_values"])
        self.current_patient = sd["current_patient"]


class CTCHead:
    """``enc`` + ``ctc_lin`` of the recipe (yaml :87-111) on flat device buffers.

    Parameters (fp32 master, fp32 gradients, bf16 shadow for the GEMM operands) are laid out in the order of
    ``hparams.model = ModuleList([enc, ctc_lin])`` with its state_dict names (``0.linear1.w.weight`` ...,
    ``1.w.weight``); BatchNorm running statistics are separate buffers (not seen by the optimizer)."""

    def __init__(self, input_dim: int = 1024, dnn_neurons: int = 1024, output_neurons: int = 76,
                 dropouts: Sequence[float] = (0.15, 0.15, 0.0), leaky_slope: float = 0.01, bn_eps: float = 1e-5,
                 bn_momentum: float = 0.1, device: str = "cuda:0", seed: int = 1234):
        if not torch.cuda.is_available():
            raise RuntimeError("ssak_amd needs an MI355X: there is no CPU fallback for the acoustic head")
        if input_dim % 8 or dnn_neurons % 8:
            raise ValueError("input_dim and dnn_neurons must be multiples of 8")
        self.device = torch.device(device)
        self.input_dim, self.D, self.V = input_dim, dnn_neurons, output_neurons
        self.Vp = (output_neurons + 7) // 8 * 8  # inert padding classes (bias -1e4), as for the engine's lm_head
        self.dropouts = tuple(float(p) for p in dropouts)
        self.nblk = len(self.dropouts)
        self.slope, self.bn_eps, self.bn_momentum = leaky_slope, bn_eps, bn_momentum
        self.training = True
        self.sync_bn = False  # set by Brain under data parallelism: BatchNorm statistics over the global batch
        self.dynamic_tiles = False  # set by Brain under data parallelism: ticket tile order of the persistent GEMMs
        self.layout: Dict[str, tuple] = {}
        cur = 0

        def add(name, shape):
            nonlocal cur
            n = int(np.prod(shape))
            self.layout[name] = (cur, n, tuple(shape))
            cur += (n + 7) // 8 * 8

        din = input_dim
        for i in range(1, self.nblk + 1):
            add(f"0.linear{i}.w.weight", (dnn_neurons, din))
            add(f"0.linear{i}.w.bias", (dnn_neurons,))
            add(f"0.bn{i}.norm.weight", (dnn_neurons,))
            add(f"0.bn{i}.norm.bias", (dnn_neurons,))
            din = dnn_neurons
        add("1.w.weight", (self.Vp, dnn_neurons))
        add("1.w.bias", (self.Vp,))
        self.num_params = cur
        with torch.cuda.device(self.device):
            self.params = torch.zeros(cur, dtype=torch.float32, device=self.device)
            self.grads = torch.zeros(cur, dtype=torch.float32, device=self.device)
            self.shadow = torch.zeros(cur, dtype=torch.bfloat16, device=self.device)
            self.running_mean = [torch.zeros(dnn_neurons, dtype=torch.float32, device=self.device) for _ in range(self.nblk)]
            self.running_var = [torch.ones(dnn_neurons, dtype=torch.float32, device=self.device) for _ in range(self.nblk)]
            self.num_batches_tracked = 0
            self._bn_ws = torch.empty(hip.lib.ssak_batchnorm_workspace_bytes(dnn_neurons), dtype=torch.uint8, device=self.device)
            self._cs_ws = torch.empty(hip.lib.ssak_colsum_workspace_bytes(max(dnn_neurons, self.Vp)), dtype=torch.uint8,
                                      device=self.device)
        # torch.nn.Linear's default initialisation (U(+-1/sqrt(fan_in)) for weight and bias), BatchNorm weight 1 / bias 0
        g = torch.Generator().manual_seed(seed)
        sd = {}
        din = input_dim
        for i in range(1, self.nblk + 1):
            b = 1.0 / math.sqrt(din)
            sd[f"0.linear{i}.w.weight"] = (torch.rand(dnn_neurons, din, generator=g) * 2 - 1) * b
            sd[f"0.linear{i}.w.bias"] = (torch.rand(dnn_neurons, generator=g) * 2 - 1) * b
            sd[f"0.bn{i}.norm.weight"] = torch.ones(dnn_neurons)
            sd[f"0.bn{i}.norm.bias"] = torch.zeros(dnn_neurons)
            din = dnn_neurons
        b = 1.0 / math.sqrt(dnn_neurons)
        sd["1.w.weight"] = (torch.rand(output_neurons, dnn_neurons, generator=g) * 2 - 1) * b
        sd["1.w.bias"] = (torch.rand(output_neurons, generator=g) * 2 - 1) * b
        self.load_state_dict(sd)
        self._seed = int(np.random.SeedSequence(seed).generate_state(1, dtype=np.uint64)[0])
        self._saved = None

    # ------------------------------------------------------------------ parameters
    def param(self, name):
        off, n, shape = self.layout[name]
        return self.params[off:off + n].view(shape)

    def grad(self, name):
        off, n, shape = self.layout[name]
        return self.grads[off:off + n].view(shape)

    def _shadow(self, name):
        off, n, shape = self.layout[name]
        return self.shadow[off:off + n].view(shape)

    def state_dict(self):
        sd = {}
        for n in self.layout:
            t = self.param(n).detach().cpu().clone()
            sd[n] = t[:self.V] if n.startswith("1.w.") else t
        for i in range(self.nblk):
            sd[f"0.bn{i + 1}.norm.running_mean"] = self.running_mean[i].cpu().clone()
            sd[f"0.bn{i + 1}.norm.running_var"] = self.running_var[i].cpu().clone()
            sd[f"0.bn{i + 1}.norm.num_batches_tracked"] = torch.tensor(self.num_batches_tracked)
        return sd

    def load_state_dict(self, sd, strict: bool = True):
        missing = [n for n in self.layout if n not in sd]
        if strict and missing:
            raise RuntimeError(f"state_dict mismatch: missing {missing[:4]}")
        for n, (off, numel, shape) in self.layout.items():
            if n not in sd:
                continue
            t = torch.as_tensor(sd[n]).to(torch.float32)
            if n.startswith("1.w.") and t.shape[0] == self.V and self.Vp != self.V:
                pad = (self.Vp - self.V,) + tuple(t.shape[1:])
                t = torch.cat([t, torch.full(pad, 0.0 if n.endswith("weight") else -1.0e4)], 0)
            if tuple(t.shape) != shape:
                raise RuntimeError(f"size mismatch for {n}: {tuple(t.shape)} vs {shape}")
            self.params[off:off + numel].copy_(t.reshape(-1).to(self.device))
        for i in range(self.nblk):
            k = f"0.bn{i + 1}.norm."
            if k + "running_mean" in sd:
                self.running_mean[i].copy_(torch.as_tensor(sd[k + "running_mean"]).to(self.device))
                self.running_var[i].copy_(torch.as_tensor(sd[k + "running_var"]).to(self.device))
                self.num_batches_tracked = int(sd.get(k + "num_batches_tracked", 0))
        self.sync_weights()
        return self

    def sync_weights(self):
        """bf16 operand copy of the fp32 master (after loading; the optimizer kernel keeps it current afterwards)."""
        with torch.cuda.device(self.device):
            hip.check(hip.lib.ssak_cast_f32_bf16(hip.ptr(self.params), hip.ptr(self.shadow), self.num_params, hip.stream()))

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    # ------------------------------------------------------------------ forward / backward
    def forward(self, feats: torch.Tensor) -> torch.Tensor:
        """feats [B, F, input_dim] bf16 -> logits [B, F, output_neurons] fp32 (a view of the padded buffer)."""
        assert feats.is_cuda and feats.dtype == torch.bfloat16 and feats.is_contiguous() and feats.shape[-1] == self.input_dim
        B, F, _ = feats.shape
        M, D = B * F, self.D
        tr = self.training
        self._seed = (self._seed * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        h = feats.view(M, self.input_dim)
        saved = []
        with torch.cuda.device(self.device):
            st = hip.stream()
            for i in range(self.nblk):
                din = h.shape[1]
                a = torch.empty((M, D), dtype=torch.bfloat16, device=self.device)
                hip.gemm(h, self._shadow(f"0.linear{i + 1}.w.weight"), a, M, D, din, lda=din, ldb=din, ldc=D,
                         bias=self.param(f"0.linear{i + 1}.w.bias"), dynamic_tiles=self.dynamic_tiles)
                y = torch.empty_like(a)
                mean = torch.empty(D, dtype=torch.float32, device=self.device)
                rstd = torch.empty(D, dtype=torch.float32, device=self.device)
                gs = None
                if tr and self.sync_bn:
                    gs = torch.empty(2 * D + 1, dtype=torch.float64, device=self.device)
                    hip.check(hip.lib.ssak_batchnorm_stats(hip.ptr(a), M, D, hip.ptr(gs), hip.ptr(self._bn_ws), self._bn_ws.numel(), st))
                    torch.distributed.all_reduce(gs)
                hip.check(hip.lib.ssak_batchnorm_act_fwd(
                    hip.ptr(a), hip.ptr(y), M, D, hip.ptr(self.param(f"0.bn{i + 1}.norm.weight")),
                    hip.ptr(self.param(f"0.bn{i + 1}.norm.bias")), hip.ptr(self.running_mean[i]), hip.ptr(self.running_var[i]),
                    self.bn_momentum, self.bn_eps, int(tr), self.slope, self.dropouts[i], hip.C.c_uint64(self._seed), i,
                    hip.ptr(mean), hip.ptr(rstd), hip.ptr(gs), hip.ptr(self._bn_ws), self._bn_ws.numel(), st))
                saved.append((h, a, mean, rstd))
                h = y
            logits = torch.empty((B, F, self.Vp), dtype=torch.float32, device=self.device)
            hip.gemm(h, self._shadow("1.w.weight"), logits, M, self.Vp, D, lda=D, ldb=D, ldc=self.Vp, bias=self.param("1.w.bias"), dynamic_tiles=self.dynamic_tiles)
        if tr:
            self.num_batches_tracked += 1
        self._saved = (saved, h, self._seed, (B, F)) if tr else None
        return logits

    __call__ = forward

    def backward(self, dlogits: torch.Tensor, need_input_grad: bool = False) -> Optional[torch.Tensor]:
        """dlogits [B, F, Vp] fp32 (from ``hip.ctc_loss``) -> gradients of every head parameter in ``self.grads``;
        returns d loss / d feats [B, F, input_dim] bf16 when ``need_input_grad`` (the unfrozen wav2vec2)."""
        if self._saved is None:
            raise RuntimeError("backward() needs a training-mode forward")
        saved, h_last, seed, (B, F) = self._saved
        M, D, Vp = B * F, self.D, self.Vp
        assert dlogits.shape == (B, F, Vp) and dlogits.dtype == torch.float32 and dlogits.is_contiguous()
        dev = self.device
        with torch.cuda.device(dev):
            st = hip.stream()

            def colsum(X, N, out):
                hip.check(hip.lib.ssak_colsum_bf16(hip.ptr(X), N, M, N, hip.ptr(out), hip.ptr(self._cs_ws), self._cs_ws.numel(), st))

            d = torch.empty((M, Vp), dtype=torch.bfloat16, device=dev)
            hip.check(hip.lib.ssak_cast_f32_bf16(hip.ptr(dlogits), hip.ptr(d), M * Vp, st))
            # ctc_lin: dW = d^T h, db = column sums of d, dh = d W
            hip.gemm(d, h_last, self.grad("1.w.weight"), Vp, D, M, a_kmajor=True, b_kmajor=True, lda=Vp, ldb=D, ldc=D, split_k=0, dynamic_tiles=self.dynamic_tiles)
            colsum(d, Vp, self.grad("1.w.bias"))
            dh = torch.empty((M, D), dtype=torch.bfloat16, device=dev)
            hip.gemm(d, self._shadow("1.w.weight"), dh, M, D, Vp, lda=Vp, b_kmajor=True, ldb=D, ldc=D, dynamic_tiles=self.dynamic_tiles)
            for i in reversed(range(self.nblk)):
                h_in, a, mean, rstd = saved[i]
                din = h_in.shape[1]
                da = torch.empty((M, D), dtype=torch.bfloat16, device=dev)
                bn_args = (M, D, hip.ptr(self.param(f"0.bn{i + 1}.norm.weight")), hip.ptr(self.param(f"0.bn{i + 1}.norm.bias")),
                           hip.ptr(mean), hip.ptr(rstd), self.slope, self.dropouts[i], hip.C.c_uint64(seed), i,
                           hip.ptr(self.grad(f"0.bn{i + 1}.norm.weight")), hip.ptr(self.grad(f"0.bn{i + 1}.norm.bias")))
                if self.sync_bn:  # local parameter gradients + local totals, all-reduce, then dx from the global totals
                    gs = torch.empty(2 * D + 1, dtype=torch.float64, device=dev)
                    hip.check(hip.lib.ssak_batchnorm_act_bwd(hip.ptr(dh), hip.ptr(a), None, *bn_args, hip.ptr(gs), None,
                                                             hip.ptr(self._bn_ws), self._bn_ws.numel(), st))
                    torch.distributed.all_reduce(gs)
                    hip.check(hip.lib.ssak_batchnorm_act_bwd(hip.ptr(dh), hip.ptr(a), hip.ptr(da), *bn_args, None, hip.ptr(gs),
                                                             hip.ptr(self._bn_ws), self._bn_ws.numel(), st))
                else:
                    hip.check(hip.lib.ssak_batchnorm_act_bwd(hip.ptr(dh), hip.ptr(a), hip.ptr(da), *bn_args, None, None,
                                                             hip.ptr(self._bn_ws), self._bn_ws.numel(), st))
                hip.gemm(da, h_in, self.grad(f"0.linear{i + 1}.w.weight"), D, din, M, a_kmajor=True, b_kmajor=True, lda=D, ldb=din,
                         ldc=din, split_k=0, dynamic_tiles=self.dynamic_tiles)
                colsum(da, D, self.grad(f"0.linear{i + 1}.w.bias"))
                if i > 0 or need_input_grad:
                    dh = torch.empty((M, din), dtype=torch.bfloat16, device=dev)
                    hip.gemm(da, self._shadow(f"0.linear{i + 1}.w.weight"), dh, M, din, D, lda=D, b_kmajor=True, ldb=din, ldc=din, dynamic_tiles=self.dynamic_tiles)
                else:
                    dh = None
        self._saved = None
        return None if dh is None else dh.view(B, F, self.input_dim)


class Adadelta:
    """torch.optim.Adadelta on the head's flat buffers (yaml :119-122), clip coefficient read on the device."""

    def __init__(self, head: CTCHead, lr: float = 1.0, rho: float = 0.95, eps: float = 1e-8, weight_decay: float = 0.0):
        self.head, self.lr, self.rho, self.eps, self.weight_decay = head, lr, rho, eps, weight_decay
        self.square_avg = torch.zeros_like(head.params)
        self.acc_delta = torch.zeros_like(head.params)

    def step(self, gnorm_sq: Optional[torch.Tensor] = None, max_norm: float = 0.0, grad_scale: float = 1.0):
        h = self.head
        with torch.cuda.device(h.device):
            hip.check(hip.lib.ssak_adadelta_step(hip.ptr(h.params), hip.ptr(h.grads), hip.ptr(self.square_avg),
                                                 hip.ptr(self.acc_delta), hip.ptr(h.shadow), h.num_params, hip.ptr(gnorm_sq),
                                                 max_norm, grad_scale, self.lr, self.rho, self.eps, self.weight_decay, hip.stream()))

    def state_dict(self):
        return {"square_avg": self.square_avg.cpu(), "acc_delta": self.acc_delta.cpu(), "lr": self.lr}

    def load_state_dict(self, sd):
        self.square_avg.copy_(sd["square_avg"])
        self.acc_delta.copy_(sd["acc_delta"])
        self.lr = float(sd["lr"])


class Brain:
    """fit_batch / evaluate_batch / on_stage_end of the recipe's ``Trainer`` (wav2vec_train.py:38-214).

    ``wav2vec2`` is the engine-backed model used through its hidden-state entry points; ``freeze_wav2vec`` (yaml :29,
    default True) runs it without gradient and in evaluation mode.  ``normalize_wav`` / ``output_norm`` are the two
    ``F.layer_norm(x, x.shape[1:])`` of speechbrain's HuggingFaceWav2Vec2 wrapper (eps 1e-5, no affine).
    Lengths are relative (fraction of the padded length), as everywhere in speechbrain."""

    def __init__(self, wav2vec2: Wav2Vec2ForCTC, head: CTCHead, freeze_wav2vec: bool = True, normalize_wav: bool = True,
                 output_norm: bool = True, lr: float = 1.0, lr_wav2vec: float = 1e-4, max_grad_norm: float = 5.0,
                 blank_index: int = 0, annealing=(0.8, 0.9), improvement_threshold: float = 0.0025, vocab=None,
                 sync_batchnorm: bool = True):
        self.wav2vec2, self.head = wav2vec2, head
        self.device = head.device
        self.freeze = freeze_wav2vec
        self.normalize_wav, self.output_norm = normalize_wav, output_norm
        self.max_grad_norm, self.blank_index = max_grad_norm, blank_index
        self.model_optimizer = Adadelta(head, lr=lr)
        self.wav2vec_optimizer = None
        if not freeze_wav2vec:
            from .trainer import AdamW
            # torch.optim.Adam(lr) (yaml :122-123): no weight decay, constant lr between annealings; the joint clip is done here
            self.wav2vec_optimizer = AdamW(wav2vec2, lr=lr_wav2vec, weight_decay=0.0, max_grad_norm=0.0, warmup_steps=0,
                                           total_steps=1 << 62)
        self.lr_annealing_model = NewBobScheduler(lr, annealing[0], improvement_threshold, 0)
        self.lr_annealing_wav2vec = NewBobScheduler(lr_wav2vec, annealing[1], improvement_threshold, 0)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._sumsq_ws = torch.empty(1024, dtype=torch.float32, device=self.device)
        self._un_ws = None
        self.dist = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.world = torch.distributed.get_world_size() if self.dist else 1
        self._works = []
        if self.dist and self.world > 1:
            # collectives share the chip with the persistent GEMMs: ticket tile order for the encoder's and the head's products
            wav2vec2.set_option(hip.W2V2_OPT_DYNAMIC_TILES, 1)
            head.dynamic_tiles = True
        # BatchNorm over the global batch under data parallelism; False = per-rank statistics, what the
        # reference's nn.DataParallel / DDP without SyncBatchNorm computes
        head.sync_bn = bool(self.dist and sync_batchnorm)
        if self.dist:  # per-rank dropout streams (replicas start from identical seeds)
            head._seed = (head._seed ^ (0x9E3779B97F4A7C15 * (torch.distributed.get_rank() + 1))) % (1 << 64)
        if self.dist and not freeze_wav2vec:
            wav2vec2.set_grad_ready_callback(
                lambda off, cnt: self._works.append(torch.distributed.all_reduce(wav2vec2.grads[off:off + cnt], async_op=True)))
        self.optimizer_step = 0
        self.vocab = vocab  # list of output symbols: enables the device WER of the validation stage (:66-93)
        self._wer = None
        self._fwd = None

    # ------------------------------------------------------------------ pieces
    def _utt_norm(self, x: torch.Tensor, want_stats: bool):
        B = x.shape[0]
        n = x[0].numel()
        need = hip.lib.ssak_utt_norm_workspace_bytes(B)
        if self._un_ws is None or self._un_ws.numel() < need:
            self._un_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        y = torch.empty_like(x)
        stats = torch.empty((B, 2), dtype=torch.float32, device=self.device) if want_stats else None
        hip.check(hip.lib.ssak_utt_norm_fwd(hip.ptr(x), hip.ptr(y), B, n, int(x.dtype == torch.bfloat16), 1e-5, hip.ptr(stats),
                                            hip.ptr(self._un_ws), self._un_ws.numel(), hip.stream()))
        return y, stats

    def extract_features(self, wavs: torch.Tensor, training: bool):
        """``modules.wav2vec2(wavs)``: [B, T] fp32 -> [B, F, H] bf16."""
        w = self.wav2vec2
        grad = training and not self.freeze
        wavs = wavs.to(device=self.device, dtype=torch.float32).contiguous()
        with torch.cuda.device(self.device):
            if self.normalize_wav:
                wavs, _ = self._utt_norm(wavs, False)
            w.train(grad)
            hidden, _ = w.forward_hidden(wavs)  # no attention mask: the wrapper calls self.model(wav)[0]
            stats = None
            if self.output_norm:
                hidden, stats = self._utt_norm(hidden, grad)
        return hidden, stats

    def compute_forward(self, wavs, wav_lens, stage=TRAIN):
        """-> (logits [B, F, V] fp32, wav_lens); ``log_softmax`` of the recipe's p_ctc is fused into the CTC kernel."""
        training = stage == TRAIN
        self.head.train(training)
        feats, stats = self.extract_features(wavs, training)
        logits = self.head(feats)
        self._fwd = (feats, stats) if training else None
        return logits, torch.as_tensor(wav_lens, dtype=torch.float32)

    @staticmethod
    def _abs_lens(rel, n):
        return torch.round(torch.as_tensor(rel, dtype=torch.float32).cpu() * n).to(torch.int32)

    def compute_objectives(self, predictions, tokens, tokens_lens, stage=TRAIN):
        """CTC loss as speechbrain.nnet.losses.ctc_loss(reduction="mean") on log_softmax(logits): absolute lengths by
        rounding the relative ones, zero_infinity, torch's "mean" (per-utterance loss / target length, then the batch mean)."""
        logits, wav_lens = predictions
        B, F, Vp = logits.shape
        in_lens = self._abs_lens(wav_lens, F)
        tokens = torch.as_tensor(tokens).cpu().to(torch.int32)
        tgt_lens = self._abs_lens(tokens_lens, tokens.shape[1])
        labels = torch.where(torch.arange(tokens.shape[1])[None, :] < tgt_lens[:, None], tokens, torch.full_like(tokens, -1))
        with torch.cuda.device(self.device):
            loss, nll, dlogits = hip.ctc_loss(logits, in_lens, labels, self.blank_index, "mean", True, 1.0,
                                              want_grad=stage == TRAIN)
            if stage != TRAIN and self.vocab is not None:
                from .metrics import WerAccumulator
                if self._wer is None:
                    self._wer = WerAccumulator(self.vocab, self.blank_index, self.device, delimiter=" ")
                self._wer.add(logits, labels.to(self.device), in_lens)
        self._dlogits = dlogits
        return loss

    def backward_encoder(self, dfeats: torch.Tensor, feats: torch.Tensor, stats: Optional[torch.Tensor]):
        """d loss / d feats -> wav2vec2 gradients: back through the output normalisation, then the engine's backward."""
        with torch.cuda.device(self.device):
            if self.output_norm:
                dh = torch.empty_like(dfeats)
                hip.check(hip.lib.ssak_utt_norm_bwd(hip.ptr(dfeats), hip.ptr(feats), hip.ptr(dh), dfeats.shape[0],
                                                    dfeats[0].numel(), 1, hip.ptr(stats), hip.ptr(self._un_ws),
                                                    self._un_ws.numel(), hip.stream()))
                dfeats = dh
            self.wav2vec2.backward_hidden(dfeats)

    def fit_batch(self, wavs, wav_lens, tokens, tokens_lens, check_finite: bool = True):
        """One optimizer step (:95-137, the fp32 branch); returns the detached loss tensor."""
        outputs = self.compute_forward(wavs, wav_lens, TRAIN)
        loss = self.compute_objectives(outputs, tokens, tokens_lens, TRAIN)
        feats, stats = self._fwd
        dfeats = self.head.backward(self._dlogits, need_input_grad=not self.freeze)
        head_work = torch.distributed.all_reduce(self.head.grads, async_op=True) if self.dist else None
        w = self.wav2vec2
        with torch.cuda.device(self.device):
            st = hip.stream()
            if not self.freeze:
                self.backward_encoder(dfeats, feats, stats)
            if head_work is not None:
                head_work.wait()
            for wk in self._works:
                wk.wait()
            self._works.clear()
            # check_gradients: non-finite loss -> no update; otherwise clip the joint norm of all trainable parameters
            # (the decision is made on the MINIMUM of the ranks' flags: the gradients are already summed over ranks, so a rank
            # with a finite local loss must not apply what a non-finite rank poisoned, and replicas must not diverge)
            ok = True
            if check_finite:
                flag = torch.isfinite(loss).all().to(torch.float32).reshape(1)
                if self.dist:
                    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                ok = bool(flag.item() > 0)
            if ok:
                scale = 1.0 / self.world
                hip.check(hip.lib.ssak_grad_sumsq(hip.ptr(self.head.grads), self.head.num_params, hip.ptr(self.gnorm_sq),
                                                  hip.ptr(self._sumsq_ws), self._sumsq_ws.numel() * 4, st))
                if not self.freeze:
                    hip.check(hip.lib.ssak_grad_sumsq_add(hip.ptr(w.grads), w.num_trainable, hip.ptr(self.gnorm_sq),
                                                          hip.ptr(self._sumsq_ws), self._sumsq_ws.numel() * 4, st))
                    o = self.wav2vec_optimizer
                    o.step_count += 1
                    hip.check(hip.lib.ssak_adamw_step(hip.ptr(w.params), hip.ptr(w.grads), hip.ptr(o.exp_avg), hip.ptr(o.exp_avg_sq),
                                                      hip.ptr(w.shadow), w.num_trainable, hip.ptr(self.gnorm_sq), self.max_grad_norm,
                                                      scale, o.lr, o.betas[0], o.betas[1], o.eps, 0.0, o.step_count, st))
                    w.sync_weights(full=False)
                self.model_optimizer.step(self.gnorm_sq, self.max_grad_norm, scale)
        self.optimizer_step += 1
        self._fwd = self._dlogits = None
        return loss.detach()

    def evaluate_batch(self, wavs, wav_lens, tokens, tokens_lens, stage=VALID):
        predictions = self.compute_forward(wavs, wav_lens, stage)
        return self.compute_objectives(predictions, tokens, tokens_lens, stage).detach()

    def on_stage_end(self, stage, stage_loss: float):
        """After a validation pass: NewBob on the validation loss for both learning rates (:181-194); returns the stats the
        recipe logs (loss, WER when token classes were given, the learning rates in force during the finished stage)."""
        stats = {"loss": stage_loss}
        if stage == VALID:
            if self._wer is not None:
                stats["WER"] = 100.0 * self._wer.compute()["wer"]
                self._wer = None
            old_m, new_m = self.lr_annealing_model(stage_loss)
            old_w, new_w = self.lr_annealing_wav2vec(stage_loss)
            self.model_optimizer.lr = new_m
            if self.wav2vec_optimizer is not None:
                self.wav2vec_optimizer.lr = new_w
            stats.update(lr_model=old_m, lr_wav2vec=old_w)
        return stats
