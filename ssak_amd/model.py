"""Host-side mirror of the acoustic-model interface the reference calls.

``Wav2Vec2ForCTC`` keeps the call contract of ``transformers.Wav2Vec2ForCTC`` as used by SSAK --
``model(input_values, attention_mask=..., labels=...)`` returning ``.loss`` / ``.logits``
(ssak/train/transformers/wav2vec_train.py:387-415 via HF Trainer; ssak/infer/transformers_infer.py:235) --
but every tensor operation runs in ``libssak_hip.so`` (HIP kernels for gfx950).  torch supplies device
buffers, the stream and (in ``ssak_amd.trainer``) the RCCL process group; nothing else.

State lives in four flat fp32 device buffers (params, grads, exp_avg, exp_avg_sq) plus one bf16 shadow,
laid out by the engine (``ssak_w2v2_param_info``); ``state_dict``/``load_state_dict`` speak the HF names.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import numpy as np
import torch

from . import hip
from .config import Wav2Vec2Config


class CTCOutput:
    __slots__ = ("loss", "logits", "nll", "frame_lens")

    def __init__(self, loss, logits, nll=None, frame_lens=None):
        self.loss, self.logits, self.nll, self.frame_lens = loss, logits, nll, frame_lens


def conv_out_lengths(cfg: Wav2Vec2Config, lengths):
    """floor((L-k)/s)+1 chained over the conv layers (transformers modeling_wav2vec2.py:997-1016)."""
    L = np.asarray(lengths, dtype=np.int64)
    for k, s in zip(cfg.conv_kernel, cfg.conv_stride):
        L = np.floor_divide(L - k, s) + 1
    return L


def compute_mask_indices(shape, mask_prob, mask_length, lengths=None, min_masks=0, rng=np.random) -> np.ndarray:
    """SpecAugment span sampling with the draw order of HF ``_compute_mask_indices``
    (transformers modeling_wav2vec2.py:101-217): one uniform for the probabilistic rounding, then one
    ``choice`` without replacement per utterance.  Host-side like the reference, off the device's critical path."""
    B, S = shape
    if mask_length > S:
        raise ValueError("`mask_length` has to be smaller than `sequence_length`")
    eps = rng.rand(1).item()

    def nspan(L):
        n = max(int(mask_prob * L / mask_length + eps), min_masks)
        if n * mask_length > S:
            n = S // mask_length
        if L - (mask_length - 1) < n:
            n = max(L - (mask_length - 1), 0)
        return n

    mask = np.zeros((B, S), dtype=bool)
    nmax = nspan(S)
    if nmax == 0:
        return mask
    for b in range(B):
        L = S if lengths is None else int(lengths[b])
        n = nspan(L)
        idx = rng.choice(np.arange(L - (mask_length - 1)), n, replace=False)
        dummy = S - 1 if len(idx) == 0 else idx[0]
        idx = np.concatenate([idx, np.ones(nmax - n, dtype=np.int32) * dummy])
        span = np.minimum(idx[:, None] + np.arange(mask_length)[None, :], S - 1).reshape(-1)
        mask[b, span.astype(np.int64)] = True
    return mask


class Wav2Vec2ForCTC:
    def __init__(self, config: Wav2Vec2Config, device: str = "cuda:0", freeze_feature_encoder: bool = True,
                 seed: int = 69, exact: Optional[bool] = None):
        """``exact`` (default: environment SSAK_EXACT=1): the fp32-exact VERIFICATION mode of the engine -- float activations,
        fp32 matrix products on the master weights, exact erf GELU -- comparable with the fp32 reference at 1e-4; slow."""
        if not torch.cuda.is_available():
            raise RuntimeError("ssak_amd needs an MI355X: there is no CPU fallback for the acoustic model")
        self.config = config
        self.device = torch.device(device)
        self.training = False
        self.freeze = freeze_feature_encoder
        import os
        self.exact = bool(int(os.environ.get("SSAK_EXACT", "0"))) if exact is None else bool(exact)
        c = self._c_config(config, freeze_feature_encoder)
        c.exact = int(self.exact)
        self._finish_init(c, seed)

    @staticmethod
    def _c_config(config, freeze_feature_encoder):
        c = hip.W2V2Config()
        c.vocab_size = (config.vocab_size + 7) // 8 * 8  # engine wants a multiple of 8; see _pad_vocab
        c.hidden_size, c.num_layers = config.hidden_size, config.num_hidden_layers
        c.num_heads, c.intermediate_size = config.num_attention_heads, config.intermediate_size
        c.num_conv_layers = len(config.conv_dim)
        for i, (d, k, s) in enumerate(zip(config.conv_dim, config.conv_kernel, config.conv_stride)):
            c.conv_dim[i], c.conv_kernel[i], c.conv_stride[i] = d, k, s
        c.conv_bias = int(config.conv_bias)
        c.feat_extract_norm = {"group": 0, "layer": 1}[config.feat_extract_norm]
        c.do_stable_layer_norm = int(config.do_stable_layer_norm)
        c.num_conv_pos_embeddings = config.num_conv_pos_embeddings
        c.num_conv_pos_embedding_groups = config.num_conv_pos_embedding_groups
        c.layer_norm_eps = config.layer_norm_eps
        c.attention_dropout, c.hidden_dropout = config.attention_dropout, config.hidden_dropout
        c.activation_dropout, c.feat_proj_dropout = config.activation_dropout, config.feat_proj_dropout
        c.final_dropout = config.final_dropout
        c.freeze_feature_encoder = int(freeze_feature_encoder)
        return c

    @classmethod
    def grad_ranges(cls, config, freeze_feature_encoder: bool = True):
        """[(offset, count)] of the gradient buckets the backward announces, from the configuration alone (no GPU needed)."""
        c = cls._c_config(config, freeze_feature_encoder)
        off, cnt = (C.c_long * 128)(), (C.c_long * 128)()
        n = hip.lib.ssak_w2v2_grad_ranges(C.byref(c), off, cnt, 128)
        if n < 0:
            hip.check(n)
        return [(off[i], cnt[i]) for i in range(n)]

    def can_fold_normalisation(self) -> bool:
        """True when the waveform normalisation (a1) can ride in conv0's GroupNorm statistics instead of its own pass: the wav2vec2
        group-norm feature encoder, frozen (the caller still has to pass full-length, unpadded utterances)."""
        return getattr(self._c, "arch", 0) == 0 and self.config.feat_extract_norm == "group" and bool(self._c.freeze_feature_encoder)

    def announced_grad_ranges(self):
        """The (offset, count) sequence this model's backward announces (its own engine configuration, Whisper included)."""
        off, cnt = (C.c_long * 128)(), (C.c_long * 128)()
        n = hip.lib.ssak_w2v2_grad_ranges(C.byref(self._c), off, cnt, 128)
        if n < 0:
            hip.check(n)
        return [(off[i], cnt[i]) for i in range(n)]

    def _finish_init(self, c, seed):
        config = self.config
        self._seed = seed
        self._raw_input = False  # SSAK_W2V2_OPT_RAW_INPUT as last set by forward()
        self._c = c
        h = C.c_void_p()
        hip.check(hip.lib.ssak_w2v2_create(C.byref(c), C.byref(h)))
        self._h = h
        self.num_params = hip.lib.ssak_w2v2_num_params(h)
        self.num_trainable = hip.lib.ssak_w2v2_num_trainable(h)
        self.layout: Dict[str, tuple] = {}
        name = C.create_string_buffer(256)
        off, numel, nd = C.c_long(), C.c_long(), C.c_int()
        shp = (C.c_long * 4)()
        for i in range(hip.lib.ssak_w2v2_param_count(h)):
            hip.check(hip.lib.ssak_w2v2_param_info(h, i, name, 256, C.byref(off), C.byref(numel), C.byref(nd), shp))
            self.layout[name.value.decode()] = (off.value, numel.value, tuple(shp[j] for j in range(nd.value)))
        with torch.cuda.device(self.device):
            self.params = torch.zeros(self.num_params, dtype=torch.float32, device=self.device)
            self.grads = torch.zeros(self.num_params, dtype=torch.float32, device=self.device)
            self.shadow = torch.zeros(self.num_params, dtype=torch.bfloat16, device=self.device)
        hip.check(hip.lib.ssak_w2v2_bind(h, hip.ptr(self.params), hip.ptr(self.grads), hip.ptr(self.shadow)))
        self._ws = None
        self._ws_key = None
        self.train_forwards = 0
        self.kept_layers = 0
        self.last_layer_keep = None
        self.reseed(seed)
        self._last = None
        self._pinned_mask = {}

    def reseed(self, seed: int):
        """Restart the dropout-mask counter stream and the host generator of the SpecAugment spans / LayerDrop decisions."""
        self._seed = int(seed)
        self._step_seed = np.random.SeedSequence(int(seed)).generate_state(1, dtype=np.uint64)[0]
        self._host_rng = np.random.RandomState(int(seed) % (1 << 32))

    def __del__(self):
        h = getattr(self, "_h", None)
        lib = getattr(hip, "lib", None)  # may already be gone at interpreter shutdown
        if h and lib is not None:
            lib.ssak_w2v2_destroy(h)
            self._h = None

    # ------------------------------------------------------------------ parameters
    def param(self, name: str) -> torch.Tensor:
        off, n, shape = self.layout[name]
        return self.params[off:off + n].view(shape)

    def grad(self, name: str) -> torch.Tensor:
        off, n, shape = self.layout[name]
        return self.grads[off:off + n].view(shape)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        self.wait_params()
        V = self.config.vocab_size
        return {n: (self.param(n)[:V] if n in self._HEAD else self.param(n)).detach().cpu().clone() for n in self.layout}

    _HEAD = ("lm_head.weight", "lm_head.bias")
    _PAD_BIAS = -1.0e4  # padded vocabulary entries: probability exp(-1e4) = 0, hence exactly zero gradient

    def _pad_vocab(self, name: str, t: torch.Tensor) -> torch.Tensor:
        """Real tokenizers rarely have a multiple of 8 symbols: the head is padded with inert classes."""
        V, Vp = self.config.vocab_size, self.layout[name][2][0]
        if name not in self._HEAD or t.shape[0] == Vp:
            return t
        if t.shape[0] != V:
            return t  # size mismatch is reported by the caller
        pad_shape = (Vp - V,) + tuple(t.shape[1:])
        fill = 0.0 if name.endswith("weight") else self._PAD_BIAS
        return torch.cat([t, torch.full(pad_shape, fill, dtype=t.dtype)], dim=0)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        sd = {n: (self._pad_vocab(n, torch.as_tensor(t)) if n in self.layout else t) for n, t in sd.items()}
        missing = [n for n in self.layout if n not in sd]
        extra = [n for n in sd if n not in self.layout]
        if strict and (missing or extra):
            raise RuntimeError(f"state_dict mismatch: missing {missing[:4]} unexpected {extra[:4]}")
        self.wait_params()
        for n, (off, numel, shape) in self.layout.items():
            if n in sd:
                t = torch.as_tensor(sd[n]).to(torch.float32)
                if tuple(t.shape) != shape:
                    raise RuntimeError(f"size mismatch for {n}: {tuple(t.shape)} vs {shape}")
                self.params[off:off + numel].copy_(t.reshape(-1).to(self.device), non_blocking=True)
        self.sync_weights(full=True)
        return self

    def sync_weights(self, full: bool = True):
        with torch.cuda.device(self.device):
            hip.check(hip.lib.ssak_w2v2_sync_weights(self._h, int(full), hip.stream()))

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def freeze_feature_encoder(self):
        if not self.freeze:
            raise RuntimeError("construct the model with freeze_feature_encoder=True")
        return self

    def num_frames(self, T: int) -> int:
        return hip.lib.ssak_w2v2_num_frames(self._h, int(T))

    # ------------------------------------------------------------------ forward / backward
    def _workspace(self, B, T, training):
        key = (B, T, bool(training))
        need = hip.lib.ssak_w2v2_workspace_bytes(self._h, B, T, int(training))
        if need == 0:
            raise ValueError(f"input of {T} samples is too short for the feature encoder")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._ws_key = key
        return self._ws

    def __call__(self, input_values, attention_mask=None, labels=None, mask_time_indices=None, layer_keep=None,
                 lengths=None, dropout_seed=None, **kw):
        return self.forward(input_values, attention_mask, labels, mask_time_indices, layer_keep, lengths, dropout_seed, **kw)

    def _prelude(self, input_values, attention_mask, mask_time_indices, layer_keep, lengths, training, dropout_seed=None):
        """Everything before the engine call: device copies, workspace, the host-drawn SpecAugment spans / LayerDrop
        decisions and the step's dropout seed."""
        cfg = self.config
        x = input_values.to(device=self.device, dtype=torch.float32).contiguous()
        B, T = x.shape[0], x.shape[-1]  # [B, samples] (wav2vec2) or [B, mel bins, feature frames] (Whisper encoder)
        if attention_mask is not None and lengths is None:
            lengths = attention_mask.to(self.device).sum(-1)
        lens_dev = None
        if lengths is not None:
            lens_dev = torch.as_tensor(lengths).to(device=self.device, dtype=torch.int32).contiguous()
        F = self.num_frames(T)
        ws = self._workspace(B, T, training)
        # stochastic regularisers drawn on the host ahead of the step, as the reference does
        mask_dev, keep_arr = None, None
        if training:
            if mask_time_indices is None and cfg.mask_time_prob > 0:
                fl = None if lengths is None else conv_out_lengths(cfg, torch.as_tensor(lengths).cpu().numpy())
                mask_time_indices = compute_mask_indices((B, F), cfg.mask_time_prob, cfg.mask_time_length, fl,
                                                         cfg.mask_time_min_masks, rng=self._host_rng)
            if layer_keep is None and cfg.layerdrop > 0:
                layer_keep = self._host_rng.rand(cfg.num_hidden_layers) >= cfg.layerdrop
        if mask_time_indices is not None:
            if torch.is_tensor(mask_time_indices) and mask_time_indices.is_cuda:
                mask_dev = mask_time_indices.to(torch.uint8).contiguous()
            else:
                # pinned staging + async copy: a pageable H2D copy would block the host until the stream drains
                mh = torch.as_tensor(np.ascontiguousarray(mask_time_indices)).to(torch.uint8)
                pin = self._pinned_mask.get(tuple(mh.shape))
                if pin is None:
                    pin = (torch.empty(mh.shape, dtype=torch.uint8).pin_memory(),
                           torch.empty(mh.shape, dtype=torch.uint8, device=self.device), torch.cuda.Event())
                    self._pinned_mask[tuple(mh.shape)] = pin
                else:
                    pin[2].synchronize()  # the previous copy out of the pinned buffer has completed
                pin[0].copy_(mh)
                with torch.cuda.device(self.device):
                    pin[1].copy_(pin[0], non_blocking=True)
                    pin[2].record()
                mask_dev = pin[1]
        if layer_keep is not None:
            keep_arr = (C.c_uint8 * cfg.num_hidden_layers)(*[int(bool(k)) for k in layer_keep])
        if training:  # LayerDrop bookkeeping (benchmarks price the step by the layers that actually ran)
            self.last_layer_keep = None if layer_keep is None else [bool(k) for k in layer_keep]  # (AdamW(skip_unused_layers=True) reads it)
            self.train_forwards += 1
            self.kept_layers += cfg.num_hidden_layers if layer_keep is None else int(sum(bool(k) for k in layer_keep))
        self._step_seed = (int(self._step_seed) * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        # like the SpecAugment mask and the LayerDrop decisions, the step's dropout seed can be supplied by the caller
        # (regularisers-on parity runs: oracle/dropout_hash.py turns it into the masks transformers is fed)
        self._used_seed = int(self._step_seed) if dropout_seed is None else int(dropout_seed) % (1 << 64)
        flens = torch.empty(B, dtype=torch.int32, device=self.device) if lens_dev is not None else None
        return x, B, T, F, lens_dev, ws, mask_dev, keep_arr, flens

    def forward(self, input_values: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                labels: Optional[torch.Tensor] = None, mask_time_indices=None, layer_keep=None, lengths=None,
                dropout_seed: Optional[int] = None, raw_input: bool = False):
        """``raw_input``: ``input_values`` are RAW full-length waveforms (no padding inside the batch); the feature extractor's
        normalisation is folded into the first conv layer's GroupNorm (``can_fold_normalisation()``; SSAK_W2V2_OPT_RAW_INPUT)."""
        cfg = self.config
        training = self.training
        if bool(raw_input) != self._raw_input:
            if raw_input and not self.can_fold_normalisation():
                raise ValueError("raw_input needs the group-norm feature encoder, frozen")
            self.set_option(hip.W2V2_OPT_RAW_INPUT, int(bool(raw_input)))
            self._raw_input = bool(raw_input)
        if labels is not None:
            labels = torch.as_tensor(labels)
            # the reference checks labels.max() on every call, which forces a device sync when the labels live on
            # the GPU (modeling_wav2vec2.py:1686-1687); here host-resident labels are checked on the host and
            # device-resident ones are validated inside the CTC kernel (bad label -> NaN loss), so the step never syncs
            if not labels.is_cuda and labels.numel() and int(labels.max()) >= cfg.vocab_size:
                raise ValueError(f"Label values must be <= vocab_size: {cfg.vocab_size}")
        x, B, T, F, lens_dev, ws, mask_dev, keep_arr, flens = self._prelude(input_values, attention_mask, mask_time_indices,
                                                                            layer_keep, lengths, training, dropout_seed)
        Vp = self._c.vocab_size
        logits = torch.empty((B, F, Vp), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            hip.check(hip.lib.ssak_w2v2_forward(self._h, hip.ptr(x), hip.ptr(lens_dev), B, T, hip.ptr(mask_dev), keep_arr,
                                                C.c_uint64(self._used_seed), int(training), hip.ptr(logits),
                                                hip.ptr(flens), hip.ptr(ws), ws.numel(), hip.stream()))
            loss = nll = dlogits = None
            if labels is not None:
                loss, nll, dlogits = hip.ctc_loss(logits, flens, labels, cfg.pad_token_id, cfg.ctc_loss_reduction,
                                                  cfg.ctc_zero_infinity, 1.0, want_grad=training)
        self._last = (dlogits, mask_dev, lens_dev, x) if training else None  # x stays alive: conv0's backward re-reads it
        if Vp != cfg.vocab_size:
            logits = logits[..., :cfg.vocab_size]  # view: the inert padding classes are not part of the contract
        return CTCOutput(loss, logits, nll, flens)

    def forward_hidden(self, input_values: torch.Tensor, attention_mask=None, mask_time_indices=None, layer_keep=None,
                       lengths=None, keep_graph: Optional[bool] = None):
        """The encoder's last hidden state [B, F, H] bf16 -- ``Wav2Vec2Model(wav)[0]``, what the SpeechBrain recipe's
        ``modules.wav2vec2`` wraps (ssak/train/speechbrain/wav2vec_train.py:51).  ``keep_graph`` (default: training mode)
        keeps the activations for :meth:`backward_hidden`."""
        training = self.training
        x, B, T, F, lens_dev, ws, mask_dev, keep_arr, flens = self._prelude(input_values, attention_mask, mask_time_indices,
                                                                            layer_keep, lengths, training)
        hidden = torch.empty((B, F, self.config.hidden_size), dtype=torch.bfloat16, device=self.device)
        with torch.cuda.device(self.device):
            hip.check(hip.lib.ssak_w2v2_forward_hidden(self._h, hip.ptr(x), hip.ptr(lens_dev), B, T, hip.ptr(mask_dev), keep_arr,
                                                       C.c_uint64(self._used_seed), int(training), hip.ptr(hidden),
                                                       hip.ptr(flens), hip.ptr(ws), ws.numel(), hip.stream()))
        self._last = ("hidden", mask_dev, lens_dev, x) if training else None
        return hidden, flens

    def backward_hidden(self, dhidden: torch.Tensor):
        """d loss / d params from d loss / d hidden [B, F, H] bf16 (the unfrozen wav2vec2 of the SpeechBrain recipe)."""
        if self._last is None or not isinstance(self._last[0], str):
            raise RuntimeError("backward_hidden() needs a training-mode forward_hidden()")
        assert dhidden.dtype == torch.bfloat16 and dhidden.is_contiguous()
        with torch.cuda.device(self.device):
            hip.check(hip.lib.ssak_w2v2_backward_hidden(self._h, hip.ptr(dhidden), hip.ptr(self._ws), self._ws.numel(),
                                                        hip.stream()))
        self._last = None
        self._raise_callback_error()

    def backward(self, grad_scale: float = 1.0):
        """d loss / d params into ``self.grads`` (the counterpart of ``loss.backward()``)."""
        if self._last is None or self._last[0] is None or isinstance(self._last[0], str):
            raise RuntimeError("backward() needs a training-mode forward with labels")
        dlogits = self._last[0]
        if grad_scale != 1.0:
            dlogits = dlogits * grad_scale
        with torch.cuda.device(self.device):
            hip.check(hip.lib.ssak_w2v2_backward(self._h, hip.ptr(dlogits), hip.ptr(self._ws), self._ws.numel(), hip.stream()))
        self._last = None
        self._raise_callback_error()

    def set_grad_ready_callback(self, fn):
        """``fn(offset, count)`` is called during :meth:`backward` whenever grads[offset:offset+count] is final
        (kernels enqueued); ``None`` removes it.  Used by the data-parallel trainer for bucketed all-reduce."""
        if fn is None:
            self._cb = hip.GRAD_READY_FN()  # NULL function pointer
        else:
            def guarded(off, cnt, _user):
                # ctypes would print and swallow an exception raised inside the callback (a failed all-reduce would go
                # unnoticed): keep the first one and re-raise it when the backward call returns
                try:
                    fn(off, cnt)
                except BaseException as exc:  # noqa: BLE001
                    if self._cb_error is None:
                        self._cb_error = exc
            self._cb = hip.GRAD_READY_FN(guarded)
        self._cb_error = None
        hip.check(hip.lib.ssak_w2v2_set_grad_ready_callback(self._h, self._cb, None))

    def _raise_callback_error(self):
        exc, self._cb_error = getattr(self, "_cb_error", None), None
        if exc is not None:
            raise exc

    def set_option(self, option: int, value: int):
        """Per-handle execution options (``ssak_w2v2_set_option``): hip.W2V2_OPT_DYNAMIC_TILES (ticket tile order of the
        persistent GEMMs, for data-parallel runs), hip.W2V2_OPT_ATTENTION_BWD (hip.ATTN_BWD_*; one form since ABI 400)."""
        hip.check(hip.lib.ssak_w2v2_set_option(self._h, int(option), int(value)))

    def set_param_event(self, event: Optional[torch.cuda.Event], stall_begin=None, stall_end=None):
        """The optimizer runs on a side stream: ``event`` is recorded there after the update; every forward waits for it at
        its first read of a trainable parameter (``ssak_w2v2_set_param_event``).  The host-side accessors wait through
        :meth:`wait_params`."""
        self._param_event = event
        raw = [None if ev is None else C.c_void_p(ev.cuda_event) for ev in (event, stall_begin, stall_end)]
        self._param_event_keep = (event, stall_begin, stall_end)  # the engine holds raw handles
        hip.check(hip.lib.ssak_w2v2_set_param_event(self._h, *raw))

    # The flat buffers are exposed through properties that first order the CURRENT stream behind a pending side-stream
    # optimizer update (a stream-wait on an event: microseconds on the host, nothing on the device once the update is done),
    # so host code can keep reading / writing ``model.params`` / ``model.grads`` right after ``Trainer.train_step``.
    def _buf(self, name):
        self.wait_params()
        return self.__dict__[name]

    params = property(lambda self: self._buf("_params"), lambda self, v: self.__dict__.__setitem__("_params", v))
    grads = property(lambda self: self._buf("_grads"), lambda self, v: self.__dict__.__setitem__("_grads", v))
    shadow = property(lambda self: self._buf("_shadow"), lambda self, v: self.__dict__.__setitem__("_shadow", v))

    def wait_params(self):
        """Make the current stream wait for a pending side-stream optimizer update (before reading or writing params)."""
        ev = getattr(self, "_param_event", None)
        if ev is not None:
            with torch.cuda.device(self.device):
                torch.cuda.current_stream().wait_event(ev)

    def named_grads(self):
        return {n: self.grad(n) for n, (off, _, _) in self.layout.items() if off < self.num_trainable}
