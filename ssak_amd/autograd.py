"""torch.autograd entry points over the C ABI, for a PyTorch host loop that keeps its own ``loss.backward()`` /
``torch.optim`` / ``clip_grad_norm_`` calls -- the shape of the reference's loop (HF Trainer: ``loss = model(**inputs).loss``,
``loss.backward()``, ``clip_grad_norm_``, ``optimizer.step()``, ``model.zero_grad()``;
docker/transformers_modified/trainer.py:2504-2550, 1827-1855) -- instead of ``ssak_amd.trainer``.

* :func:`ctc_loss` -- ``F.ctc_loss(log_softmax(logits))`` for logits produced by any torch module: the forward runs
  ``ssak_ctc_loss_fwd_bwd`` once (loss and d loss / d logits come out of the same lattice pass), the backward hands the
  stored gradient back, scaled by the incoming one.
* :class:`TorchWav2Vec2ForCTC` -- the engine-backed model seen as a module with ONE flat leaf parameter (``.params``, fp32
  master weights) whose ``.grad`` the engine's backward fills; ``model(input_values, labels=...)`` returns an output whose
  ``.loss`` is a differentiable scalar.  ``torch.optim.AdamW(model.parameters())`` then updates the master weights and the
  next forward refreshes the bf16 operand copies (noticed through the tensor's version counter).

Everything numerical still happens in ``libssak_hip.so``; torch contributes the tape.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import hip
from .model import CTCOutput, Wav2Vec2ForCTC


class _CTCLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, in_lens, labels, blank, reduction, zero_infinity):
        loss, nll, grad = hip.ctc_loss(logits.contiguous(), in_lens, labels, blank, reduction, zero_infinity, 1.0,
                                       want_grad=logits.requires_grad)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(nll)
        return loss[0], nll

    @staticmethod
    def backward(ctx, g_loss, _g_nll):
        (grad,) = ctx.saved_tensors
        return grad * g_loss, None, None, None, None, None


def ctc_loss(logits: torch.Tensor, in_lens: Optional[torch.Tensor], labels: torch.Tensor, blank: int = 0,
             reduction: str = "mean", zero_infinity: bool = True):
    """logits [B, F, V] fp32 on the device (any autograd history), labels [B, L] with negative padding ->
    (scalar loss, per-utterance negative log-likelihoods).  Log-softmax is part of the kernel."""
    return _CTCLoss.apply(logits, in_lens, labels, blank, reduction, zero_infinity)


class _EngineStep(torch.autograd.Function):
    """Forward = engine forward + CTC; backward = engine backward into the flat gradient buffer, returned as the gradient
    of the flat parameter (autograd then owns ``params.grad``: accumulation over several backward calls adds up as usual)."""

    @staticmethod
    def forward(ctx, params, model, input_values, attention_mask, labels, mask_time_indices, layer_keep, lengths, dropout_seed=None):
        out = Wav2Vec2ForCTC.forward(model, input_values, attention_mask, labels, mask_time_indices, layer_keep, lengths, dropout_seed)
        ctx.model = model
        ctx.mark_non_differentiable(out.logits)
        model._torch_out = out
        return out.loss[0], out.logits

    @staticmethod
    def backward(ctx, g_loss, _g_logits):
        m = ctx.model
        m.backward(grad_scale=1.0)
        g = m.grads * g_loss  # a fresh tensor: the engine's buffer is overwritten by the next backward
        return g, None, None, None, None, None, None, None, None


class TorchWav2Vec2ForCTC(Wav2Vec2ForCTC):
    """``Wav2Vec2ForCTC`` with a torch-visible parameter: see the module docstring."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.params.requires_grad_(True)
        self._synced_version = self.params._version

    def parameters(self):
        return [self.params]

    def zero_grad(self, set_to_none: bool = True):
        if set_to_none:
            self.params.grad = None
        elif self.params.grad is not None:
            self.params.grad.zero_()

    def load_state_dict(self, sd, strict: bool = True):
        with torch.no_grad():
            r = super().load_state_dict(sd, strict)
        self._synced_version = self.params._version
        return r

    def forward(self, input_values, attention_mask=None, labels=None, mask_time_indices=None, layer_keep=None, lengths=None,
                dropout_seed=None):
        if self.params._version != self._synced_version:  # an optimizer (or the caller) wrote the master weights
            self.sync_weights(full=True)
            self._synced_version = self.params._version
        if labels is None or not (self.training and torch.is_grad_enabled()):
            with torch.no_grad():
                return super().forward(input_values, attention_mask, labels, mask_time_indices, layer_keep, lengths, dropout_seed)
        loss, logits = _EngineStep.apply(self.params, self, input_values, attention_mask, labels, mask_time_indices, layer_keep,
                                         lengths, dropout_seed)
        out = self._torch_out
        self._torch_out = None
        return CTCOutput(loss, out.logits, out.nll, out.frame_lens)
