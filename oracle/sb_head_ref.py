"""CPU oracle (test infrastructure only) for the SpeechBrain recipe's head and step -- plain torch, fp32.

PARITY UNPINNED against speechbrain itself: the package is not installed in this image and the reference repository
holds no numeric fixture for this recipe (its test compares a training log produced with downloaded weights,
tests/unittests/test_train_speechbrain.py:26-77).  What is restated here is the torch behaviour speechbrain wraps, at the
reference's call sites:
  * ``feats = modules.wav2vec2(wavs)`` (ssak/train/speechbrain/wav2vec_train.py:51): speechbrain's HuggingFaceWav2Vec2 --
    ``F.layer_norm(wav, wav.shape[1:])``, ``Wav2Vec2Model(wav)[0]`` (oracle/w2v2_ref.py, stage "last_hidden"),
    ``F.layer_norm(out, out.shape[1:])`` when ``output_norm``;
  * ``x = modules.enc(feats)`` / ``ctc_lin`` (:52-53) with the yaml's modules
    (ssak/train/speechbrain/fr/hyperparameters_wav2vec_finetune_cv-fr.yaml:87-111): speechbrain Linear = nn.Linear on the
    last axis; speechbrain BatchNorm1d on [B, T, C] = nn.BatchNorm1d(C) over batch and time (eps 1e-5, momentum 0.1);
    LeakyReLU(0.01); Dropout;
  * ``ctc_cost`` (:65, yaml :116-117): speechbrain.nnet.losses.ctc_loss with reduction "mean" = F.ctc_loss(log_probs^T,
    targets, round(rel_len * T), round(rel_len * L), blank, zero_infinity=True, reduction="mean");
  * ``model_optimizer.step()`` (:125-127): torch.optim.Adadelta(lr, rho 0.95, eps 1e-8) after
    clip_grad_norm_(all parameters, 5.0) (speechbrain Brain.check_gradients).
Dropout masks are inputs (the device derives them from a counter hash; the tests recover them from the device output).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def utt_norm(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    return F.layer_norm(x, x.shape[1:], eps=eps)


class _StoreBF16(torch.autograd.Function):
    """Value as the device stores it (bf16, round to nearest even), gradient passed through unchanged."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g


def head_forward(sd: Dict[str, torch.Tensor], feats: torch.Tensor, masks: Optional[Sequence] = None, train: bool = True,
                 dropouts=(0.15, 0.15, 0.0), slope: float = 0.01, eps: float = 1e-5, momentum: float = 0.1,
                 running: Optional[List] = None, stages: Optional[dict] = None, bf16_storage: bool = False) -> torch.Tensor:
    """feats [B, T, H] -> logits [B, T, V].  ``masks[i]`` [B, T, D] bool keep-mask of block i (None = keep all);
    ``running`` = [(running_mean, running_var)] per block, updated in place in training mode like nn.BatchNorm1d.

    ``bf16_storage``: round the GEMM operands the device keeps in bf16 (weights, Linear outputs, block outputs) at the
    same points, all arithmetic staying fp32.  At random initialisation the head's weight gradients are small residuals of
    large cancelling terms (BatchNorm removes the common-mode part of the CTC gradient, LeakyReLU's step makes the rest
    depend on the sign of near-zero activations): a 2^-9 relative perturbation of the forward values moves them by ~10 %,
    whatever the implementation.  The tests therefore compare gradients against this storage-matched form (tight) and
    logits / loss against the plain fp32 form."""
    r = _StoreBF16.apply if bf16_storage else (lambda t: t)
    h = feats
    for i in range(len(dropouts)):
        k = i + 1
        a = r(F.linear(h, r(sd[f"0.linear{k}.w.weight"]), sd[f"0.linear{k}.w.bias"]))
        rm, rv = running[i] if running is not None else (None, None)
        z = F.batch_norm(a.transpose(1, 2), rm, rv, sd[f"0.bn{k}.norm.weight"], sd[f"0.bn{k}.norm.bias"],
                         training=train or rm is None, momentum=momentum, eps=eps).transpose(1, 2)
        y = F.leaky_relu(z, slope)
        if train and dropouts[i] > 0 and masks is not None and masks[i] is not None:
            y = y * masks[i] / (1.0 - dropouts[i])
        y = r(y)
        if stages is not None:
            stages[f"a{k}"], stages[f"y{k}"] = a, y
        h = y
    return F.linear(h, r(sd["1.w.weight"]), sd["1.w.bias"])


def ctc_cost(logits: torch.Tensor, tokens: torch.Tensor, wav_lens, tokens_lens, blank: int = 0) -> torch.Tensor:
    logp = F.log_softmax(logits, dim=-1)
    in_lens = torch.round(torch.as_tensor(wav_lens, dtype=torch.float32) * logp.shape[1]).int()
    tgt_lens = torch.round(torch.as_tensor(tokens_lens, dtype=torch.float32) * tokens.shape[1]).int()
    return F.ctc_loss(logp.transpose(0, 1), tokens, in_lens, tgt_lens, blank, zero_infinity=True, reduction="mean")


def head_loss_and_grads(sd, feats, tokens, wav_lens, tokens_lens, masks=None, blank=0, **kw):
    """-> loss, logits, {name: grad}, d loss / d feats"""
    q = {n: t.detach().clone().float().requires_grad_(True) for n, t in sd.items()}
    f = feats.detach().clone().float().requires_grad_(True)
    logits = head_forward(q, f, masks, True, **kw)
    loss = ctc_cost(logits, tokens, wav_lens, tokens_lens, blank)
    loss.backward()
    return loss.detach(), logits.detach(), {n: q[n].grad for n in q}, f.grad


def clip_coef(grads: Sequence[torch.Tensor], max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_: min(1, max_norm / (total_norm + 1e-6))"""
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads)))
    return min(1.0, max_norm / (total + 1e-6))


def adadelta_step(p: np.ndarray, g: np.ndarray, square_avg: np.ndarray, acc_delta: np.ndarray, lr=1.0, rho=0.95, eps=1e-8,
                  weight_decay=0.0):
    """One torch.optim.Adadelta update, in place (fp32 arithmetic)."""
    f = np.float32
    if weight_decay:
        g = g + f(weight_decay) * p
    square_avg *= f(rho)
    square_avg += f(1 - rho) * g * g
    std = np.sqrt(square_avg + f(eps))
    delta = np.sqrt(acc_delta + f(eps)) / std * g
    acc_delta *= f(rho)
    acc_delta += f(1 - rho) * delta * delta
    p -= f(lr) * delta
    return p


def new_bob(values: Sequence[float], initial: float, factor: float, threshold: float = 0.0025, patient: int = 0):
    """Learning rates in force after each validation loss of ``values`` (speechbrain NewBobScheduler; yaml :124-135)."""
    lr, out, prev, cur_pat = initial, [], None, patient
    for v in values:
        if prev is not None:
            imp = 0.0 if prev == 0 else (prev - v) / prev
            if imp < threshold:
                if cur_pat == 0:
                    lr *= factor
                    cur_pat = patient
                else:
                    cur_pat -= 1
        prev = v
        out.append(lr)
    return out


def recipe_loss_and_grads(p, cfg, head_sd, wavs, tokens, wav_lens, tokens_lens, normalize_wav=True, output_norm=True, blank=0,
                          dropouts=(0.0, 0.0, 0.0)):
    """The unfrozen recipe end to end (wav2vec_train.py:39-66,111-112): -> loss, {w2v2 grads}, {head grads}."""
    from . import w2v2_ref as R
    names = R.trainable_names(cfg, True)
    q = {n: (t.detach().clone().requires_grad_(True) if n in names else t.detach()) for n, t in p.items()}
    hq = {n: t.detach().clone().float().requires_grad_(True) for n, t in head_sd.items()}
    x = torch.as_tensor(wavs, dtype=torch.float32)
    if normalize_wav:
        x = utt_norm(x)
    st = {}
    R.forward(q, cfg, x, None, None, train=True, stages=st)
    feats = st["last_hidden"]
    if output_norm:
        feats = utt_norm(feats)
    logits = head_forward(hq, feats, None, True, dropouts=dropouts)
    loss = ctc_cost(logits, tokens, wav_lens, tokens_lens, blank)
    loss.backward()
    wg = {n: (q[n].grad if q[n].grad is not None else torch.zeros_like(q[n])) for n in names if not n.startswith("lm_head")}
    return loss.detach(), wg, {n: hq[n].grad for n in hq}
