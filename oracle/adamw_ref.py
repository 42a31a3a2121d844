"""numpy restatement of the optimizer tail of the train step.  TEST INFRASTRUCTURE ONLY.

Reference: HF ``Trainer`` inner loop as recorded in ``docker/transformers_modified/trainer.py:1827-1855``
(clip_grad_norm_ max 1.0 -> ``optimizer.step()`` -> ``lr_scheduler.step()`` -> ``zero_grad``), configured by
``ssak/train/transformers/wav2vec_train.py:353-384`` (``optim="adamw_torch"``, lr 1e-4, weight_decay 0.0,
``warmup_steps=500``, linear decay).  The arithmetic is ``torch.optim.AdamW`` / ``torch.nn.utils.clip_grad_norm_``
/ ``transformers.get_linear_schedule_with_warmup`` (un-vendored).  Pinned by ``tests/golden/adamw.npz``.
"""
from __future__ import annotations

import numpy as np


def linear_warmup_lr(base_lr: float, step: int, warmup_steps: int, total_steps: int) -> float:
    """lr used for optimizer step number ``step`` (0-based): HF linear schedule with warm-up.
    Golden check from the reference's own fixture: lr 2e-7, 4e-7 logged after steps 1, 2 with
    base 1e-4, warm-up 500 (tests/expected/train_transformers/trainer_state.json:12-13,27-28)."""
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    return base_lr * max(0.0, (total_steps - step) / max(1, total_steps - warmup_steps))


def clip_coef(grads, max_norm: float = 1.0):
    """``clip_grad_norm_``: global L2 norm; coef = min(1, max_norm / (norm + 1e-6))."""
    tot = np.sqrt(sum(float((np.asarray(g, dtype=np.float64) ** 2).sum()) for g in grads))
    return tot, min(1.0, max_norm / (tot + 1e-6))


def adamw_step(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """One ``torch.optim.AdamW`` update (step is 1-based).  fp32 in, fp32 out (new p, m, v)."""
    p = np.asarray(p, dtype=np.float32)
    g = np.asarray(g, dtype=np.float32)
    p = p * np.float32(1.0 - lr * weight_decay)
    m = (np.float32(beta1) * m + np.float32(1 - beta1) * g).astype(np.float32)
    v = (np.float32(beta2) * v + np.float32(1 - beta2) * g * g).astype(np.float32)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = np.sqrt(v) / np.float32(np.sqrt(bc2)) + np.float32(eps)
    p = (p - np.float32(lr / bc1) * (m / denom)).astype(np.float32)
    return p, m, v
