"""float64 numpy restatement of CTC loss + gradient w.r.t. logits.  TEST INFRASTRUCTURE ONLY.

Reference call site: ``Wav2Vec2ForCTC.forward`` -> ``log_softmax(fp32)`` -> ``F.ctc_loss``
(transformers modeling_wav2vec2.py:1705-1728), configured by ssak with ``reduction="mean"``
and ``zero_infinity=True`` (ssak/train/transformers/wav2vec_train.py:319,325).  The algorithm
is Graves et al. 2006 (alpha/beta lattice over the blank-extended label sequence) as
implemented by torch's native ``ctc_loss`` (un-vendored dependency ``torch``,
``requirements.txt:39``).  Pinned against ``torch.nn.functional.ctc_loss`` + autograd by
``oracle/gen_golden.py`` -> ``tests/golden/ctc_cases.npz``.
"""
from __future__ import annotations

import numpy as np

NEG_INF = -np.inf


def log_softmax(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    m = x.max(axis=-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True))


def _lse2(a, b):
    m = np.maximum(a, b)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = m + np.log(np.exp(a - m) + np.exp(b - m))
    return np.where(np.isneginf(m), NEG_INF, r)


def _lse3(a, b, c):
    return _lse2(_lse2(a, b), c)


def ctc_single(logits: np.ndarray, target: np.ndarray, in_len: int, blank: int = 0):
    """One utterance.  logits [F,V] (any float), target int [L].  Returns (nll, dnll/dlogits [F,V]);
    frames >= in_len get zero gradient.  nll = +inf when no alignment exists (in_len < L + repeats)."""
    Fr, V = logits.shape
    L = len(target)
    lp = log_softmax(logits[:in_len])
    S = 2 * L + 1
    ext = np.full(S, blank, dtype=np.int64)
    ext[1::2] = target
    # skip transition s-2 -> s allowed for label positions whose label differs from the one two back
    can_skip = np.zeros(S, dtype=bool)
    if L > 1:
        can_skip[3::2] = ext[3::2] != ext[1:-2:2]
    grad = np.zeros((Fr, V), dtype=np.float64)
    if in_len == 0:
        return (0.0 if L == 0 else np.inf), grad
    alpha = np.full((in_len, S), NEG_INF)
    alpha[0, 0] = lp[0, blank]
    if S > 1:
        alpha[0, 1] = lp[0, ext[1]]
    for t in range(1, in_len):
        a = alpha[t - 1]
        a1 = np.concatenate(([NEG_INF], a))[:S]
        a2 = np.where(can_skip, np.concatenate(([NEG_INF, NEG_INF], a))[:S], NEG_INF)
        alpha[t] = _lse3(a, a1, a2) + lp[t, ext]
    ll = alpha[-1, S - 1] if S == 1 else _lse2(alpha[-1, S - 1], alpha[-1, S - 2])
    nll = -float(ll)
    if not np.isfinite(nll):
        return np.inf, grad
    beta = np.full((in_len, S), NEG_INF)
    beta[-1, S - 1] = lp[-1, ext[S - 1]]
    if S > 1:
        beta[-1, S - 2] = lp[-1, ext[S - 2]]
    skip_from = np.concatenate((can_skip, [False, False]))[2:]  # s -> s+2 allowed iff can_skip[s+2]
    for t in range(in_len - 2, -1, -1):
        b = beta[t + 1]
        b1 = np.concatenate((b, [NEG_INF]))[1:]
        b2 = np.where(skip_from, np.concatenate((b, [NEG_INF, NEG_INF]))[2:], NEG_INF)
        beta[t] = _lse3(b, b1, b2) + lp[t, ext]
    # d nll / d logits[t,c] = softmax[t,c] - sum_{s: ext[s]=c} alpha_t(s) beta_t(s) / (p * y_tc)
    ab = alpha + beta  # contains lp[t,ext] twice
    post = np.zeros((in_len, V))
    for s in range(S):
        with np.errstate(over="ignore"):
            post[:, ext[s]] += np.exp(ab[:, s] - lp[:, ext[s]] - ll)
    grad[:in_len] = np.exp(lp) - post
    return nll, grad


def ctc_loss_and_grad(logits: np.ndarray, labels: np.ndarray, in_lens, blank: int = 0,
                      reduction: str = "mean", zero_infinity: bool = True):
    """Batched.  logits [B,F,V]; labels [B,Lmax] padded with negative values (-100,
    wav2vec_train.py:100; target length = count(label >= 0), modeling_wav2vec2.py:1712-1714).
    Returns (loss scalar, dloss/dlogits [B,F,V], per-utterance nll [B])."""
    B = logits.shape[0]
    nll = np.zeros(B)
    grads = np.zeros(logits.shape, dtype=np.float64)
    tl = np.zeros(B, dtype=np.int64)
    for b in range(B):
        tgt = labels[b][labels[b] >= 0]
        tl[b] = len(tgt)
        nll[b], grads[b] = ctc_single(np.asarray(logits[b], dtype=np.float64), tgt, int(in_lens[b]), blank)
    inf = ~np.isfinite(nll)
    if zero_infinity:
        nll = np.where(inf, 0.0, nll)
        grads[inf] = 0.0
    if reduction == "mean":
        w = 1.0 / (np.maximum(tl, 1) * B)
    elif reduction == "sum":
        w = np.ones(B)
    else:
        raise ValueError(reduction)
    return float((nll * w).sum()), grads * w[:, None, None], nll
