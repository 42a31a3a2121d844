"""numpy restatement of the engine's counter-hash dropout masks.  TEST INFRASTRUCTURE ONLY.

The HIP engine never stores a dropout mask: every site recomputes ``keep(seed, site, element)`` in its forward and
backward kernels (ssak_amd/csrc/common.h ``hash_u32`` / ``keep_bit``; ssak_amd/csrc/attention.hip ``drop_rowseed`` /
``drop_word``).  This file restates those integer functions so that ``oracle/gen_golden_dropout.py`` can hand the SAME masks
to ``transformers.Wav2Vec2ForCTC`` in ``train()`` mode (its ``nn.Dropout`` modules and the ``nn.functional.dropout`` call of
``eager_attention_forward`` are patched to consume them) -- which pins WHERE each of the seven sites sits, and its
``1 / (1 - p)`` scale, against the third-party model the reference trains
(``ssak/train/transformers/wav2vec_train.py:161-165,313-325``; transformers ``modeling_wav2vec2.py:429-434`` feature projection,
``:458`` attention probabilities, ``:565-572`` feed-forward (activation + hidden), ``:591-608`` / ``:631-654`` layer residual,
``:692`` / ``:765`` encoder input, ``:1697-1700`` head).  Pinned bit-for-bit against the device functions by
``tests/test_gpu_dropout.py::test_dropout_hash_matches_device`` through the debug C-ABI entries ``ssak_debug_dropout_mask`` /
``ssak_debug_attention_dropout_mask``.

Site ("stream") ids follow ssak_amd/csrc/w2v2_engine.hip:127-131.
"""
from __future__ import annotations

import numpy as np

DS_FEATPROJ, DS_ENCIN, DS_FINAL, DS_LAYER0 = 1, 2, 3, 16
DROP_PHI = 0x9E3779B9
_M32 = np.uint64(0xFFFFFFFF)


def ds_attn(l: int) -> int:
    return DS_LAYER0 + 4 * l


def ds_hid1(l: int) -> int:
    return DS_LAYER0 + 4 * l + 1


def ds_act(l: int) -> int:
    return DS_LAYER0 + 4 * l + 2


def ds_hid2(l: int) -> int:
    return DS_LAYER0 + 4 * l + 3


def _mul32(a, b):
    return (a.astype(np.uint64) * np.uint64(b)) & _M32


def hash_u32(seed: int, stream: int, idx) -> np.ndarray:
    """common.h ``hash_u32``: the "lowbias32" mixer of (idx ^ key(seed, stream, idx >> 32)); uint64 arrays holding 32-bit values."""
    idx = np.asarray(idx, dtype=np.uint64)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    key = (seed & 0xFFFFFFFF) ^ (((seed >> 32) * 0x9E3779B1) & 0xFFFFFFFF) ^ ((int(stream) * 0x85EBCA77) & 0xFFFFFFFF)
    key = np.uint64(key) ^ _mul32(idx >> np.uint64(32), 0xC2B2AE3D)
    h = (idx & _M32) ^ key
    h ^= h >> np.uint64(16)
    h = _mul32(h, 0x7FEB352D)
    h ^= h >> np.uint64(15)
    h = _mul32(h, 0x846CA68B)
    h ^= h >> np.uint64(16)
    return h


def thresh16(p: float) -> int:
    """16-bit threshold of a drop probability: round-half-away-from-zero of p * 65536 in fp32, capped at 65535
    (gemm.hip ``drop_thresh``, norm_act.hip ``thresh_of``, attention.hip)."""
    if p <= 0:
        return 0
    v = np.float32(p) * np.float32(65536.0)
    return int(min(65535.0, np.floor(np.float64(v) + 0.5)))


def engine_scale(p: float) -> float:
    """What the kernels multiply kept elements by: 1 / (1 - thresh16 / 65536), the REALISED keep probability (fp32)."""
    t = thresh16(p)
    return float(np.float32(1.0) / (np.float32(1.0) - np.float32(t) / np.float32(65536.0))) if t else 1.0


def keep_mask(seed: int, stream: int, shape, p: float) -> np.ndarray:
    """common.h ``keep_bit`` over the flat row-major element offsets of ``shape``: one hash per element PAIR
    (low 16 bits -> even offset, high 16 bits -> odd offset), keep iff the field >= thresh16(p)."""
    n = int(np.prod(shape))
    t = thresh16(p)
    if t == 0:
        return np.ones(shape, dtype=bool)
    idx = np.arange(n, dtype=np.uint64)
    w = hash_u32(seed, stream, idx >> np.uint64(1))
    u = np.where((idx & np.uint64(1)) == 1, w >> np.uint64(16), w & np.uint64(0xFFFF))
    return (u >= np.uint64(t)).reshape(shape)


def attention_keep_mask(seed: int, stream: int, B: int, nh: int, F: int, p: float, Fk: int | None = None) -> np.ndarray:
    """attention.hip: row (b, h, q) has seed hash_u32(seed, stream, (b * nh + h) * F + q); the word of key pair kp of that row
    is one multiply-xorshift round of (row seed + kp * golden ratio); even key <- low 16 bits, odd key <- high 16 bits.
    Returns bool [B, nh, F, Fk] (Fk = F unless given)."""
    Fk = F if Fk is None else Fk
    t = thresh16(p)
    if t == 0:
        return np.ones((B, nh, F, Fk), dtype=bool)
    rows = np.arange(B * nh * F, dtype=np.uint64)
    rs = hash_u32(seed, stream, rows)[:, None]
    k = np.arange(Fk, dtype=np.uint64)[None, :]
    x = (rs + _mul32(k >> np.uint64(1), DROP_PHI)) & _M32
    x ^= x >> np.uint64(15)
    x = _mul32(x, 0x2C1B3C6D)
    x ^= x >> np.uint64(12)
    u = np.where((k & np.uint64(1)) == 1, x >> np.uint64(16), x & np.uint64(0xFFFF))
    return (u >= np.uint64(t)).reshape(B, nh, F, Fk)


def step_seed_sequence(seed: int, n: int):
    """The per-forward dropout seeds ``ssak_amd.model.Wav2Vec2ForCTC`` derives from its constructor seed: state 0 from
    ``np.random.SeedSequence(seed)``, then one 64-bit LCG step per forward (the value after the step is the one used)."""
    s = int(np.random.SeedSequence(int(seed)).generate_state(1, dtype=np.uint64)[0])
    out = []
    for _ in range(n):
        s = (s * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        out.append(s)
    return out
