"""numpy restatement of the engine's counter-hash dropout masks.  TEST INFRASTRUCTURE ONLY.

The HIP engine never stores a dropout mask: every site recomputes ``keep(seed, site, row, column)`` in its forward and
backward kernels (ssak_amd/csrc/common.h ``drop_rowkey`` / ``drop_colmul`` / ``drop_keep``).  Since round 5 ONE definition
serves every site::

    word(row, col) = rowkey(seed, site, row) * colmul(col)  mod 2^32          keep  iff  word >= thresh16(p) << 16
    rowkey = hash_u32(seed, site, row) | 1          colmul = hash_u32(COL_SEED, COL_STREAM, col) | 1

with (row, col) of the site's row-major ``[rows, cols]`` tensor and, for the attention probabilities, row = (b * nh + h) * F + q,
col = key.  (Rounds 1-4 drew a 16-bit field of a two-multiply hash per element pair; the mask generator is the build's own --
torch's generator cannot be reproduced on the device and the reference asks for no particular bits -- so it was replaced by
the cheapest form that passes ``quality_report``: one full-rate integer multiply per element.)  This file restates those
integer functions so that ``oracle/gen_golden_dropout.py`` can hand the SAME masks to ``transformers.Wav2Vec2ForCTC`` in
``train()`` mode (its ``nn.Dropout`` modules and the ``nn.functional.dropout`` call of ``eager_attention_forward`` are patched to
consume them) -- which pins WHERE each of the seven sites sits, and its ``1 / (1 - p)`` scale, against the third-party model the
reference trains (``ssak/train/transformers/wav2vec_train.py:161-165,313-325``; transformers ``modeling_wav2vec2.py:429-434``
feature projection, ``:458`` attention probabilities, ``:565-572`` feed-forward (activation + hidden), ``:591-608`` / ``:631-654``
layer residual, ``:692`` / ``:765`` encoder input, ``:1697-1700`` head).  Pinned bit-for-bit against the device functions by
``tests/test_gpu_dropout.py::test_dropout_hash_matches_device`` through the debug C-ABI entries ``ssak_debug_dropout_mask`` /
``ssak_debug_attention_dropout_mask``.

Site ("stream") ids follow ssak_amd/csrc/w2v2_engine.hip:127-131.
"""
from __future__ import annotations

import numpy as np

DS_FEATPROJ, DS_ENCIN, DS_FINAL, DS_LAYER0 = 1, 2, 3, 16
COL_SEED, COL_STREAM = 0x5EED0C01A11CE5, 0x51  # common.h DROP_COL_SEED / DROP_COL_STREAM
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


def rowkey(seed: int, stream: int, rows) -> np.ndarray:
    """common.h ``drop_rowkey``: the mixer of the row index under (seed, site), forced odd."""
    return hash_u32(seed, stream, rows) | np.uint64(1)


def colmul(cols) -> np.ndarray:
    """common.h ``drop_colmul``: the mixer of the column (key) index under a fixed key, forced odd."""
    return hash_u32(COL_SEED, COL_STREAM, cols) | np.uint64(1)


def keep_mask(seed: int, stream: int, shape, p: float) -> np.ndarray:
    """keep[row, col] over the row-major view ``[prod(shape[:-1]), shape[-1]]`` of ``shape``: word = rowkey(row) * colmul(col)
    mod 2^32, keep iff word >= thresh16(p) << 16."""
    shape = tuple(int(v) for v in shape)
    t = thresh16(p)
    if t == 0:
        return np.ones(shape, dtype=bool)
    cols = shape[-1]
    rows = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
    rk = rowkey(seed, stream, np.arange(rows, dtype=np.uint64))[:, None]
    cm = colmul(np.arange(cols, dtype=np.uint64))[None, :]
    w = (rk * cm) & _M32
    return (w >= np.uint64(t << 16)).reshape(shape)


def attention_keep_mask(seed: int, stream: int, B: int, nh: int, F: int, p: float, Fk: int | None = None) -> np.ndarray:
    """attention.hip: the same function with row = (b * nh + h) * F + q and col = key.  bool [B, nh, F, Fk] (Fk = F unless given)."""
    return keep_mask(seed, stream, (B * nh * F, F if Fk is None else Fk), p).reshape(B, nh, F, F if Fk is None else Fk)


def quality_report(p: float, rows: int = 4096, cols: int = 3072, seed: int = 1234, site: int = 17) -> dict:
    """Statistics the mask generator is held to (tests/test_oracle.py): drop rate, correlation of the drop indicators between
    neighbouring columns / rows / diagonal neighbours / another site / another seed, and the variance of the per-row and
    per-column drop counts relative to a binomial.  ``sigma`` is the iid standard error of a correlation over rows * cols samples;
    the rank-one structure of the words (rows + cols random numbers) makes the real spread a few times that."""
    d = (~keep_mask(seed, site, (rows, cols), p)).astype(np.float64)
    rate = d.mean()
    d -= p
    v = p * (1 - p)

    def corr(x, y):
        return float((x * y).mean() / v)
    d2 = (~keep_mask(seed, site + 1, (rows, cols), p)).astype(np.float64) - p
    d3 = (~keep_mask(seed + 1, site, (rows, cols), p)).astype(np.float64) - p
    cnt_r, cnt_c = (d + p).sum(1), (d + p).sum(0)
    return {"rate": float(rate), "rate_sigma": float(np.sqrt(v / d.size)), "sigma": float(1 / np.sqrt(d.size)),
            "col1": corr(d[:, :-1], d[:, 1:]), "col2": corr(d[:, :-2], d[:, 2:]), "col64": corr(d[:, :-64], d[:, 64:]),
            "row1": corr(d[:-1], d[1:]), "diag": corr(d[:-1, :-1], d[1:, 1:]), "site": corr(d, d2), "seed": corr(d, d3),
            "row_count_var": float(cnt_r.var() / (cols * v)), "col_count_var": float(cnt_c.var() / (rows * v))}


def inverse_mod_2_32(a) -> np.ndarray:
    """Multiplicative inverses of odd 32-bit values mod 2^32 (Newton iteration: the number of correct bits doubles per step)."""
    a = np.asarray(a, dtype=np.uint64)
    x = a.copy()  # correct to 3 bits for odd a
    for _ in range(5):
        x = (x * ((np.uint64(2) - ((a * x) & _M32)) & _M32)) & _M32
    return x


def smallest_key_ratio_pairs(seed: int, site: int, rows: int, top: int = 1000):
    """The rank-one weakness made concrete: two rows' words differ by ONE factor for every column, word_i = r word_j with
    r = rowkey_i * rowkey_j^-1 (mod 2^32).  Returns the `top` ordered row pairs (i, j) with the smallest |r| (r read as a signed 32-bit
    number) as arrays (r_signed, i, j), ascending in |r|: the pairs whose masks are closest to being functions of each other."""
    rk = rowkey(seed, site, np.arange(rows, dtype=np.uint64))
    iv = inverse_mod_2_32(rk)
    two32 = np.uint64(1) << np.uint64(32)
    out_m, out_r, out_i, out_j = [], [], [], []
    step = 512
    for s in range(0, rows, step):
        r = (rk[s:s + step, None] * iv[None, :]) & _M32
        mag = np.minimum(r, two32 - r)
        ii = np.arange(s, min(rows, s + step))
        mag[ii - s, ii] = two32  # (a row with itself)
        flat = mag.ravel()
        k = min(top, flat.size)
        idx = np.argpartition(flat, k - 1)[:k]
        out_m.append(flat[idx])
        out_r.append(r.ravel()[idx])
        out_i.append(idx // rows + s)
        out_j.append(idx % rows)
    m, r, i, j = (np.concatenate(v) for v in (out_m, out_r, out_i, out_j))
    o = np.argsort(m, kind="stable")[:top]
    r = r[o].astype(np.int64)
    r = np.where(r >= (1 << 31), r - (1 << 32), r)
    return r, i[o], j[o]


def keep_agreement(seed: int, site: int, rows_i, rows_j, cols: int, p: float) -> np.ndarray:
    """Fraction of the `cols` columns on which rows_i[k] and rows_j[k] of the site have the same keep bit."""
    cm = colmul(np.arange(cols, dtype=np.uint64))[None, :]
    t = np.uint64(thresh16(p) << 16)
    ki = ((rowkey(seed, site, np.asarray(rows_i, dtype=np.uint64))[:, None] * cm) & _M32) >= t
    kj = ((rowkey(seed, site, np.asarray(rows_j, dtype=np.uint64))[:, None] * cm) & _M32) >= t
    return (ki == kj).mean(1)


def ratio_agreement_expected(r: int, p: float) -> float:
    """Keep-bit agreement of two rows whose words satisfy word_i = r word_j (mod 2^32), r a small odd integer with |r| p < 1, word_j
    uniform: for r > 0 both drop when word_j < thresh / r (probability p / r), for r < 0 never; independence would give p^2."""
    pe = thresh16(p) / 65536.0
    joint_drop = pe / r if r > 0 else 0.0
    return 1.0 - 2.0 * pe + 2.0 * joint_drop


def step_seed_sequence(seed: int, n: int):
    """The per-forward dropout seeds ``ssak_amd.model.Wav2Vec2ForCTC`` derives from its constructor seed: state 0 from
    ``np.random.SeedSequence(seed)``, then one 64-bit LCG step per forward (the value after the step is the one used)."""
    s = int(np.random.SeedSequence(int(seed)).generate_state(1, dtype=np.uint64)[0])
    out = []
    for _ in range(n):
        s = (s * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        out.append(s)
    return out
