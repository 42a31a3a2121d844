"""GPU: forced alignment through the C ABI against the torch-op goldens and the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = ["tiny", "base", "blank5", "garbage", "flat", "tight", "infeasible"]


@pytest.fixture(scope="module")
def align():
    from ssak_amd import align as A
    return A


@pytest.mark.parametrize("name", CASES)
def test_align_golden_bit_exact(align, gold, name):
    """Trellis identical bit for bit (fp32 add / max, double cumsum of column 0, the inf borders), same path."""
    z = gold("align.npz")
    em = torch.from_numpy(z[f"{name}_emission"]).cuda()
    tok, blank = z[f"{name}_tokens"].tolist(), int(z[f"{name}_blank"])
    trellis = align.get_trellis(em, tok, blank_id=blank, first_as_garbage=bool(z[f"{name}_garbage"]))
    got, want = trellis.cpu().numpy(), z[f"{name}_trellis"]
    if name == "garbage":
        # column 0 of this variant is log(1 - exp(.)): torch's vectorised exp / log differ in the last ulp between host
        # CPUs (the golden was written on another machine), so only this case is compared with a tolerance
        fin = np.isfinite(want)
        assert np.array_equal(fin, np.isfinite(got)) and np.array_equal(got[~fin], want[~fin])
        assert np.allclose(got[fin], want[fin], rtol=2e-6, atol=1e-5)
    else:
        assert np.array_equal(got, want)
    if not int(z[f"{name}_ok"]):
        with pytest.raises(RuntimeError, match="Failed to align"):
            align.backtrack(trellis, em, tok, blank_id=blank)
        return
    path = align.backtrack(trellis, em, tok, blank_id=blank)
    assert [p.token_index for p in path] == z[f"{name}_path_token"].tolist()
    assert [p.time_index for p in path] == z[f"{name}_path_time"].tolist()
    assert np.allclose([p.score for p in path], z[f"{name}_path_score"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("F,V,L", [(700, 40, 300), (1500, 32, 1100), (3000, 51, 2500)])
def test_align_vs_oracle_multi_wave(align, F, V, L):
    """Transcripts wider than one wave / one workgroup pass (several columns per thread) against the CPU oracle."""
    from oracle import align_ref
    g = torch.Generator().manual_seed(F + L)
    logits = torch.randn(F, V, generator=g) * 2
    tok = torch.randint(1, V, (L,), generator=g).tolist()
    pos = sorted(torch.randperm(F, generator=g)[:L].tolist())
    logits[:, 0] += 3.0
    for k, p in enumerate(pos):
        logits[p, tok[k]] += 8.0
    em = torch.log_softmax(logits, -1)
    ref_tr = align_ref.get_trellis(em.numpy(), tok, 0)
    ref_path = align_ref.backtrack(ref_tr, em.numpy(), tok, 0)
    trellis, path = align.forced_align(em.cuda(), tok, 0)
    assert np.array_equal(trellis.cpu().numpy(), ref_tr)
    assert [(p.token_index, p.time_index) for p in path] == [(p.token_index, p.time_index) for p in ref_path]
    assert np.allclose([p.score for p in path], [p.score for p in ref_path], rtol=1e-6)


def test_align_long_audio_properties(align):
    """A long recording (12 000 frames, 3 000 characters; K = 3 columns per thread): checks that need no CPU trellis loop.
    The recursion is verified cell by cell in one vectorised expression (every row from the row above it, bit-exact),
    column 0 and the borders directly, and the walk by the oracle's backtrack run on the DEVICE trellis."""
    from oracle import align_ref
    F, V, L = 12000, 32, 3000
    g = torch.Generator().manual_seed(9)
    logits = torch.randn(F, V, generator=g)
    tok = torch.randint(1, V, (L,), generator=g).tolist()
    pos = sorted(torch.randperm(F, generator=g)[:L].tolist())
    logits[:, 0] += 3.0
    for k, p in enumerate(pos):
        logits[p, tok[k]] += 9.0
    em_h = torch.log_softmax(logits, -1)
    trellis, path = align.forced_align(em_h.cuda(), tok, 0)
    tr, em = trellis.cpu().numpy(), em_h.numpy()
    et = em[:, tok]
    want = np.maximum(tr[:-1, 1:] + em[:, :1], np.maximum(tr[:-1, 1:] + et, tr[:-1, :-1] + et))
    assert np.array_equal(tr[1:, 1:], want)
    c0 = np.cumsum(em[:, 0].astype(np.float64)).astype(np.float32)
    assert tr[0, 0] == 0 and np.array_equal(tr[1:F + 1 - L, 0], c0[:F - L]) and np.isinf(tr[F + 1 - L:, 0]).all()
    assert (tr[0, 1:] == -np.inf).all() and np.isfinite(tr[F, L])
    ref = align_ref.backtrack(tr, em, tok, 0)
    assert [(p.token_index, p.time_index) for p in path] == [(p.token_index, p.time_index) for p in ref]
    assert np.allclose([p.score for p in path], [p.score for p in ref], rtol=1e-6)
    ti = np.array([p.token_index for p in path])
    tt = np.array([p.time_index for p in path])
    assert ti[0] == 0 and ti[-1] == L - 1 and set(np.diff(ti).tolist()) <= {0, 1} and (np.diff(tt) == 1).all()
    assert tt[-1] + 1 == int(np.argmax(tr[:, L]))
    segs = align.merge_repeats("".join(chr(97 + t % 26) for t in tok), path)
    assert len(segs) == L and all(a.end == b.start for a, b in zip(segs, segs[1:]))


def test_align_rejects_bad_tokens(align):
    em = torch.log_softmax(torch.randn(20, 8), -1).cuda()
    with pytest.raises(IndexError):
        align.forced_align(em, [1, 9, 2], 0)
    with pytest.raises(IndexError):
        align.forced_align(em, [], 0)
