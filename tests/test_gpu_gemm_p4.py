"""The four-wave hand-scheduled GEMM (ssak_amd/csrc/gemm_p4.hip) against an fp32 matmul: bit-exact on integer-valued operands
(every partial sum is exact in fp32, so ANY summation order must give the same bits -- a fragment map, a swizzle, a stale LDS
slot or a mis-counted wait cannot).  Every case also checks that the product really ran on gemm_p4_kernel (the dispatcher
falls back to the eight-wave kernel silently otherwise).  Math replaced: the Linear layers of Wav2Vec2FeedForward /
Wav2Vec2Attention reached from ssak/train/transformers/wav2vec_train.py:415."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import ssak_amd.hip as h
    return h


def _ran_on_p4(hip, fn):
    hip.prof_enable(1)
    hip.prof_collect()
    fn()
    torch.cuda.synchronize()
    hip.prof_enable(0)
    names = [e[0] for e in hip.prof_collect() if e[1] > 0]
    assert any(n.startswith("gemm_p4_kernel") for n in names), names
    return names


def _operands(M, N, K, seed, lo=-2, hi=3, nb=1):
    g = torch.Generator().manual_seed(seed)
    A = torch.randint(lo, hi, (nb * M, K), generator=g).to(torch.bfloat16)
    W = torch.randint(lo, hi, (N, K), generator=g).to(torch.bfloat16)
    return A, W


# (M, N, K, tile height; 0 = the library's choice): one tile with a row tail; the train step's products as the library plans them --
# the 192-row form in one round, three rounds of 256-row tiles with every workgroup re-priming (756 tiles on 256 CUs) and a last
# tile row of 96 valid rows --; deep K; K = 256 (the shortest pipeline: no steady-state K tile at all); every tile height with
# row tails that leave whole waves without rows
SHAPES = [(300, 256, 256, 256), (15968, 768, 768, 0), (15968, 3072, 768, 0), (4000, 768, 3072, 192), (8193, 512, 256, 128),
          (3077, 1024, 1024, 256), (15968, 2304, 768, 0), (70000, 256, 256, 128), (33000, 512, 384, 192), (257, 256, 512, 128),
          # odd K-tile counts (3, 5, 7): three A stages re-point to the next output tile inside the first K tile, and the B-stage
          # parity carries over an odd count of K tiles from one output tile to the next; several rounds of tiles per workgroup
          (40000, 512, 192, 128), (40000, 512, 192, 192), (70000, 256, 192, 256), (50000, 768, 320, 192), (66000, 512, 448, 256),
          (33000, 512, 448, 128)]


@pytest.mark.parametrize("M,N,K,tile", SHAPES)
def test_p4_plain_bf16_bit_exact(hip, M, N, K, tile):
    A, W = _operands(M, N, K, M + N + K)
    bias = torch.randint(-4, 5, (N,), generator=torch.Generator().manual_seed(1)).float()
    ref = (A.float() @ W.float().T + bias).to(torch.bfloat16)
    C = torch.full((M, N), float("nan"), dtype=torch.bfloat16).cuda()
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    _ran_on_p4(hip, lambda: hip.gemm(Ad, Wd, C, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, plan_tile=tile))
    assert torch.equal(C.cpu(), ref), (C.cpu().float() - ref.float()).abs().max()
    # the same launch again and again: a persistent workgroup's hand-off between output tiles must not depend on what the
    # previous launch left in LDS / in flight
    for _ in range(3):
        C.fill_(float("nan"))
        hip.gemm(Ad, Wd, C, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, plan_tile=tile)
        assert torch.equal(C.cpu(), ref)


def test_p4_fp32_out_alpha_batches(hip):
    """General epilogue form: fp32 output, alpha, accumulate, a batch of 3 operand slices (one weight)."""
    M, N, K, nb = 1100, 512, 384, 3
    A, W = _operands(M, N, K, 7, nb=nb)
    ref = (A.float().view(nb, M, K) @ W.float().T) * 0.5
    C = torch.ones(nb, M, N, dtype=torch.float32).cuda()
    Ad, Wd = A.cuda(), W.cuda()
    _ran_on_p4(hip, lambda: hip.gemm(Ad, Wd, C, M, N, K, lda=K, ldb=K, ldc=N, nb1=nb, sa=(M * K, 0), sc=(M * N, 0), alpha=0.5,
                                     accumulate=True, plan_tile=192))
    assert torch.equal(C.cpu(), ref + 1.0)


def test_p4_toeplitz_rows(hip):
    """Overlapping A rows (channels-last Conv1d as a GEMM: lda < K) with K = 2 * 512 (the k = 2 layers of the conv stack)."""
    g = torch.Generator().manual_seed(3)
    Bn, Tin, Cc, Co, k, s = 2, 2001, 512, 512, 2, 2
    Tout = (Tin - k) // s + 1
    x = torch.randint(-2, 3, (Bn, Tin, Cc), generator=g).to(torch.bfloat16)
    w = torch.randint(-1, 2, (Co, Cc, k), generator=g).to(torch.bfloat16)
    ref = torch.nn.functional.conv1d(x.float().transpose(1, 2), w.float(), stride=s).transpose(1, 2)
    wk = w.permute(0, 2, 1).contiguous().view(Co, k * Cc)
    y = torch.empty(Bn, Tout, Co, dtype=torch.float32).cuda()
    xd, wd = x.cuda(), wk.cuda()
    _ran_on_p4(hip, lambda: hip.gemm(xd, wd, y, Tout, Co, k * Cc, lda=s * Cc, ldb=k * Cc, ldc=Co, nb1=Bn, sa=(Tin * Cc, 0), sc=(Tout * Co, 0), plan_tile=192))
    assert torch.equal(y.cpu(), ref)


def test_p4_mul_aux_feed_forward_dx(hip):
    """Feed-forward backward form on the four-wave kernel (dI = (dY W2) * f with f the 8-bit gelu' factor code of the forward,
    plus column sums = the bias gradient), at the headline shape, against fp32 torch on the decoded factor.  The epilogue code
    is shared with the eight-wave kernel; this checks that the four-wave kernel hands it the right accumulators, factor codes
    (preloaded before the last K tile drains) and coordinates (two 64-column halves per wave).  The forward half of the pair
    (SSAK_EPI_GELU_SAVE_GRAD) stays on the eight-wave kernel (profiles/r04_ab_ffn_up_kernel.log) and is covered by
    tests/test_gpu_ops.py::test_gemm_feed_forward_epilogue_pair."""
    M, N, K = 15968, 3072, 768
    g = torch.Generator().manual_seed(11)
    dY = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W2 = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    f8 = torch.randint(0, 256, (M, N), generator=g, dtype=torch.int32).to(torch.uint8).cuda()
    dI = torch.full((M, N), float("nan"), dtype=torch.bfloat16).cuda()
    cs = torch.zeros(N, dtype=torch.float32).cuda()
    _ran_on_p4(hip, lambda: hip.gemm(dY, W2, dI, M, N, K, lda=K, ldb=K, ldc=N, epilogue=hip.EPI_MUL_AUX, aux_in=f8, colsum_out=cs,
                                     drop_p=0.1))
    f = (f8.float() - 26.0) * (1.26 / 254 / 0.9)
    ref = (dY.float() @ W2.float().T) * f
    rel = float((dI.float() - ref).norm() / ref.norm())
    assert rel < 4e-3, rel
    assert float((dI.float() - ref).abs().max()) < 0.02 * float(ref.abs().max())
    want_cs = dI.float().sum(0)
    assert bool(((cs - want_cs).abs() <= 2e-3 * dI.float().abs().sum(0) + 1e-3).all())


@pytest.mark.parametrize("M,N,K,tile", [(15968, 3072, 768, 0), (15968, 2304, 768, 0), (40000, 768, 384, 192), (15968, 768, 3072, 0)])
def test_p4_ticket_tile_order_bit_exact(hip, M, N, K, tile):
    """ssak_gemm_desc.dynamic_tiles on the four-wave kernel (what the data-parallel trainers switch on): every tile drawn from
    the per-XCD ticket counters, the next tile's ticket requested inside K tile 0 and read three K tiles later.  Multi-round
    shapes, integer operands (bit-exact whatever the order), repeated launches on one stream (every launch leaves its counters
    at zero) and two streams at once (separate counter slots)."""
    A, W = _operands(M, N, K, M + 3 * N + K)
    bias = torch.randint(-4, 5, (N,), generator=torch.Generator().manual_seed(2)).float()
    ref = (A.float() @ W.float().T + bias).to(torch.bfloat16)
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    C = torch.full((M, N), float("nan"), dtype=torch.bfloat16).cuda()
    _ran_on_p4(hip, lambda: hip.gemm(Ad, Wd, C, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, plan_tile=tile, dynamic_tiles=True))
    assert torch.equal(C.cpu(), ref)
    for _ in range(3):
        C.fill_(float("nan"))
        hip.gemm(Ad, Wd, C, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, plan_tile=tile, dynamic_tiles=True)
        assert torch.equal(C.cpu(), ref)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    C1, C2 = torch.full_like(C, float("nan")), torch.full_like(C, float("nan"))
    torch.cuda.synchronize()
    for _ in range(2):
        with torch.cuda.stream(s1):
            hip.gemm(Ad, Wd, C1, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, plan_tile=tile, dynamic_tiles=True)
        with torch.cuda.stream(s2):
            hip.gemm(Ad, Wd, C2, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, plan_tile=tile, dynamic_tiles=True)
    torch.cuda.synchronize()
    assert torch.equal(C1.cpu(), ref) and torch.equal(C2.cpu(), ref)
