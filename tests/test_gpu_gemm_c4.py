"""The co-resident four-wave GEMM (ssak_amd/csrc/gemm_c4.hip: 128 x 256 tiles, two workgroups per CU, 32-deep K tiles in three LDS
stages) against an fp32 matmul: bit-exact on integer-valued operands (every partial sum is exact in fp32, so ANY summation
order must give the same bits -- a fragment map, a swizzle, a stale LDS stage or a mis-counted wait cannot), and its two
feed-forward epilogues against fp32 torch.  Every case checks that the product really ran on gemm_c4_kernel.  Math replaced:
the Linear layers of Wav2Vec2FeedForward / Wav2Vec2Attention reached from ssak/train/transformers/wav2vec_train.py:415
(SURVEY.md section 8, a7)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
CORES = 129  # SSAK_PLAN_TILE_CORESIDENT


@pytest.fixture(scope="module")
def hip():
    import ssak_amd.hip as h
    return h


def _ran_on_c4(hip, fn):
    hip.prof_enable(1)
    hip.prof_collect()
    fn()
    torch.cuda.synchronize()
    hip.prof_enable(0)
    names = [e[0] for e in hip.prof_collect() if e[1] > 0]
    assert any(n.startswith("gemm_c4_kernel") for n in names), names
    return names


def _operands(M, N, K, seed, lo=-2, hi=3):
    g = torch.Generator().manual_seed(seed)
    A = torch.randint(lo, hi, (M, K), generator=g).to(torch.bfloat16)
    W = torch.randint(lo, hi, (N, K), generator=g).to(torch.bfloat16)
    return A, W


# one tile with a row tail; the train step's products (three rounds of 512 workgroups with every workgroup re-priming, a last tile
# row of 96 valid rows); deep K; K = 192 / 256 / 320 (the shortest pipelines: six, eight and ten K tiles); row tails that leave
# whole waves without rows; fewer tiles than workgroups (the dispatcher keeps M < 256 on the small-tile kernels)
SHAPES = [(300, 256, 256), (15968, 768, 768), (15968, 3072, 768), (4000, 768, 3072), (8193, 512, 192), (3077, 1024, 1024),
          (15968, 2304, 768), (70000, 256, 320), (33000, 512, 384), (257, 256, 512), (385, 512, 448)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_c4_plain_bf16_bit_exact(hip, M, N, K):
    A, W = _operands(M, N, K, M + N + K)
    bias = torch.randint(-4, 5, (N,), generator=torch.Generator().manual_seed(1)).float()
    ref = (A.float() @ W.float().T + bias).to(torch.bfloat16)
    C = torch.full((M, N), float("nan"), dtype=torch.bfloat16).cuda()
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    _ran_on_c4(hip, lambda: hip.gemm(Ad, Wd, C, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, plan_tile=CORES))
    assert torch.equal(C.cpu(), ref), (C.cpu().float() - ref.float()).abs().max()
    # the same launch again and again: a persistent workgroup's hand-off between output tiles must not depend on what the
    # previous launch left in LDS / in flight
    for _ in range(3):
        C.fill_(float("nan"))
        hip.gemm(Ad, Wd, C, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, plan_tile=CORES)
        assert torch.equal(C.cpu(), ref)
    # no bias (a null descriptor: zeros)
    C.fill_(float("nan"))
    hip.gemm(Ad, Wd, C, M, N, K, lda=K, ldb=K, ldc=N, plan_tile=CORES)
    assert torch.equal(C.cpu(), (A.float() @ W.float().T).to(torch.bfloat16))


def test_c4_batched_slices(hip):
    """A batch of 3 operand slices (one weight), output row stride wider than N."""
    M, N, K, nb, ldc = 1100, 512, 384, 3, 640
    g = torch.Generator().manual_seed(7)
    A = torch.randint(-2, 3, (nb * M, K), generator=g).to(torch.bfloat16)
    W = torch.randint(-2, 3, (N, K), generator=g).to(torch.bfloat16)
    ref = (A.float().view(nb, M, K) @ W.float().T).to(torch.bfloat16)
    C = torch.full((nb, M, ldc), float("nan"), dtype=torch.bfloat16).cuda()
    Ad, Wd = A.cuda(), W.cuda()
    _ran_on_c4(hip, lambda: hip.gemm(Ad, Wd, C, M, N, K, lda=K, ldb=K, ldc=ldc, nb1=nb, sa=(M * K, 0), sc=(M * ldc, 0), plan_tile=CORES))
    assert torch.equal(C[:, :, :N].cpu(), ref)


def test_c4_feed_forward_pair(hip):
    """Both feed-forward epilogues on the co-resident kernel at the headline shape: the up-projection (GELU, the saved 8-bit factor
    f = gelu'(x) keep / (1 - p), dropout bits = oracle.dropout_hash.keep_mask) bit-identical to the eight-wave kernel's output for
    the same descriptor, and the backward product (dY W2) * f with its column sums against fp32 torch on the decoded factor."""
    from oracle import dropout_hash as DH
    M, N, K = 15968, 3072, 768
    g = torch.Generator().manual_seed(11)
    X = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
    W1 = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    b1 = (torch.randn(N, generator=g) * 0.1).float().cuda()
    kw = dict(lda=K, ldb=K, ldc=N, bias=b1, drop_p=0.1, drop_stream=DH.ds_act(2), drop_seed=0xFEEDFACE12345)
    y0, f0 = torch.empty(M, N, dtype=torch.bfloat16).cuda(), torch.empty(M, N, dtype=torch.uint8).cuda()
    hip.gemm(X, W1, y0, M, N, K, epilogue=hip.EPI_GELU_SAVE_GRAD, aux_out=f0, plan_tile=256, **kw)
    y1, f1 = torch.full_like(y0, float("nan")), torch.zeros_like(f0)
    _ran_on_c4(hip, lambda: hip.gemm(X, W1, y1, M, N, K, epilogue=hip.EPI_GELU_SAVE_GRAD, aux_out=f1, plan_tile=CORES, **kw))
    # (the K rotation differs between the kernels: fp32 summation order, i.e. the last bf16 bit of a few outputs)
    assert float((y1.float() - y0.float()).abs().max()) <= 2.0 ** -7 * float(y0.float().abs().max())
    keep = torch.from_numpy(DH.keep_mask(kw["drop_seed"], kw["drop_stream"], (M, N), 0.1)).cuda()
    # dropped elements: code 26 (= exactly 0) and a zero output; kept ones: the code of gelu'(x) (26 only where gelu' rounds to 0)
    assert bool((f1[~keep] == 26).all()) and bool((y1[~keep] == 0).all())
    assert float((f1[keep] != 26).float().mean()) > 0.98 and float((y1[keep] != 0).float().mean()) > 0.98
    assert int((f1.int() - f0.int()).abs().max()) <= 1
    # backward
    dY = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W2 = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    dI = torch.full((M, N), float("nan"), dtype=torch.bfloat16).cuda()
    cs = torch.zeros(N, dtype=torch.float32).cuda()
    _ran_on_c4(hip, lambda: hip.gemm(dY, W2, dI, M, N, K, lda=K, ldb=K, ldc=N, epilogue=hip.EPI_MUL_AUX, aux_in=f1, colsum_out=cs,
                                     drop_p=0.1, plan_tile=CORES))
    f = (f1.float() - 26.0) * (1.26 / 254 / 0.9)
    ref = (dY.float() @ W2.float().T) * f
    rel = float((dI.float() - ref).norm() / ref.norm())
    assert rel < 4e-3, rel
    assert float((dI.float() - ref).abs().max()) < 0.02 * float(ref.abs().max())
    want_cs = dI.float().sum(0)
    assert bool(((cs - want_cs).abs() <= 2e-3 * dI.float().abs().sum(0) + 1e-3).all())
