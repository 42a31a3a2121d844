"""Development check of the 256x256 phase-interleaved GEMM (run with SSAK_GEMM_P8=1): exact integer products on
ragged shapes, all four layouts, K tails, split-K, batches; repeated to screen for LDS-DMA races."""
import itertools
import sys
import torch
import ssak_amd.hip as h


def ref(A, B, a_km, b_km):
    A = A.float().T if a_km else A.float()
    B = B.float() if b_km else B.float().T
    return A @ B


def one(M, N, K, a_km, b_km, split_k=1, reps=3):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    Mp, Np, Kp = (M + 7) // 8 * 8, (N + 7) // 8 * 8, (K + 7) // 8 * 8
    A = torch.randint(-3, 4, (Kp, Mp) if a_km else (Mp, Kp), generator=g).to(torch.bfloat16)
    B = torch.randint(-3, 4, (Kp, Np) if b_km else (Np, Kp), generator=g).to(torch.bfloat16)
    if a_km:
        A[K:, :] = 0
        A[:, M:] = 0
    else:
        A[M:, :] = 0
        A[:, K:] = 0
    if b_km:
        B[K:, :] = 0
        B[:, N:] = 0
    else:
        B[N:, :] = 0
        B[:, K:] = 0
    Av = A[:K, :M] if a_km else A[:M, :K]
    Bv = B[:K, :N] if b_km else B[:N, :K]
    r = ref(Av.cuda(), Bv.cuda(), a_km, b_km)
    Ad, Bd = A.cuda(), B.cuda()
    bad = 0
    for _ in range(reps):
        Cc = torch.full((M, Np), -7.0, dtype=torch.float32).cuda()
        h.gemm(Ad, Bd, Cc, M, N, K, a_kmajor=a_km, b_kmajor=b_km, lda=A.shape[1], ldb=B.shape[1], ldc=Np, split_k=split_k,
               pads_are_zero=True)
        ok = torch.equal(Cc[:, :N], r) and bool((Cc[:, N:] == -7.0).all())
        bad += 0 if ok else 1
        if not ok:
            d = (Cc[:, :N] - r).abs()
            idx = torch.nonzero(d > 0)
            print("   mismatch", int((d > 0).sum()), "elements; first", idx[:4].tolist(), "max", float(d.max()))
    print(f"M={M} N={N} K={K} a_km={a_km} b_km={b_km} sk={split_k}: {'OK' if bad == 0 else 'FAIL'}", flush=True)
    return bad


shapes = [(256, 256, 64), (256, 256, 128), (512, 256, 192), (499, 768, 768), (700, 520, 200), (1000, 264, 72),
          (3000, 1000, 3072), (257, 257, 520)]
fails = 0
for (M, N, K), (a, b) in itertools.product(shapes, [(False, False), (False, True), (True, True), (True, False)]):
    fails += one(M, N, K, a, b)
fails += one(768, 768, 7984, True, True, split_k=4)
fails += one(768, 3072, 4000, True, True, split_k=3)
fails += one(2000, 768, 768, False, False, split_k=2)
print("FAILS", fails)
sys.exit(1 if fails else 0)
