import sys, torch
sys.path.insert(0, '.')
import ssak_amd.hip as h
def bench(M, N, K, sk, iters=20):
    A = torch.randn(K, M, device="cuda").to(torch.bfloat16); B = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.float32, device="cuda")
    kw = dict(a_kmajor=True, b_kmajor=True, lda=M, ldb=N, ldc=N, split_k=sk)
    for _ in range(3): h.gemm(A, B, C, M, N, K, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): h.gemm(A, B, C, M, N, K, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N) in ((3072, 768), (2304, 768), (768, 768)):
    print(M, N, " ".join(f"sk{sk}:{bench(M, N, 15968, sk):.0f}" for sk in (0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 20, 24, 28)), flush=True)
