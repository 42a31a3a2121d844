"""Is the K loop bound by the schedule or by power?  The vendor GEMM and ssak_gemm_bf16 on the big K-contiguous shapes with RANDOM operands (the bench's
data: the matrix pipe toggles every bit, the chip runs power-limited at 1.7-1.9 GHz) and with ALL-ZERO operands (no toggling: the same
instruction stream at the top clock).  usage: PYTHONPATH=. python tools/bench_gemm_data.py"""
import torch

import ssak_amd.hip as h


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, m, n, k in [("ffn1", 15968, 3072, 768), ("ffn2", 15968, 768, 3072), ("4096^3", 4096, 4096, 4096), ("8192^3", 8192, 8192, 8192)]:
    for mode in ("random", "zeros", "ones"):
        if mode == "random":
            A = torch.randn(m, k, device="cuda").to(torch.bfloat16)
            W = torch.randn(n, k, device="cuda").to(torch.bfloat16)
        elif mode == "zeros":
            A = torch.zeros(m, k, device="cuda", dtype=torch.bfloat16)
            W = torch.zeros(n, k, device="cuda", dtype=torch.bfloat16)
        else:
            A = torch.ones(m, k, device="cuda", dtype=torch.bfloat16)
            W = torch.ones(n, k, device="cuda", dtype=torch.bfloat16)
        C = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        t_v = timeit(lambda: torch.matmul(A, W.t(), out=C))
        t_o = timeit(lambda: h.gemm(A, W, C, m, n, k, lda=k, ldb=k, ldc=n))
        fl = 2.0 * m * n * k
        print(f"{name:8s} {mode:7s}: vendor {t_v:8.1f} us {fl / t_v / 1e6:7.1f} TF/s | ssak {t_o:8.1f} us {fl / t_o / 1e6:7.1f} TF/s | ssak / vendor time {t_o / t_v:.3f}", flush=True)
