"""Small-M products (the reference's default --batch_size 8 gives M = 3 992 rows: 48 output tiles of 256 x 256 for an N = 768 product on
256 CUs): the library's default plan next to explicit tile heights and deterministic split-K factors, per shape of the train step.
usage: PYTHONPATH=. python tools/bench_small_m.py [B ...]"""
import sys

import torch

import ssak_amd.hip as h


def timeit(fn, n=30):
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


for B in [int(a) for a in sys.argv[1:]] or [8, 16]:
    M = B * 499
    print(f"---- B = {B}: M = {M}")
    for name, n, k, bkm in [("ffn2 fwd", 768, 3072, False), ("ffn1 dX", 768, 3072, False), ("qkv dX", 768, 2304, False), ("qkv fwd", 2304, 768, False),
                            ("out fwd", 768, 768, False), ("ffn1 fwd (plain)", 3072, 768, False)]:
        A = torch.randn(M, k, device="cuda").to(torch.bfloat16)
        W = torch.randn(n, k, device="cuda").to(torch.bfloat16)
        C = torch.empty(M, n, dtype=torch.bfloat16, device="cuda")
        bias = torch.randn(n, device="cuda")
        fl = 2.0 * M * n * k
        row = f"{name:17s} N {n:4d} K {k:4d}:"
        base = timeit(lambda: h.gemm(A, W, C, M, n, k, lda=k, ldb=k, ldc=n, bias=bias))
        row += f" default {base:6.1f} us ({fl / base / 1e6:5.0f} TF) | auto-split {timeit(lambda: h.gemm(A, W, C, M, n, k, lda=k, ldb=k, ldc=n, bias=bias, split_k=0)):6.1f} |"
        for tile in (256, 192, 128):
            for s in (1, 2, 3, 4, 6):
                if s > 1 and k // 64 // s < 4:
                    continue
                t = timeit(lambda: h.gemm(A, W, C, M, n, k, lda=k, ldb=k, ldc=n, bias=bias, split_k=s, plan_tile=tile))
                row += f" t{tile}s{s} {t:5.1f}"
            row += " |"
        print(row, flush=True)
