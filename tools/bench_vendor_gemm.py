"""Reference point, not part of the product: the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the train step's shapes,
next to ssak_gemm_bf16.  usage: PYTHONPATH=. python tools/bench_vendor_gemm.py [B=32]"""
import sys
import torch
import ssak_amd.hip as h

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = B * 499


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


for name, m, n, k in [("qkv", M, 2304, 768), ("out_proj", M, 768, 768), ("ffn1", M, 3072, 768), ("ffn2", M, 768, 3072), ("4096^3", 4096, 4096, 4096),
                      ("8192^3", 8192, 8192, 8192)]:
    A = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    W = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    C = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    t_v = timeit(lambda: torch.matmul(A, W.t(), out=C))
    t_o = timeit(lambda: h.gemm(A, W, C, m, n, k, lda=k, ldb=k, ldc=n))
    fl = 2.0 * m * n * k
    print(f"{name:10s} {m:6d} x {n:5d} x {k:5d}: vendor {t_v:8.1f} us {fl / t_v / 1e6:7.1f} TF/s | ssak {t_o:8.1f} us {fl / t_o / 1e6:7.1f} TF/s", flush=True)
# weight-gradient form (TN, fp32 out)
for name, m, n in [("dW ffn", 3072, 768), ("dW qkv", 2304, 768), ("dW proj", 768, 768)]:
    dY = torch.randn(M, m, device="cuda").to(torch.bfloat16)
    X = torch.randn(M, n, device="cuda").to(torch.bfloat16)
    C = torch.empty(m, n, dtype=torch.float32, device="cuda")
    Cb = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    t_v = timeit(lambda: torch.matmul(dY.t(), X, out=Cb))
    t_o = timeit(lambda: h.gemm(dY, X, C, m, n, M, a_kmajor=True, b_kmajor=True, lda=m, ldb=n, ldc=n, split_k=0))
    fl = 2.0 * m * n * M
    print(f"{name:10s} {m:6d} x {n:5d} x {M:5d}: vendor {t_v:8.1f} us {fl / t_v / 1e6:7.1f} TF/s (bf16 out) | ssak {t_o:8.1f} us {fl / t_o / 1e6:7.1f} TF/s (fp32 out)", flush=True)
