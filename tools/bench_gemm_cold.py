"""Reference point, not part of the product: the train step's forward / dX products with COLD operands -- eight rotating
operand sets (the activations of a step are read once, from HBM; a warm micro-benchmark keeps them in the Infinity Cache) --
next to the warm numbers and to the vendor GEMM under the same rotation.  usage: PYTHONPATH=. python tools/bench_gemm_cold.py [sets=8]"""
import sys
import torch
import ssak_amd.hip as h

SETS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M = 32 * 499


def timeit(fn, n=40):
    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, m, n, k in [("qkv", M, 2304, 768), ("out_proj", M, 768, 768), ("ffn1", M, 3072, 768), ("ffn2", M, 768, 3072), ("qkv dX", M, 768, 2304)]:
    As = [torch.randn(m, k, device="cuda").to(torch.bfloat16) for _ in range(SETS)]
    Ws = [torch.randn(n, k, device="cuda").to(torch.bfloat16) for _ in range(SETS)]
    C = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    fl = 2.0 * m * n * k
    row = f"{name:9s} {m} x {n:4d} x {k:4d}:"
    for label, sets in (("warm", 1), ("cold", SETS)):
        t_v = timeit(lambda i: torch.matmul(As[i % sets], Ws[i % sets].t(), out=C))
        t_o = timeit(lambda i: h.gemm(As[i % sets], Ws[i % sets], C, m, n, k, lda=k, ldb=k, ldc=n))
        t_a = timeit(lambda i: h.gemm(As[i % sets], Ws[0], C, m, n, k, lda=k, ldb=k, ldc=n))  # only the activations rotate
        row += f"  {label}: vendor {t_v:6.1f} us {fl / t_v / 1e6:6.0f} TF | ssak {t_o:6.1f} us {fl / t_o / 1e6:6.0f} TF (A only: {t_a:6.1f} us)"
    print(row, flush=True)
