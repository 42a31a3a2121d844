"""Reference point: the train step's K-contiguous products per tile height (ssak_gemm_desc.plan_tile), warm and with cold operands
(eight rotating sets).  usage: PYTHONPATH=. python tools/bench_gemm_tiles.py"""
import torch
import ssak_amd.hip as h

M = 32 * 499
SETS = 8


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


for name, m, n, k in [("qkv", M, 2304, 768), ("out_proj", M, 768, 768), ("ffn1", M, 3072, 768), ("ffn2", M, 768, 3072), ("conv1", 32 * 15999, 512, 1536)]:
    As = [torch.randn(m, k, device="cuda").to(torch.bfloat16) for _ in range(SETS)]
    Ws = [torch.randn(n, k, device="cuda").to(torch.bfloat16) for _ in range(SETS)]
    C = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    fl = 2.0 * m * n * k
    row = f"{name:9s} {m} x {n:4d} x {k:4d}:"
    for tile in (0, 256, 192, 128):
        t_w = timeit(lambda i: h.gemm(As[0], Ws[0], C, m, n, k, lda=k, ldb=k, ldc=n, plan_tile=tile))
        t_c = timeit(lambda i: h.gemm(As[i % SETS], Ws[i % SETS], C, m, n, k, lda=k, ldb=k, ldc=n, plan_tile=tile))
        row += f"  tile {tile:3d}: warm {t_w:6.1f} us {fl / t_w / 1e6:5.0f} TF, cold {t_c:6.1f} us {fl / t_c / 1e6:5.0f} TF |"
    print(row, flush=True)
    del As, Ws, C
