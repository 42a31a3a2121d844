"""The co-resident kernel (gemm_c4.hip, plan_tile 129) next to the library's default plan on the train step's K-contiguous
products, per epilogue, warm and with cold operands (eight rotating sets).  usage: PYTHONPATH=. python tools/bench_gemm_c4.py"""
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


for name, m, n, k, epi in [("qkv", M, 2304, 768, "plain"), ("out_proj", M, 768, 768, "plain"), ("ffn_up", M, 3072, 768, "gelu_save"),
                           ("ffn_dx", M, 3072, 768, "mul_aux"), ("ffn_down", M, 768, 3072, "plain"), ("qkv_dx", M, 768, 2304, "plain"),
                           ("plain3072", M, 3072, 768, "plain")]:
    As = [torch.randn(m, k, device="cuda").to(torch.bfloat16) for _ in range(SETS)]
    Ws = [(torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(SETS)]
    bias = torch.randn(n, device="cuda")
    C = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    f8 = torch.randint(0, 256, (m, n), dtype=torch.int32, device="cuda").to(torch.uint8)
    cs = torch.zeros(n, dtype=torch.float32, device="cuda")
    fl = 2.0 * m * n * k
    row = f"{name:9s} {m} x {n:4d} x {k:4d} {epi:9s}:"
    for tile in (0, 129):
        def run(i, tile=tile):
            kw = dict(lda=k, ldb=k, ldc=n, plan_tile=tile)
            if epi == "plain":
                h.gemm(As[i % SETS], Ws[i % SETS], C, m, n, k, bias=bias, **kw)
            elif epi == "gelu_save":
                h.gemm(As[i % SETS], Ws[i % SETS], C, m, n, k, bias=bias, epilogue=h.EPI_GELU_SAVE_GRAD, aux_out=f8, drop_p=0.1, drop_stream=5, drop_seed=77, **kw)
            else:
                h.gemm(As[i % SETS], Ws[i % SETS], C, m, n, k, epilogue=h.EPI_MUL_AUX, aux_in=f8, colsum_out=cs, drop_p=0.1, **kw)
        t_w = timeit(lambda i: run(0))
        t_c = timeit(run)
        row += f"  tile {tile:3d}: warm {t_w:6.1f} us {fl / t_w / 1e6:5.0f} TF, cold {t_c:6.1f} us {fl / t_c / 1e6:5.0f} TF |"
    print(row, flush=True)
    del As, Ws, C, f8
