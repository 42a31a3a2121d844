"""Cost of the GEMM epilogue variants on the FFN1 shape (development aid)."""
import sys
import torch
import ssak_amd.hip as h

M, N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 15968, 3072, 768
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
pre = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
bias = torch.randn(N, device="cuda")


def run(name, **kw):
    for _ in range(3):
        h.gemm(A, B, C, M, N, K, lda=K, ldb=K, ldc=N, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        h.gemm(A, B, C, M, N, K, lda=K, ldb=K, ldc=N, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:34s} {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF/s", flush=True)


run("plain")
run("bias", bias=bias)
run("bias+gelu", bias=bias, epilogue=h.EPI_GELU)
run("bias+gelu+preact", bias=bias, epilogue=h.EPI_GELU, aux_out=pre)
run("bias+gelu+preact+dropout", bias=bias, epilogue=h.EPI_GELU, aux_out=pre, drop_p=0.1, drop_seed=1, drop_stream=2)
run("gelugrad(aux)+dropout", epilogue=h.EPI_MUL_GELU_GRAD, aux_in=pre, drop_p=0.1, drop_seed=1, drop_stream=2)
run("gelugrad(aux)", epilogue=h.EPI_MUL_GELU_GRAD, aux_in=pre)
codes = torch.empty(M, N, dtype=torch.uint8, device="cuda")
run("FFN-up: bias+gelu+gelu' codes+dropout", bias=bias, epilogue=h.EPI_GELU_SAVE_GRAD, aux_out=codes, drop_p=0.1, drop_seed=1, drop_stream=2)
