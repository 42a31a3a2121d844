"""Run one GEMM shape repeatedly (for rocprofv3 --pmc passes)."""
import sys
import torch
import ssak_amd.hip as h
M, N, K = (int(a) for a in sys.argv[1:4])
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(20):
    h.gemm(A, B, C, M, N, K, lda=K, ldb=K, ldc=N)
torch.cuda.synchronize()
