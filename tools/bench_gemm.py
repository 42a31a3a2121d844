"""Micro-benchmark of ssak_gemm_bf16 on the shapes of the Wav2Vec2-base train step (development aid)."""
import sys
import torch
import ssak_amd.hip as h

def bench(name, M, N, K, a_km=False, b_km=False, nb=1, split_k=1, out=torch.bfloat16, iters=20):
    A = torch.randn((K, M) if a_km else (M, K), device="cuda").to(torch.bfloat16).repeat(nb, 1)
    B = torch.randn((K, N) if b_km else (N, K), device="cuda").to(torch.bfloat16)
    C = torch.empty(nb * M, N, dtype=out, device="cuda")
    kw = dict(a_kmajor=a_km, b_kmajor=b_km, lda=A.shape[1], ldb=B.shape[1], ldc=N, nb1=nb,
              sa=(A.shape[0] // nb * A.shape[1], 0), sc=(M * N, 0), split_k=split_k)
    for _ in range(3):
        h.gemm(A, B, C, M, N, K, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        h.gemm(A, B, C, M, N, K, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:28s} M={M:6d} N={N:5d} K={K:6d} nb={nb:3d} sk={split_k} {ms*1e3:9.1f} us  {2*M*N*K*nb/ms/1e9:8.1f} TF/s", flush=True)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = B * 499
bench("qkv", M, 2304, 768)
bench("out_proj", M, 768, 768)
bench("ffn1", M, 3072, 768)
bench("ffn2", M, 768, 3072)
bench("dx ffn2 (NN)", M, 3072, 768, b_km=True)
bench("dx ffn1 (NN)", M, 768, 3072, b_km=True)
for sk in (0, 1, 2, 4, 8):
    bench("dW ffn (TN)", 3072, 768, M, a_km=True, b_km=True, split_k=sk, out=torch.float32)
bench("dW ffn2 (TN)", 768, 3072, M, a_km=True, b_km=True, split_k=0, out=torch.float32)
bench("dW qkv (TN)", 2304, 768, M, a_km=True, b_km=True, split_k=0, out=torch.float32)
for sk in (0, 4, 8, 16):
    bench("dW proj (TN)", 768, 768, M, a_km=True, b_km=True, split_k=sk, out=torch.float32)
bench("conv1 per-utt", 15999, 512, 1536, nb=B)
bench("conv3 per-utt", 3999, 512, 1536, nb=B)
bench("lm_head", M, 32, 768, out=torch.float32)
bench("4096^3", 4096, 4096, 4096)
