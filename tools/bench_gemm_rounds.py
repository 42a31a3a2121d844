import torch, ssak_amd.hip as h
M=32*499
def timeit(fn,n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for name,m,n,k in [("qkv",M,2304,768),("ffn1 plain",M,3072,768),("conv1-like",131072,512,1536),("out-proj (1 round)",M,768,768),("ffn2 (1 round)",M,768,3072)]:
    A=torch.randn(m,k,device="cuda").to(torch.bfloat16); W=torch.randn(n,k,device="cuda").to(torch.bfloat16)
    C=torch.empty(m,n,dtype=torch.bfloat16,device="cuda"); bias=torch.randn(n,device="cuda")
    t=timeit(lambda: h.gemm(A,W,C,m,n,k,lda=k,ldb=k,ldc=n,bias=bias))
    print(f"{name:20s} {t:8.1f} us {2.0*m*n*k/t/1e6:7.1f} TF",flush=True)
