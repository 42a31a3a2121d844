import torch, ssak_amd.hip as h
def timeit(fn,n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for name,m,n,k in [("8192^3",8192,8192,8192),("4096^3",4096,4096,4096),("6144x8192x8192",6144,8192,8192),("conv1-like",65536,512,1536)]:
    for mode in ("random","zeros"):
        A=(torch.randn(m,k,device="cuda") if mode=="random" else torch.zeros(m,k,device="cuda")).to(torch.bfloat16)
        W=(torch.randn(n,k,device="cuda") if mode=="random" else torch.zeros(n,k,device="cuda")).to(torch.bfloat16)
        C=torch.empty(m,n,dtype=torch.bfloat16,device="cuda")
        row=f"{name:16s} {mode:6s}:"
        for tile in (256,192,128):
            t=timeit(lambda: h.gemm(A,W,C,m,n,k,lda=k,ldb=k,ldc=n,plan_tile=tile))
            row+=f"  t{tile} {t:8.1f} us {2.0*m*n*k/t/1e6:7.1f} TF"
        tv=timeit(lambda: torch.matmul(A,W.t(),out=C))
        row+=f"  | vendor {tv:8.1f} us {2.0*m*n*k/tv/1e6:7.1f} TF"
        print(row,flush=True)
