"""Phase time stamps of the wave-specialised attention backward (library built with -DATT_WS_STAMPS, SSAK_HIP_LIB=...)."""
import numpy as np
import torch
import ssak_amd.hip as h
B, F, nh = 32, 499, 12
H = nh * 64
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * F, 3 * H, generator=g) * 0.8).to(torch.bfloat16).cuda()
dctx = (torch.randn(B * F, H, generator=g) * 0.5).to(torch.bfloat16).cuda()
ctx = torch.empty((B * F, H), dtype=torch.bfloat16, device="cuda")
lse = torch.empty((B, nh, F), dtype=torch.float32, device="cuda")
delta = torch.zeros((B, nh, F), dtype=torch.float32, device="cuda")
dqkv = torch.empty_like(qkv)
p = 0.1
h.check(h.lib.ssak_attention_fwd(h.ptr(qkv), h.ptr(ctx), h.ptr(lse), None, B, F, nh, H, p, 1, 3, h.stream()))
h.attention_bwd_mode(0)
for _ in range(3):
    h.check(h.lib.ssak_attention_bwd(h.ptr(qkv), h.ptr(ctx), h.ptr(lse), None, h.ptr(dctx), h.ptr(delta), h.ptr(dqkv), B, F, nh, H, p, 1, 3, h.stream()))
torch.cuda.synchronize()
st = delta.view(-1)[: 2 * 24 * 8 * 2].cpu().numpy().view(np.uint64).reshape(2, 24, 8).astype(np.int64)
t0 = st[0, 0, 0]
for role, name in ((0, "producer: top | S,dP+math pair 0 | pair 1 | handed over"), (1, "consumer: top | dV,dK | dQ+store | statistics")):
    print(name)
    for q in range(2, 12):
        r = st[role, q]
        print(f"  step {q:2d}: start {r[0] - t0:7d}  phases {r[1] - r[0]:6d} {r[2] - r[1]:6d} {r[3] - r[2]:6d}   next top in {st[role, q + 1, 0] - r[3]:6d}   step {st[role, q + 1, 0] - r[0]:6d}"
              + (f"   [dV,dK: to first read {r[4] - r[0]:5d}, first wait {r[5] - r[4]:5d}, 16 MFMA {r[6] - r[5]:5d}, second half {r[1] - r[6]:5d}]" if role == 1 else ""))
