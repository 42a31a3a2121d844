"""A few launches of the attention kernels on the train-step shape, for rocprofv3 counter passes."""
import sys
import torch
import ssak_amd.hip as h
B, F, nh = 32, 499, 12
H = nh * 64
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * F, 3 * H, generator=g) * 0.8).to(torch.bfloat16).cuda()
dctx = (torch.randn(B * F, H, generator=g) * 0.5).to(torch.bfloat16).cuda()
ctx = torch.empty((B * F, H), dtype=torch.bfloat16, device="cuda")
lse = torch.empty((B, nh, F), dtype=torch.float32, device="cuda")
delta = torch.empty((B, nh, F), dtype=torch.float32, device="cuda")
dqkv = torch.empty_like(qkv)
for _ in range(3):
    h.check(h.lib.ssak_attention_fwd(h.ptr(qkv), h.ptr(ctx), h.ptr(lse), None, B, F, nh, H, p, 1, 3, h.stream()))
    for split in (0, 1):
        h.attention_bwd_mode(bool(split))
        h.check(h.lib.ssak_attention_bwd(h.ptr(qkv), h.ptr(ctx), h.ptr(lse), None, h.ptr(dctx), h.ptr(delta), h.ptr(dqkv), B, F, nh, H, p, 1, 3, h.stream()))
torch.cuda.synchronize()
