"""conv0 + GroupNorm + GELU (the first feature-encoder layer) on the headline shape: per-launch time of the three kernels together
(moments, channel statistics, apply).  usage: PYTHONPATH=. python tools/bench_conv0.py [B=32]"""
import sys

import torch

import ssak_amd.hip as h

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator().manual_seed(0)
x = (torch.randn(B, 160000, generator=g) * 0.1).cuda()
w = (torch.randn(512, 10, generator=g) * 0.3).cuda()
gamma = (1.0 + 0.2 * torch.randn(512, generator=g)).cuda()
beta = (0.2 * torch.randn(512, generator=g)).cuda()
for raw in (False, True):
    for _ in range(3):
        h.conv0_gn_gelu(x, w, gamma, beta, raw=raw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = h.conv0_gn_gelu(x, w, gamma, beta, raw=raw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    by = B * (160000 * 4 + 31999 * 512 * 2)
    print(f"conv0 {'raw (folded normalisation)' if raw else 'normalised input'}: {us:7.1f} us per call, {by / us / 1e3:7.1f} GB/s of algorithmic traffic (incl. output allocation)")
