"""Fused attention kernels on the train-step shape (B x 12 heads x 499 frames, head_dim 64): time per launch with and
without dropout, the backward alone and with the q|k|v bias sums taken in its kernels.  Buffers are allocated once and the library is
called directly, so the figures are kernel time + launch.  usage: PYTHONPATH=. python tools/bench_attn.py [B=32] [F=499] [nh=12]"""
import sys
import torch
import ssak_amd.hip as h

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
F = int(sys.argv[2]) if len(sys.argv) > 2 else 499
nh = int(sys.argv[3]) if len(sys.argv) > 3 else 12
H = nh * 64
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * F, 3 * H, generator=g) * 0.8).to(torch.bfloat16).cuda()
dctx = (torch.randn(B * F, H, generator=g) * 0.5).to(torch.bfloat16).cuda()
ctx = torch.empty((B * F, H), dtype=torch.bfloat16, device="cuda")
lse = torch.empty((B, nh, F), dtype=torch.float32, device="cuda")
delta = torch.empty((B, nh, F), dtype=torch.float32, device="cuda")
dqkv = torch.empty_like(qkv)
lib, ptr, st = h.lib, h.ptr, h.stream


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fl = 4.0 * B * nh * F * F * 64
for p in (0.0, 0.1):
    fwd = lambda: h.check(lib.ssak_attention_fwd(ptr(qkv), ptr(ctx), ptr(lse), None, B, F, nh, H, p, 1, 3, st()))
    bwd = lambda mode=1: h.check(lib.ssak_attention_bwd(ptr(qkv), ptr(ctx), ptr(lse), None, ptr(dctx), ptr(delta), ptr(dqkv), B, F, nh, H, p, 1, 3, mode, st()))
    tf = timeit(fwd)
    out = [f"p={p}: fwd {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF/s)"]
    tb = timeit(bwd)
    out.append(f"bwd (dQ; dK + dV) {tb:7.1f} us ({2 * fl / tb / 1e6:6.1f} TF/s)")
    bias = torch.zeros(3 * H, dtype=torch.float32, device="cuda")
    ws = torch.empty(lib.ssak_attention_bwd_bias_workspace_bytes(B, F, H), dtype=torch.uint8, device="cuda")
    tbb = timeit(lambda: h.check(lib.ssak_attention_bwd_bias(ptr(qkv), ptr(ctx), ptr(lse), None, ptr(dctx), ptr(delta), ptr(dqkv), ptr(bias),
                                                             B, F, nh, H, p, 1, 3, ptr(ws), ws.numel(), st())))
    out.append(f"bwd + q|k|v bias sums {tbb:7.1f} us")
    print("   ".join(out))
