"""Fused attention kernels on the train-step shape (B x 12 heads x 499 frames, head_dim 64): time per launch with and
without dropout.  usage: PYTHONPATH=. python tools/bench_attn.py [B=32] [F=499] [nh=12]"""
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


def timeit(fn, n=20):
    for _ in range(3):
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
    kw = dict(drop_p=p, seed=1, stream_id=3) if p else {}
    ctx, lse = h.attention_fwd(qkv, B, F, nh, **kw)
    tf = timeit(lambda: h.attention_fwd(qkv, B, F, nh, **kw))
    tb = timeit(lambda: h.attention_bwd(qkv, ctx, lse, dctx, B, F, nh, **kw))
    print(f"p={p}: fwd {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF/s)   bwd (delta+dq+dkv) {tb:7.1f} us ({3.5 * fl / tb / 1e6:6.1f} TF/s of executed flops)")
