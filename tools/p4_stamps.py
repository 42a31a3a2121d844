"""Where a tile of the four-wave GEMM spends its time: a library built from tools/probes/p4_stamps.patch.txt (git apply it, then make OBJ=build/obj_st LIB=tools/ab_stamps.so
EXTRA=-DP4_STAMPS) records the shader clock at the tile's start, when its first K tile's fragments are read, after the K loop and after the
epilogue, per workgroup and tile.  usage: SSAK_HIP_LIB=$PWD/tools/ab_stamps.so PYTHONPATH=. python tools/p4_stamps.py"""
import ctypes

import numpy as np
import torch

import ssak_amd.hip as h

M = 32 * 499
lib = h.lib
lib.ssak_debug_p4_stamps.restype = ctypes.c_int
buf = np.zeros((256, 8, 4), dtype=np.uint64)
for name, n, k in [("qkv", 2304, 768), ("out-proj", 768, 768), ("ffn2", 768, 3072), ("qkv dX", 768, 2304), ("ffn1 plain (256-row tiles)", 3072, 768)]:
    A = torch.randn(M, k, device="cuda").to(torch.bfloat16)
    W = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    C = torch.empty(M, n, dtype=torch.bfloat16, device="cuda")
    bias = torch.randn(n, device="cuda")
    for _ in range(3):
        h.gemm(A, W, C, M, n, k, lda=k, ldb=k, ldc=n, bias=bias)
    torch.cuda.synchronize()
    lib.ssak_debug_p4_stamps(buf.ctypes.data_as(ctypes.c_void_p))  # (clears)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    h.gemm(A, W, C, M, n, k, lda=k, ldb=k, ldc=n, bias=bias)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    lib.ssak_debug_p4_stamps(buf.ctypes.data_as(ctypes.c_void_p))
    st = buf.astype(np.float64)
    used = st[:, :, 0] > 0
    t0 = st[:, :, 0][used].min()
    rounds = int(used.sum(1).max())
    # the cycle counter runs at a fixed 100 MHz on this chip: report in its ticks and as a share of the launch
    span = st[:, :, 3][used].max() - t0
    print(f"{name}: launch {us:.1f} us, {int(used.sum())} tiles, up to {rounds} per workgroup; kernel span {span:.0f} ticks")
    for r in range(rounds):
        u = used[:, r]
        if not u.any():
            continue
        s = st[u, r, :]
        print(f"   tile {r}: start {np.median(s[:, 0] - t0):8.0f}  wait+first reads {np.median(s[:, 1] - s[:, 0]):7.0f}  K loop {np.median(s[:, 2] - s[:, 1]):7.0f}"
              f"  epilogue {np.median(s[:, 3] - s[:, 2]):7.0f}  (medians over {int(u.sum())} workgroups; ticks)")
