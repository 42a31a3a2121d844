"""Forced alignment: device kernel vs the CPU restatement of the reference's per-frame torch loop.
usage: PYTHONPATH=. python tools/bench_align.py"""
import time

import numpy as np
import torch

from ssak_amd import align


def torch_loop(emission, tokens, blank_id=0):
    """The reference's get_trellis (align_transcriptions.py:27-53) as it runs: a Python loop of torch CPU ops."""
    F, L = emission.size(0), len(tokens)
    trellis = torch.empty((F + 1, L + 1))
    trellis[0, 0] = 0
    trellis[1:, 0] = torch.cumsum(emission[:, blank_id], 0)
    trellis[0, -L:] = -float("inf")
    trellis[-L:, 0] = float("inf")
    for t in range(F):
        trellis[t + 1, 1:] = torch.maximum(trellis[t, 1:] + emission[t, blank_id],
                                           torch.maximum(trellis[t, 1:] + emission[t, tokens], trellis[t, :-1] + emission[t, tokens]))
    return trellis


for F, L in [(499, 100), (3000, 600), (12000, 2400)]:
    g = torch.Generator().manual_seed(F)
    em = torch.log_softmax(torch.randn(F, 32, generator=g), -1)
    tok = torch.randint(1, 32, (L,), generator=g).tolist()
    emd = em.cuda()
    align.forced_align(emd, tok, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        tr, path = align.forced_align(emd, tok, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    ref = torch_loop(em, tok)
    dc = time.perf_counter() - t0
    same = bool(torch.equal(tr.cpu(), ref))
    print(f"F={F:6d} L={L:5d}: device trellis+backtrack {dt * 1e3:8.2f} ms ({dt / F * 1e6:5.2f} us/frame)   "
          f"torch CPU trellis loop {dc * 1e3:9.1f} ms   trellis identical: {same}")
