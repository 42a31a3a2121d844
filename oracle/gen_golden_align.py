"""Writes tests/golden/align.npz: forced-alignment vectors computed with the torch ops the reference's
ssak/utils/align_transcriptions.py calls (torch.cumsum, torch.maximum, torch.argmax on float32 CPU tensors), on seeded
emissions.  Run in the build container: python -m oracle.gen_golden_align.  The numpy oracle (oracle/align_ref.py) must
reproduce these bit for bit (tests/test_oracle.py)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def trellis_torch(emission, tokens, blank_id=0, first_as_garbage=False):
    # align_transcriptions.py:27-53 (USE_MAX, USE_CHAR_REPEATED)
    F, L = emission.size(0), len(tokens)
    trellis = torch.empty((F + 1, L + 1))
    trellis[0, 0] = 0
    if first_as_garbage:
        trellis[1:, 0] = (1 - emission[:, tokens[0]].exp()).log()
    else:
        trellis[1:, 0] = torch.cumsum(emission[:, blank_id], 0)
    trellis[0, -L:] = -float("inf")
    trellis[-L:, 0] = float("inf")
    for t in range(F):
        trellis[t + 1, 1:] = torch.maximum(trellis[t, 1:] + emission[t, blank_id],
                                           torch.maximum(trellis[t, 1:] + emission[t, tokens], trellis[t, :-1] + emission[t, tokens]))
    return trellis


def backtrack_torch(trellis, emission, tokens, blank_id=0):
    # align_transcriptions.py:79-123; returns (token_index, time_index, score) rows, oldest first, or None on failure
    j = trellis.size(1) - 1
    t_start = torch.argmax(trellis[:, j]).item()
    path = []
    for t in range(t_start, 0, -1):
        stayed = trellis[t - 1, j] + emission[t - 1, blank_id]
        stayed = torch.maximum(stayed, trellis[t - 1, j] + emission[t - 1, tokens[j - 1]])
        changed = trellis[t - 1, j - 1] + emission[t - 1, tokens[j - 1]]
        if changed < stayed and t < emission.shape[0]:
            prob = torch.maximum(emission[t - 1, 0], emission[t, tokens[j - 1]]).exp().item()
        else:
            prob = emission[t - 1, tokens[j - 1] if changed > stayed else 0].exp().item()
        path.append((j - 1, t - 1, prob))
        if changed > stayed:
            j -= 1
            if j == 0:
                break
    else:
        return None
    return path[::-1]


def case(seed, F, V, L, blank, peaky, garbage=False):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(F, V, generator=g) * 2.0
    tokens = torch.randint(1 if blank == 0 else 0, V, (L,), generator=g).tolist()
    tokens = [t if t != blank else (t + 1) % V for t in tokens]
    if peaky and L > 0:  # emissions that follow a plausible alignment, as a trained model's would
        pos = sorted(torch.randperm(F, generator=g)[:L].tolist())
        logits[:, blank] += 3.0
        for k, p in enumerate(pos):
            logits[p, tokens[k]] += 8.0
    emission = torch.log_softmax(logits, dim=-1)
    tr = trellis_torch(emission, tokens, blank, garbage)
    path = backtrack_torch(tr, emission, tokens, blank)
    return emission.numpy(), np.array(tokens, np.int32), tr.numpy(), path


if __name__ == "__main__":
    out = {}
    specs = {"tiny": (1, 12, 6, 4, 0, False, False), "base": (2, 499, 32, 97, 0, True, False), "blank5": (3, 200, 40, 33, 5, True, False),
             "garbage": (4, 150, 32, 20, 0, True, True), "flat": (5, 300, 32, 64, 0, False, False), "tight": (6, 40, 8, 37, 0, True, False),
             "infeasible": (7, 10, 8, 14, 0, False, False)}
    for name, (seed, F, V, L, blank, peaky, garbage) in specs.items():
        em, tok, tr, path = case(seed, F, V, L, blank, peaky, garbage)
        out[f"{name}_emission"] = em
        out[f"{name}_tokens"] = tok
        out[f"{name}_blank"] = np.int32(blank)
        out[f"{name}_garbage"] = np.int32(garbage)
        out[f"{name}_trellis"] = tr
        out[f"{name}_ok"] = np.int32(path is not None)
        if path is not None:
            out[f"{name}_path_token"] = np.array([p[0] for p in path], np.int32)
            out[f"{name}_path_time"] = np.array([p[1] for p in path], np.int32)
            out[f"{name}_path_score"] = np.array([p[2] for p in path], np.float64)
        print(name, "F", F, "L", L, "path", None if path is None else len(path))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "align.npz"), **out)
