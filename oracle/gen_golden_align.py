"""Writes tests/golden/align.npz (+ align_segments.json): forced-alignment vectors computed BY THE REFERENCE'S OWN FUNCTIONS --
`get_trellis`, `backtrack`, `merge_repeats`, `merge_words` of /root/reference/ssak/utils/align_transcriptions.py:27-70,79-123,141-175,
imported here -- on seeded emissions.  The module's other imports (ssak.utils.text -> num2words, ssak.utils.viewer -> pyaudio,
ssak.infer.general -> speechbrain) have nothing to do with these four functions and are absent from the image, so three empty
stand-in modules are put into sys.modules for the duration of the import; nothing of the alignment arithmetic is restated on
this path.  Run in the build container (the reference does not travel): python -m oracle.gen_golden_align.
The torch restatement below (trellis_torch / backtrack_torch) is kept as a second opinion: the generator asserts that it is
bit-identical to the reference on every case.  The numpy oracle (oracle/align_ref.py) must reproduce the file bit for bit
(tests/test_oracle.py), and the HIP kernels are held to it on the GPU (tests/test_gpu_align.py)."""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("SSAK_REFERENCE", "/root/reference")


def import_reference_align():
    """ssak.utils.align_transcriptions of the reference, with stand-ins for the three unrelated modules it imports at the top."""
    names = {"ssak.utils.text": ["transliterate"], "ssak.utils.viewer": ["PlayWav"],
             "ssak.infer.general": ["load_model", "compute_logits", "compute_log_probas", "decode_log_probas", "get_model_vocab",
                                    "get_model_sample_rate"]}
    saved = {n: sys.modules.get(n) for n in list(names) + ["ssak.infer"]}
    sys.path.insert(0, REFERENCE)
    try:
        import matplotlib
        matplotlib.use("Agg")
        if "ssak.infer" not in sys.modules:
            pkg = types.ModuleType("ssak.infer")
            pkg.__path__ = []
            sys.modules["ssak.infer"] = pkg
        for n, attrs in names.items():
            m = types.ModuleType(n)
            for a in attrs:
                setattr(m, a, None)
            sys.modules[n] = m
        import importlib
        mod = importlib.import_module("ssak.utils.align_transcriptions")
    finally:
        sys.path.remove(REFERENCE)
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
    assert os.path.realpath(mod.__file__).startswith(os.path.realpath(REFERENCE)), mod.__file__
    assert mod.USE_MAX and mod.USE_CHAR_REPEATED
    return mod


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
    # the reference's functions ...
    tr = REF.get_trellis(emission, tokens, blank, garbage)
    try:
        path = [(p.token_index, p.time_index, p.score) for p in REF.backtrack(tr, emission, tokens, blank)]
    except RuntimeError as e:
        assert "Failed to align" in str(e)
        path = None
    # ... and the restatement, which must be indistinguishable
    tr2 = trellis_torch(emission, tokens, blank, garbage)
    assert tr.numpy().tobytes() == tr2.numpy().tobytes(), "restated trellis differs from the reference"
    assert path == backtrack_torch(tr2, emission, tokens, blank), "restated path differs from the reference"
    return emission.numpy(), np.array(tokens, np.int32), tr.numpy(), path


if __name__ == "__main__":
    REF = import_reference_align()
    out, segs_out = {}, {}
    specs = {"tiny": (1, 12, 6, 4, 0, False, False), "base": (2, 499, 32, 97, 0, True, False), "blank5": (3, 200, 40, 33, 5, True, False),
             "garbage": (4, 150, 32, 20, 0, True, True), "flat": (5, 300, 32, 64, 0, False, False), "tight": (6, 40, 8, 37, 0, True, False),
             "infeasible": (7, 10, 8, 14, 0, False, False)}
    alphabet = "abcdefghijklmnopqrstuvwxyz' "
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
            # merge_repeats / merge_words of the reference on a transcript spelled from the token ids (a word break every 5th token)
            transcript = "".join(" " if k % 5 == 4 else alphabet[int(t) % 27] for k, t in enumerate(tok))
            pts = [REF.Point(*p) for p in path]
            segs = REF.merge_repeats(transcript, pts)
            words = REF.merge_words(segs)
            segs_out[name] = {"transcript": transcript,
                              "segments": [[s_.label, int(s_.start), int(s_.end), float(s_.score)] for s_ in segs],
                              "words": [[w.label, int(w.start), int(w.end), float(w.score)] for w in words]}
        print(name, "F", F, "L", L, "path", None if path is None else len(path))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "align.npz"), **out)
    with open(os.path.join(ROOT, "tests", "golden", "align_segments.json"), "w") as f:
        json.dump({"source": "merge_repeats / merge_words of the reference's ssak/utils/align_transcriptions.py:141-175 on align.npz's paths",
                   "cases": segs_out}, f)
