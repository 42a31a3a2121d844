#!/usr/bin/env python3
"""Per-kernel comparison of two bench.py JSON lines (same box): us per step of every profiler slot, side by side.
Usage: python tools/compare_bench.py A.json B.json"""
import json
import sys


def load(p):
    txt = [l for l in open(p).read().splitlines() if l.startswith("{")]
    return json.loads(txt[-1])


a, b = load(sys.argv[1]), load(sys.argv[2])
ka = {k["kernel"]: k for k in a["roofline"]["kernels"]}
kb = {k["kernel"]: k for k in b["roofline"]["kernels"]}
print(f"{'slot':86s} {'A us':>8s} {'B us':>8s} {'B-A':>7s}   A frac  B frac")
tot = 0.0
for name in sorted(set(ka) | set(kb), key=lambda n: -(ka.get(n) or kb.get(n))["us_per_step"]):
    x, y = ka.get(name), kb.get(name)
    ua, ub = (x or {}).get("us_per_step", 0.0), (y or {}).get("us_per_step", 0.0)
    if "workload" in (x or y):
        continue
    tot += ub - ua
    print(f"{name[:86]:86s} {ua:8.1f} {ub:8.1f} {ub - ua:+7.1f}   {(x or {}).get('frac', 0):6.3f}  {(y or {}).get('frac', 0):6.3f}")
print(f"sum of differences {tot:+.1f} us;  A {a['value']:.1f} utt/s {a['ms_per_step']:.3f} ms   B {b['value']:.1f} utt/s {b['ms_per_step']:.3f} ms")
