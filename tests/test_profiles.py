"""CPU: the committed profile artefacts of the newest round are ONE build's (round-2 verdict: "tracked profile is not the
profile of HEAD").  tools/profile_round.sh is the only writer of profiles/rNN_*; it stamps every artefact with the commit, the
hash of the kernel sources and the sha256 of the library that ran (tools/stamp.py) and lists the files' own hashes in
profiles/rNN_stamp.json.  This test fails when
  * a file of the round was edited or replaced after the run (hash mismatch), or two artefacts carry different stamps;
  * the rocprofv3 kernel statistics lack a kernel that the bench line's roofline.kernels[] names.
It warns (does not fail: a kernel edit after the last GPU call must not turn the CPU suite red) when the kernel sources have
changed since the profile was taken.
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def _newest_round():
    rounds = sorted(re.match(r"(r\d+)_stamp\.json", os.path.basename(f)).group(1) for f in glob.glob(os.path.join(PROF, "r*_stamp.json")))
    return rounds[-1] if rounds else None


def test_round_profile_artefacts_are_one_build():
    R = _newest_round()
    if R is None:
        pytest.skip("no stamped profile round yet (tools/profile_round.sh writes profiles/rNN_stamp.json)")
    st = json.load(open(os.path.join(PROF, f"{R}_stamp.json")))
    stamp = st["stamp"]
    assert stamp["lib_sha256"] and stamp["source_sha256"]
    for name, sha in st["files"].items():
        got = hashlib.sha256(open(os.path.join(PROF, name), "rb").read()).hexdigest()
        assert got == sha, f"profiles/{name} changed after tools/profile_round.sh wrote it"
    bench = json.load(open(os.path.join(PROF, f"{R}_bench_b32.json")))
    under = json.load(open(os.path.join(PROF, f"{R}_bench_b32_under_rocprof.json")))
    traffic = json.load(open(os.path.join(PROF, f"{R}_hbm_traffic.json")))
    sq = json.load(open(os.path.join(PROF, f"{R}_pmc_sq.json")))
    for what, s in (("bench line", bench["build"]), ("bench line under rocprofv3", under["build"]), ("hbm traffic", traffic["stamp"]),
                    ("sq counters", sq["stamp"])):
        assert s["lib_sha256"] == stamp["lib_sha256"] and s["source_sha256"] == stamp["source_sha256"], what
    # the traffic figure of the headline line is this round's, or absent -- never another build's
    roof = bench["roofline"]
    if roof.get("traffic") is not None:
        assert f"{R}_hbm_traffic.json" in roof["traffic_source"]
        assert roof["traffic"] == traffic["kernels"][roof["kernel"]]["hbm_bytes_per_launch"]
    assert "primary" in roof and 0 < roof["primary"]["frac"] < 1
    # every kernel the roofline names exists in the rocprofv3 statistics of the same build
    with open(os.path.join(PROF, f"{R}_bench_b32_kernel_stats.csv")) as f:
        names = [r["Name"] for r in csv.DictReader(f)]
    flat = " | ".join(n.replace("(anonymous namespace)::", "").replace("void ", "") for n in names)
    missing = []
    for k in roof["kernels"]:
        if k.get("workload"):
            continue  # a slot measured in one of the secondary workloads (log-mel: the Whisper window step), not in the traced headline step
        ident = k["kernel"].split(" (")[0]
        if ident.startswith("row / element-wise") or ident.startswith("positional-conv") or ident.startswith("softmax"):
            continue  # class slots that bundle many small helpers under a descriptive name
        for part in ident.split(" + "):
            part = part.strip()
            if part.startswith("gemm"):
                part = re.sub(r"\s+", " ", part)
                ok = any(re.sub(r"\s+", " ", n).startswith(part) for n in flat.split(" | "))
            else:
                ok = part in flat
            if not ok:
                missing.append(part)
    assert not missing, f"kernels named in {R}_bench_b32.json roofline.kernels[] but absent from {R}_bench_b32_kernel_stats.csv: {missing}"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stamp as S
    if S.source_sha256() != stamp["source_sha256"]:
        warnings.warn(f"kernel sources changed since profiles/{R}_* were measured: re-run tools/profile_round.sh {R} on the GPU box")
