"""Summarise one rocprofv3 --pmc pass (counter_collection.csv) per kernel: tools/pmc_summary.py DIR OUT.json [top_n].

Sums every counter over the launches of a kernel and divides by the launch count; adds the ratios the roofline discussion uses
when their counters are present (MFMA busy share of CU-busy cycles, LDS-active share).  SQ_* counters are summed over the chip
by rocprofv3 (all SEs / XCDs)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import stamp  # noqa: E402

d, out_path = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = max(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)  # the newest pass (the directory keeps older ones)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[k].add(r["Dispatch_Id"])
res = {}
for k, c in acc.items():
    n = len(launches[k])
    e = {"launches": n}
    for name, v in sorted(c.items()):
        e[name + "_per_launch"] = round(v / n, 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("SQ_BUSY_CU_CYCLES", 0) > 0:
        # the MFMA counter ticks per SIMD (4 per CU), the CU-busy counter per CU: /4 = share of the matrix pipes' cycles
        e["mfma_busy_share_of_simd_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CU_CYCLES"] / 4.0, 4)
    if "SQ_LDS_IDX_ACTIVE" in c and c.get("SQ_BUSY_CU_CYCLES", 0) > 0:
        e["lds_active_share_of_cu_busy"] = round(c["SQ_LDS_IDX_ACTIVE"] / c["SQ_BUSY_CU_CYCLES"], 4)
    if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES", 0) > 0:
        e["wave_wait_share"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
    res[k] = e
order = sorted(res, key=lambda k: -sum(v for n, v in acc[k].items() if n in ("SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")) or -res[k]["launches"])
json.dump({"stamp": stamp.current(), "source": f.split("gpurun_out/")[-1], "kernels": {k: res[k] for k in order[:top]}}, open(out_path, "w"), indent=1)
print("wrote", min(top, len(res)), "of", len(res), "kernels to", out_path)
