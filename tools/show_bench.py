"""Development aid: print the headline figures and the per-kernel survey of one bench.py JSON line.
usage: python tools/show_bench.py gpurun_out/check_bench.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
print(d["value"], d["unit"], d["ms_per_step"], "ms; primary", d["roofline"]["primary"]["achieved"], d["roofline"]["primary"]["frac"],
      "share", d["roofline"]["primary"]["share_of_step"], "long", d.get("long_run"))
for k in d["roofline"]["kernels"]:
    print(f'{k["kernel"][:86]:86s} {k["launches_per_step"]:5.1f} {k["us_per_step"]:8.1f} {k["achieved"]:8.1f} {k["frac"] if k["frac"] is not None else float("nan"):.3f}')
if d.get("secondary"):
    print(json.dumps(d["secondary"])[:2000])
if d.get("ingest"):
    print(d["ingest"])
