#!/bin/bash
# Same-box A/B of builds of the same ABI (the pool's boxes differ by several per cent, so only runs on ONE box compare):
#   tools/ab.sh [-n ROUNDS] other1.so [other2.so ...]
# alternates bench.py runs of the shipped library and of each other build (SSAK_HIP_LIB; ssak_amd/hip.py refuses another ABI)
# and prints, per run: utterances/s and ms per step of the timed region, of the 100-step run after it (free of start-up
# effects), and the GEMM / attention slots of the survey.  Other builds are made next to the product, never inside the package:
#   make OBJ=build/obj_x LIB=tools/ab_x.so EXTRA=-DSOME_VARIANT        (every .so is git-ignored; tools/*.so travel to the GPU box)
N=3
if [ "$1" = "-n" ]; then N=$2; shift 2; fi
show='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]; lr=d.get("long_run") or {}
print("%-12s %8.1f %7.3f | long %8.1f %7.3f | %s" % (sys.argv[1], d["value"], d["ms_per_step"], lr.get("value", 0), lr.get("ms_per_step", 0),
      " | ".join("%s %.0f" % (k["kernel"].replace("gemm_", "").replace("_kernel", "")[:34], k["us_per_step"]) for k in r["kernels"][:12] if "gemm" in k["kernel"] or "attn" in k["kernel"])))'
for i in $(seq $N); do
  python bench.py --steps 20 --warmup 5 --long-steps 100 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "$show" "ship"
  for lib in "$@"; do
    SSAK_HIP_LIB=$PWD/${lib#$PWD/} python bench.py --steps 20 --warmup 5 --long-steps 100 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "$show" "$(basename $lib .so)"
  done
done
