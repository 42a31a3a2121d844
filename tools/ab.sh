#!/bin/bash
# A/B of two builds on the SAME box (the pool's boxes differ by several per cent): alternates bench.py runs of the current
# libssak_hip.so and of the build named by $SSAK_AB_BASE (default tools/ab_base.so (git-ignored like every .so), kept OUTSIDE the package so that it never
# ships next to the product library; it must have the ABI of the current binding, ssak_amd/hip.py refuses another one),
# printing utt/s, ms/step and the top kernel slots.  Make the baseline with: git stash; make; cp ssak_amd/lib/libssak_hip.so tools/ab_base.so; git stash pop; make
BASE=${SSAK_AB_BASE:-$PWD/tools/ab_base.so}
N=${1:-3}
show='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], d["value"], d["ms_per_step"], " | ".join("%s %.0f" % (k["kernel"].split("(")[0][:34], k["us_per_step"]) for k in r["kernels"][:9]))'
for i in $(seq $N); do
  python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show" "new "
  SSAK_HIP_LIB=$BASE python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show" "base"
done
