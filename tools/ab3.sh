#!/bin/bash
# Same-box comparison of several builds of the same ABI: tools/ab3.sh N lib1.so lib2.so ...  (alternating bench.py runs)
N=${1:-2}; shift
show='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], d["value"], d["ms_per_step"], " | ".join("%s %.0f" % (k["kernel"].split("(")[0][:34], k["us_per_step"]) for k in r["kernels"][:9]))'
for i in $(seq $N); do
  for lib in "$@"; do
    SSAK_HIP_LIB=$PWD/$lib python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show" "$(basename $lib .so)"
  done
done
