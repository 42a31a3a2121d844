#!/bin/bash
# Same-box A/B of the shipped library against another build of the same ABI: tools/ab_quick.sh <other.so> [rounds]
# (utterances/s and ms per step of the long run, which is free of start-up effects, + the slots that differ)
OTHER=$1
N=${2:-3}
show='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], d["long_run"]["value"], d["long_run"]["ms_per_step"], "|", " | ".join("%s %.0f" % (k["kernel"][:40], k["us_per_step"]) for k in r["kernels"][:12] if "gemm" in k["kernel"]))'
for i in $(seq $N); do
  python bench.py --steps 20 --warmup 5 --long-steps 100 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "$show" "ship "
  SSAK_HIP_LIB=$OTHER python bench.py --steps 20 --warmup 5 --long-steps 100 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "$show" "other"
done
