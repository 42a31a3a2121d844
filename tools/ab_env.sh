#!/bin/bash
# Same-box A/B of bench.py's own switches (one library): tools/ab_env.sh [-n ROUNDS] "" "SSAK_BENCH_FRAGMENTS=1" "SSAK_OPT_STREAM=0" ...
# every argument is one set of environment assignments ("" = the defaults); the sets are run in turn, ROUNDS times.
# Switches bench.py reads: SSAK_BENCH_FRAGMENTS=1 (fragment-ordered weight copies), SSAK_BENCH_POSCONV_GEMM=1 (positional convolution
# as Toeplitz GEMMs), SSAK_TILE_ORDER=1 (ticket tile order without a process group), SSAK_OPT_STREAM=0 (optimizer in line).
N=3
if [ "$1" = "-n" ]; then N=$2; shift 2; fi
show='import json,sys
d=json.loads(sys.stdin.read()); lr=d.get("long_run") or {}
print("%-40s %8.1f %7.3f | long %8.1f %7.3f | optimizer tail exposed %s us" % (sys.argv[1] or "(defaults)", d["value"], d["ms_per_step"], lr.get("value", 0), lr.get("ms_per_step", 0),
      (d.get("optimizer_tail") or {}).get("exposed_us_per_step")))'
for i in $(seq $N); do
  for set in "$@"; do
    env $set python bench.py --steps 20 --warmup 5 --long-steps 100 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "$show" "$set"
  done
done
