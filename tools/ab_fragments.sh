#!/bin/bash
# Same-box A/B of the fragment-ordered-weights option (SSAK_W2V2_OPT_FRAGMENT_WEIGHTS): interleaved bench.py runs, one line each.
show='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
pc=[k for k in r["kernels"] if "p8bd" in k["kernel"] or "<3, false" in k["kernel"] or "row / element" in k["kernel"]]
print(sys.argv[1], d["value"], d["ms_per_step"], " | ".join("%s %.0f us (%.2f)" % (k["kernel"][:34], k["us_per_step"], k["frac"]) for k in pc))'
for i in 1 2 3; do
  SSAK_BENCH_FRAGMENTS=1 python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show" "fragments"
  python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show" "lds      "
done
