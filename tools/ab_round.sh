#!/bin/bash
# Same-box comparison of this tree against ANOTHER WHOLE TREE of the repository (an earlier round: its own bench.py, binding and
# library), e.g.   git archive <commit> | tar -x -C ab_r03 && make -C ab_r03 -j8 && tools/ab_round.sh ab_r03 3
# (ab_*/ is git-ignored and travels to the GPU box with the snapshot).  Alternates the two bench.py runs, one line each.
OTHER=$1
N=${2:-3}
show='import json,sys
d=json.loads(sys.stdin.read()); p=d["roofline"].get("primary") or {}
print("%-10s %8.1f utt/s %7.3f ms/step | all-GEMM %s TFLOP/s (%s)" % (sys.argv[1], d["value"], d["ms_per_step"], p.get("achieved"), p.get("frac")))'
for i in $(seq $N); do
  (cd $OTHER && python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1) | python -c "$show" "$OTHER"
  python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary --long-steps 0 2>/dev/null | tail -1 | python -c "$show" "this tree"
done
