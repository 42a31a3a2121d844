# print one profiler slot of bench.py with / without an environment switch: tools/dbg/slot.sh VAR=VALUE substring [N]
V=$1; S=$2; N=${3:-2}
show='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], d["value"], d["ms_per_step"], [(k["kernel"][:40], k["us_per_step"], k["frac"]) for k in r["kernels"] if sys.argv[2] in k["kernel"]])'
for i in $(seq $N); do
  python bench.py --steps 40 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show" "default " "$S"
  env $V python bench.py --steps 40 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show" "$V" "$S"
done
