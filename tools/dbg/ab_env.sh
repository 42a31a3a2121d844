# A/B of one build with / without an environment switch on the same box: tools/dbg/ab_env.sh VAR=VALUE [N]
V=$1
N=${2:-3}
show='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], d["value"], d["ms_per_step"], " | ".join("%s %.0f" % (k["kernel"].split("(")[0][:34], k["us_per_step"]) for k in r["kernels"][:9]))'
for i in $(seq $N); do
  python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show" "default "
  env $V python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show" "$V"
done
