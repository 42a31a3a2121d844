import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], d["value"], d["ms_per_step"])
for k in r["kernels"]:
    if "(N = " in k["kernel"] or "fragment" in k["kernel"]:
        print("   %-62s n=%5.1f avg %7.1f us  step %7.1f us (%.2f)" % (k["kernel"][:62], k["launches_per_step"], k["avg_launch_us"], k["us_per_step"], k["frac"]))
