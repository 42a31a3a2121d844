#!/bin/bash
# Development aid (GPU box): the four-wave GEMM's tests, the library GEMM tests, the vendor comparison and one bench line.
timeout -k 10 300 python -m pytest tests/test_gpu_gemm_p4.py -x -q 2>&1 | tail -5
timeout -k 10 200 python -m pytest tests/test_gpu_ops.py -x -q -k gemm 2>&1 | tail -3
PYTHONPATH=. timeout -k 10 120 python tools/bench_vendor_gemm.py 2>&1 | grep -v amdgpu.ids | head -6
timeout -k 10 500 python bench.py --steps 30 --warmup 3 --no-cpu-baseline ${BENCH_FLAGS} > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err
tail -3 gpurun_out/check_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/check_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["primary"]["achieved"], d["roofline"]["primary"]["share_of_step"], d.get("long_run"))
for k in d["roofline"]["kernels"][:14]:
    print(k["kernel"][:70], k["launches_per_step"], k["us_per_step"], k["achieved"], k["frac"])
print(json.dumps(d.get("secondary"))[:1800])
print(d.get("ingest"))
PY
