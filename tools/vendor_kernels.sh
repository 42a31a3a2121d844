#!/bin/bash
# Which Tensile solutions does hipBLASLt pick where it beats this library's plain GEMM (ffn1, 4096^3, 8192^3)?  Kernel names
# (macro tile MT, depthU, LDS buffering, workgroup shape are spelled out in them) with average durations, from a kernel trace of
# tools/bench_vendor_gemm.py.  Run from the repository root on the GPU box; writes profiles/r06_vendor_solutions.log.
set -e
ROOT=$PWD
export TMPDIR=/tmp
rm -rf gpurun_out/prof_vendor
cd /tmp
PYTHONPATH=$ROOT rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_vendor -- python3 $ROOT/tools/bench_vendor_gemm.py > $ROOT/gpurun_out/r06_vendor_solutions.txt 2> $ROOT/gpurun_out/prof_vendor.err
cd $ROOT
f=$(ls -t gpurun_out/prof_vendor/*/*kernel_stats.csv | head -1)
{
  grep -v amdgpu.ids gpurun_out/r06_vendor_solutions.txt
  echo "---- kernels of that run (rocprofv3 --kernel-trace --stats): name, calls, total ns, average ns"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if n.startswith("Cijk") or "gemm" in n.lower():
        print(f'{r["Calls"]:>5s} {r["TotalDurationNs"]:>12s} {float(r["AverageNs"]):10.0f}  {n}')
PY
} > profiles/r06_vendor_solutions.log
cat profiles/r06_vendor_solutions.log
