#!/bin/bash
# Round profile on the GPU box (run from the repository root; locally: `python tools/stamp.py write` first, so that the box
# knows the commit).  THE ONLY WRITER of profiles/rNN_*: the bench line, the rocprofv3 kernel statistics, the HBM traffic and
# the SQ counter passes of ONE build, each in its own rocprofv3 run (counters never combined with trace domains other than
# --kernel-trace), every artefact stamped with the commit, the hash of the kernel sources and the sha256 of the library that
# ran (tools/stamp.py); tests/test_profiles.py fails when the artefacts of a round disagree.
set -e
R=${1:-r04}
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out profiles
rm -rf gpurun_out/prof_${R} gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
python3 tools/stamp.py show > gpurun_out/${R}_stamp_before.json
# 1. the plain bench line (no profiler attached): what the driver will measure
python3 bench.py --batch 32 --steps 100 --warmup 5 > gpurun_out/${R}_bench_b32.json 2> gpurun_out/${R}_bench_b32.err
echo "bench done: $(python3 -c "import json;d=json.load(open('gpurun_out/${R}_bench_b32.json'));print(d['value'], d['ms_per_step'])")"
CMD="python3 $ROOT/bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --long-steps 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${R} -- python3 $ROOT/bench.py --batch 32 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --long-steps 0 > $ROOT/gpurun_out/prof_${R}_bench.json 2> $ROOT/gpurun_out/prof_${R}.err
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_fetch -- $CMD > /dev/null 2> $ROOT/gpurun_out/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_write -- $CMD > /dev/null 2> $ROOT/gpurun_out/pmc_write.err
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sq -- $CMD > /dev/null 2> $ROOT/gpurun_out/pmc_sq.err
echo "sq done"
# the Whisper log-mel front end (a13) on its own: per-kernel times of 23 calls at B = 8
rm -rf $ROOT/gpurun_out/prof_${R}_logmel
PYTHONPATH=$ROOT rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${R}_logmel -- python3 $ROOT/tools/prof_logmel.py 8 > $ROOT/gpurun_out/${R}_logmel.log 2> $ROOT/gpurun_out/prof_${R}_logmel.err
echo "log-mel done: $(grep log-mel $ROOT/gpurun_out/${R}_logmel.log)"
cd $ROOT
# this library next to hipBLASLt (torch.matmul) on the train step's GEMM shapes: reference point only
PYTHONPATH=$ROOT python3 tools/bench_vendor_gemm.py 2> /dev/null > gpurun_out/${R}_vendor_gemm.log
echo "vendor comparison done"
python3 tools/pmc_traffic.py profiles/${R}_hbm_traffic.json
python3 tools/pmc_summary.py gpurun_out/pmc_sq profiles/${R}_pmc_sq.json 40
f=$(ls -t gpurun_out/prof_${R}/*/*kernel_stats.csv | head -1)  # (newest: the directory was removed above, one process writes one file)
cp $f profiles/${R}_bench_b32_kernel_stats.csv
cp $(ls -t gpurun_out/prof_${R}_logmel/*/*kernel_stats.csv | head -1) profiles/${R}_logmel_kernel_stats.csv
grep log-mel gpurun_out/${R}_logmel.log > profiles/${R}_logmel.log
cp gpurun_out/${R}_vendor_gemm.log profiles/${R}_vendor_gemm.log
cp gpurun_out/prof_${R}_bench.json profiles/${R}_bench_b32_under_rocprof.json
cp gpurun_out/${R}_bench_b32.json profiles/${R}_bench_b32.json
# second bench line AFTER the traffic file exists: its roofline.traffic is read from this round's (same-library) passes
python3 bench.py --batch 32 --steps 100 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/${R}_bench_b32_b.json 2>> gpurun_out/${R}_bench_b32.err
python3 - <<PY
import hashlib, json, sys
sys.path.insert(0, "tools")
import stamp
R = "${R}"
a, b = json.load(open(f"gpurun_out/{R}_bench_b32.json")), json.load(open(f"gpurun_out/{R}_bench_b32_b.json"))
# the committed line is the SECOND run (its roofline.traffic comes from this round's passes) with the cpu_baseline leg of the
# first; both throughputs are listed (same build, same box, minutes apart) -- no picking
best = dict(b, **{k: v for k, v in a.items() if k not in b})  # every side leg of the first run (cpu_baseline, secondary, ingest, online ...)
if isinstance(best.get("online"), dict) and "value" in best["online"]:
    best["online"]["resident_value_same_run"] = a["value"]  # vs_resident_batch was taken against the first run's own figure
lm = [k for k in a["roofline"].get("kernels", []) if k.get("workload")]  # (the log-mel slot rides on the secondary workloads of the first run)
best["roofline"]["kernels"] = best["roofline"].get("kernels", []) + lm
best["runs"] = [{"value": a["value"], "ms_per_step": a["ms_per_step"]}, {"value": b["value"], "ms_per_step": b["ms_per_step"]}]
json.dump(best, open(f"profiles/{R}_bench_b32.json", "w"))
files = [f"{R}_bench_b32.json", f"{R}_bench_b32_under_rocprof.json", f"{R}_bench_b32_kernel_stats.csv", f"{R}_hbm_traffic.json", f"{R}_pmc_sq.json",
         f"{R}_logmel_kernel_stats.csv", f"{R}_logmel.log", f"{R}_vendor_gemm.log"]
st = stamp.current()
before = json.load(open(f"gpurun_out/{R}_stamp_before.json"))
assert before["lib_sha256"] == st["lib_sha256"] and before["source_sha256"] == st["source_sha256"], "the tree changed while profiling"
json.dump({"stamp": st, "round": R, "files": {f: hashlib.sha256(open("profiles/" + f, "rb").read()).hexdigest() for f in files}},
          open(f"profiles/{R}_stamp.json", "w"), indent=1)
print("stamp", st)
PY
cp profiles/${R}_* gpurun_out/ 2>/dev/null || true
head -12 profiles/${R}_bench_b32_kernel_stats.csv
