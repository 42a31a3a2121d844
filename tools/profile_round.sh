#!/bin/bash
# Round profile on the GPU box (run from the repository root): kernel statistics + HBM traffic + SQ counters of the SAME bench
# command, each in its own rocprofv3 run (counters never combined with trace domains other than --kernel-trace).
set -e
R=${1:-r02}
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out profiles
CMD="python3 $ROOT/bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${R} -- python3 $ROOT/bench.py --batch 32 --steps 20 --warmup 3 --no-cpu-baseline > $ROOT/gpurun_out/prof_${R}_bench.json 2> $ROOT/gpurun_out/prof_${R}.err
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_fetch -- $CMD > /dev/null 2> $ROOT/gpurun_out/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_write -- $CMD > /dev/null 2> $ROOT/gpurun_out/pmc_write.err
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sq -- $CMD > /dev/null 2> $ROOT/gpurun_out/pmc_sq.err
echo "sq done"
cd $ROOT
python3 tools/pmc_traffic.py profiles/${R}_hbm_traffic.json
python3 tools/pmc_summary.py gpurun_out/pmc_sq profiles/${R}_pmc_sq.json 30
f=$(ls gpurun_out/prof_${R}/*/*kernel_stats.csv | tail -1)
cp $f profiles/${R}_bench_b32_kernel_stats.csv
cp gpurun_out/prof_${R}_bench.json profiles/${R}_bench_b32_under_rocprof.json
cp profiles/${R}_*.json profiles/${R}_*.csv gpurun_out/ 2>/dev/null || true
head -12 profiles/${R}_bench_b32_kernel_stats.csv
