set -e
ROOT=$PWD
export TMPDIR=/tmp PYTHONPATH=$ROOT
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_attn_a -- python3 $ROOT/tools/dbg/attn_once.py > /dev/null 2> $ROOT/gpurun_out/pmc_attn_a.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_attn_b -- python3 $ROOT/tools/dbg/attn_once.py > /dev/null 2> $ROOT/gpurun_out/pmc_attn_b.err
cd $ROOT
python3 tools/pmc_summary.py gpurun_out/pmc_attn_a gpurun_out/pmc_attn_a.json 10
python3 tools/pmc_summary.py gpurun_out/pmc_attn_b gpurun_out/pmc_attn_b.json 10
