#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel stats + PMC passes of the streamed MFMA engine on one
# configuration.  Outputs under gpurun_out/$TAG/ ; summarise with tools/rocprof_summary.py into profiles/.
GAME=${1:-gomoku}
BATCH=${2:-512}
TAG=${3:-streamed_$GAME}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python muzero-general_amd/tools/streamed_bench.py $GAME $BATCH --iters 5"
$CMD > $OUT/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $CMD > $OUT/rocprof_stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o run -- $CMD > $OUT/rocprof_mfma.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_lds -o run -- $CMD > $OUT/rocprof_lds.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- $CMD > $OUT/rocprof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- $CMD > $OUT/rocprof_write.log 2>&1
python muzero-general_amd/tools/rocprof_summary.py $OUT rb_ > $OUT/summary.txt 2>&1
find $OUT -size +4M -delete
