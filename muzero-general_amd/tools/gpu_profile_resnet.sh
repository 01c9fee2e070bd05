#!/bin/bash
# GPU box: rocprofv3 kernel stats + PMC passes of the residual-network workloads (C3 / C4).
TAG=${1:-resnet_prof}
WL=${2:-"c3 c4"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for W in $WL; do
  BENCH="python bench.py --workload $W --steps 2 --warmup 1 --cpu-seconds 0"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_stats -o run -- $BENCH > $OUT/${W}_stats.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/${W}_pmc_sq -o run -- $BENCH > $OUT/${W}_pmc_sq.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/${W}_pmc_mfma -o run -- $BENCH > $OUT/${W}_pmc_mfma.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${W}_pmc_fetch -o run -- $BENCH > $OUT/${W}_pmc_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${W}_pmc_write -o run -- $BENCH > $OUT/${W}_pmc_write.log 2>&1
done
find $OUT -name '*kernel_trace.csv' -size +4M -delete
find $OUT -size +12M -delete
