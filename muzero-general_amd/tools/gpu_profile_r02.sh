#!/bin/bash
# Runs on the GPU box (through gpurun): round-2 evidence for the C2 whole-search kernel (mzx::fc2_search_kernel):
# bench line, rocprofv3 kernel stats and PMC passes (separate --pmc runs, as MI355X_MICROARCH.md prescribes),
# plus the bench lines of the other single-GPU workloads.  Outputs under gpurun_out/$TAG/; copy what should be
# judged into profiles/.
TAG=${1:-r02prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --also none --selfplay-moves 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $BENCH > $OUT/rocprof_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- $BENCH > $OUT/rocprof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- $BENCH > $OUT/rocprof_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -o run -- $BENCH > $OUT/rocprof_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o run -- $BENCH > $OUT/rocprof_sq2.log 2>&1
for w in c3 c5; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --cpu-seconds 0 --selfplay-moves 0 > $OUT/bench_$w.log 2>&1
done
find $OUT -size +8M -delete
ls -laR $OUT > $OUT/files.txt
