#!/bin/bash
# Runs on the GPU box (through gpurun): round-3 final evidence.  rocprofv3 kernel stats of the default bench command
# and PMC traffic passes (separate --pmc runs, MI355X_MICROARCH.md) of the large-shard connect4 workload on the
# streamed engine.  Outputs under gpurun_out/$TAG/, summaries (tools/rocprof_summary.py) beside them.
TAG=${1:-r03final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python bench.py --cpu-seconds 0 --selfplay-moves 0"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench/stats -o run -- $BENCH > $OUT/rocprof_bench_stats.log 2>&1
python muzero-general_amd/tools/rocprof_summary.py $OUT/bench > $OUT/summary_bench.txt 2>&1
C4L="python bench.py --workload c4-large --steps 1 --warmup 0 --cpu-seconds 0 --selfplay-moves 0"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c4l/pmc_fetch -o run -- $C4L > $OUT/rocprof_c4l_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c4l/pmc_write -o run -- $C4L > $OUT/rocprof_c4l_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/c4l/pmc_mfma -o run -- $C4L > $OUT/rocprof_c4l_mfma.log 2>&1
python muzero-general_amd/tools/rocprof_summary.py $OUT/c4l rb_gemm_kernel > $OUT/summary_c4l.txt 2>&1
find $OUT -size +4M -delete
