#!/bin/bash
TAG=${1:-r04j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python muzero-general_amd/tools/streamed_bench.py connect4 512 --mode 3 --iters 20"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $CMD > $OUT/rocprof.log 2>&1
python muzero-general_amd/tools/rocprof_summary.py $OUT rb_ 2>&1 | head -12
CMD="python muzero-general_amd/tools/streamed_bench.py connect4 4608 --mode 3 --iters 10"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b/stats -o run -- $CMD > $OUT/rocprof_b.log 2>&1
python muzero-general_amd/tools/rocprof_summary.py $OUT/b rb_ 2>&1 | head -12
