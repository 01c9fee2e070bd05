#!/bin/bash
# Round 4, last check on the final tree: the -m gpu suite, smoke, the default bench line, and the rocprofv3 kernel-trace
# summary of the same bench command.
TAG=${1:-r04final5}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc $?" >> $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
echo "bench rc $?" >> $OUT/bench_default.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -o run -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
python muzero-general_amd/tools/rocprof_summary.py $OUT/stats > $OUT/rocprof_bench_default.txt 2>&1
rm -rf $OUT/stats
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu.log | tail -20
tail -2 $OUT/smoke.log
head -12 $OUT/rocprof_bench_default.txt | cut -c1-160
