#!/bin/bash
# Round 4, last check on the final tree: the -m gpu suite, smoke, the default bench line.
TAG=${1:-r04final3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc $?" >> $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
echo "bench rc $?" >> $OUT/bench_default.err
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu.log | tail -20
tail -2 $OUT/smoke.log
