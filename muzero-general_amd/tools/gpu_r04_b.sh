#!/bin/bash
# Round 4, second GPU call: the at-size parity file in full (no -x), then the default bench (refill self-play legs).
TAG=${1:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_streamed_at_size.py -q -s > $OUT/pytest_at_size.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_at_size.log
timeout 600 python bench.py --also c4 > $OUT/bench_c2_c4.log 2> $OUT/bench_c2_c4.err
echo "bench rc $?" >> $OUT/bench_c2_c4.err
tail -5 $OUT/pytest_at_size.log
