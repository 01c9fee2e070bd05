#!/bin/bash
# Round 4: the whole -m gpu suite with towers + tails, A/B timings, the default bench line.
TAG=${1:-r04f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
SB="python muzero-general_amd/tools/streamed_bench.py"
{
for b in 1024 4608; do
  $SB connect4 $b --mode 3 --iters 10
  MZX_RB_TAIL=0 $SB connect4 $b --mode 3 --iters 10
done
$SB gomoku 512 --mode 1 --iters 5
MZX_RB_TAIL=0 $SB gomoku 512 --mode 1 --iters 5
$SB atari 512 --mode 1 --iters 3
} > $OUT/ab.log 2>&1
timeout 900 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
echo "bench rc $?" >> $OUT/bench_default.err
grep -v amdgpu.ids $OUT/ab.log
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu.log | tail -30
