#!/bin/bash
# Round 4, third GPU call: the tower kernel -- correctness on the small streamed tests, then A/B timings.
TAG=${1:-r04c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_streamed.py -q -s -k "tower or operator_by_operator or heads" > $OUT/pytest_streamed.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_streamed.log
SB="python muzero-general_amd/tools/streamed_bench.py"
{
for b in 512 1024 2048 4608; do
  $SB connect4 $b --mode 3 --iters 10
  $SB connect4 $b --mode 4 --iters 10
done
MZX_RB_TOWER_T=6 $SB connect4 4608 --mode 3 --iters 10
MZX_RB_TOWER_T=2 $SB connect4 4608 --mode 3 --iters 10
MZX_RB_TOWER_T=3 $SB connect4 1024 --mode 3 --iters 10
$SB gomoku 512 --mode 1 --iters 5
$SB gomoku 512 --mode 5 --iters 5
$SB atari 512 --mode 1 --iters 3
$SB atari 512 --mode 5 --iters 3
$SB atari 256 --mode 1 --iters 3
$SB atari 256 --mode 5 --iters 3
} > $OUT/ab.log 2>&1
grep -v amdgpu.ids $OUT/ab.log
tail -5 $OUT/pytest_streamed.log
