#!/bin/bash
# Round 4: samples per workgroup of the tower kernel, by shard size (calibrates rb_tower_shape's cost model).
TAG=${1:-r04e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
SB="python muzero-general_amd/tools/streamed_bench.py"
{
for b in 1024 2048 3072 4608; do
  for t in 2 3 4 5 6; do
    echo "== batch $b T $t"
    MZX_RB_TOWER_T=$t $SB connect4 $b --mode 3 --iters 10
  done
done
for b in 256 512 1024; do
  echo "== atari $b T 1 / 2"
  MZX_RB_TOWER_T=1 $SB atari $b --mode 1 --iters 3
  MZX_RB_TOWER_T=2 $SB atari $b --mode 1 --iters 3
done
} > $OUT/tower_t.log 2>&1
grep -v amdgpu.ids $OUT/tower_t.log
