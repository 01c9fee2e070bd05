#!/bin/bash
# Round 4 experiment (profiles/r04_tower_experiments.txt section 9): uneven half-shards for connect4.  MZX_ROW_SPLIT_FIRST
# existed only in the experiment build (rb_split_first: `first` = that many trees); the even split won and the knob is gone.
TAG=${1:-r04l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --workload c4 --steps 3 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0"
{
for f in 512 640 768 896; do echo "== 1024 trees, first half $f"; MZX_ROW_SPLIT_FIRST=$f $B --trees 1024; done
for f in 1024 1280 1536; do echo "== 2048 trees, first half $f"; MZX_ROW_SPLIT_FIRST=$f $B --trees 2048; done
for f in 768 512; do echo "== 1280 trees, first half $f"; MZX_ROW_SPLIT_FIRST=$f $B --trees 1280; done
} > $OUT/split.log 2>&1
