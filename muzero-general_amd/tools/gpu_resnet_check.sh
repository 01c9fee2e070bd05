#!/bin/bash
# GPU box: parity of the fused residual engine, then the whole GPU suite, then C3/C4 bench lines
# (per-operator network vs fused MFMA network).  Outputs under gpurun_out/$TAG/.
TAG=${1:-resnet}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "fused_resnet" > $OUT/pytest_fused.log 2>&1; echo "rc=$?" >> $OUT/pytest_fused.log
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
for W in c3 c4; do
  for NM in per-operator fused; do
    timeout 300 python bench.py --workload $W --net-mode $NM --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/bench_${W}_${NM}.log 2>&1; echo "rc=$?" >> $OUT/bench_${W}_${NM}.log
  done
done
