#!/bin/bash
# Round 4: tower kernel with early requests (next layer's first weight fragment, BatchNorm terms): parity + timings.
TAG=${1:-r04k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_streamed.py -q -k "tower or operator_by_operator" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
SB="python muzero-general_amd/tools/streamed_bench.py"
{
$SB connect4 512 --mode 3 --iters 20
$SB connect4 1024 --mode 3 --iters 20
$SB connect4 4608 --mode 3 --iters 10
$SB gomoku 512 --mode 1 --iters 5
$SB atari 512 --mode 1 --iters 3
} > $OUT/nn.log 2>&1
grep -v amdgpu $OUT/nn.log
tail -3 $OUT/pytest.log
