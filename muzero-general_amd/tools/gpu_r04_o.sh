#!/bin/bash
# Round 4: grouped head launches (MZX_RB_HEADS=2: one rb_gemm_multi_kernel launch per MLP level): parity, A/B.
TAG=${1:-r04o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_streamed.py -q -k "tower" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
SB="python muzero-general_amd/tools/streamed_bench.py"
{
for b in 512 4608; do for h in 3 2 0 3; do echo "heads=$h"; MZX_RB_HEADS=$h $SB connect4 $b --mode 3 --iters 20; done; done
for h in 3 2; do echo "heads=$h"; MZX_RB_HEADS=$h $SB gomoku 512 --mode 1 --iters 5; done
} 2>&1 | grep -v amdgpu
B="python bench.py --steps 3 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0"
{
for h in 3 2 3 2; do echo "== c4-1024 heads=$h"; MZX_RB_HEADS=$h $B --workload c4 --trees 1024; done
for h in 3 2; do echo "== c4-large heads=$h"; MZX_RB_HEADS=$h python bench.py --workload c4-large --steps 1 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0; done
for h in 3 2; do echo "== gomoku heads=$h"; MZX_RB_HEADS=$h python bench.py --workload gomoku --steps 1 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0; done
} > $OUT/ab.log 2>&1
python - $OUT <<'PY'
import json, sys
for ln in open(sys.argv[1] + "/ab.log"):
    if ln.startswith("=="): print(ln.strip())
    if ln.startswith("{"):
        j = json.loads(ln)
        print("   ", j["config"]["trees_per_gpu"], round(j["value"]), round(j["ms_per_step"], 2), round(j["roofline"]["frac"], 4))
PY
