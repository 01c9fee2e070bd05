#!/bin/bash
# Round 4: rb_heads_kernel, third version (weights through LDS slabs): parity, kernel time, A/B.
TAG=${1:-r04n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_streamed.py -q -k "tower" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
SB="python muzero-general_amd/tools/streamed_bench.py"
CMD="$SB connect4 512 --mode 3 --iters 20"
MZX_RB_HEADS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $CMD > $OUT/rocprof.log 2>&1
python muzero-general_amd/tools/rocprof_summary.py $OUT rb_ 2>&1 | head -6
{
for b in 512 4608; do MZX_RB_HEADS=1 $SB connect4 $b --mode 3 --iters 20; $SB connect4 $b --mode 3 --iters 20; done
MZX_RB_HEADS=1 $SB gomoku 512 --mode 1 --iters 5; $SB gomoku 512 --mode 1 --iters 5
} 2>&1 | grep -v amdgpu
B="python bench.py --workload c4 --steps 3 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0"
{
echo "== 1024 heads"; MZX_RB_HEADS=1 $B --trees 1024
echo "== 1024 no heads"; $B --trees 1024
echo "== c4-large heads"; MZX_RB_HEADS=1 python bench.py --workload c4-large --steps 1 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0
} > $OUT/ab.log 2>&1
python - <<'PY'
import json
for ln in open("gpurun_out/r04n/ab.log"):
    if ln.startswith("=="): print(ln.strip())
    if ln.startswith("{"):
        j = json.loads(ln)
        print("   ", j["config"]["trees_per_gpu"], round(j["value"]), round(j["ms_per_step"], 2), round(j["roofline"]["frac"], 4))
PY
