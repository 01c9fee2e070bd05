#!/bin/bash
# Round 4: rb_heads_kernel (all head MLPs of a program in one launch) -- correctness, then A/B.
TAG=${1:-r04h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_streamed.py tests/test_gpu_streamed_at_size.py tests/test_gpu_parity.py -q -s -k "tower or operator_by_operator or heads or at_size or full_size_residual or streamed_search or two_half or row_kernels" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
B="python bench.py --workload c4 --steps 3 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0"
{
for t in 1024 1536 3072; do
  echo "== $t trees: heads kernel"; $B --trees $t
  echo "== $t trees: MZX_RB_HEADS=0"; MZX_RB_HEADS=0 $B --trees $t
done
echo "== c4-large"; python bench.py --workload c4-large --steps 1 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0
echo "== gomoku"; python bench.py --workload gomoku --steps 1 --warmup 0 --also none --cpu-seconds 0 --selfplay-moves 0
echo "== 640 / 768 trees"; $B --trees 640; MZX_SEARCH_STREAMED_MIN=0 $B --trees 640; $B --trees 768
} > $OUT/ab.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail
python - <<'PY'
import json
for ln in open("gpurun_out/r04h/ab.log"):
    if ln.startswith("=="): print(ln.strip())
    if ln.startswith("{"):
        j = json.loads(ln)
        print("   ", j["config"]["trees_per_gpu"], round(j["value"]), round(j["ms_per_step"], 2), round(j["roofline"]["frac"], 4), j["config"]["search_kernel"][:40], j["config"].get("instantiations", {}))
PY
