#!/bin/bash
# Round 4: routing of connect4 by shard size (streamed towers vs whole-search kernel; two half-shards or one), new tests.
TAG=${1:-r04g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streamed_at_size.py -q -s -k "full_size_residual or override or connect4-512 or connect4-1024" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
B="python bench.py --workload c4 --steps 3 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0"
{
for t in 1024 1536 2048 3072; do
  echo "== $t trees: routed (split by default)"; $B --trees $t
  echo "== $t trees: routed, undivided"; MZX_ROW_SPLIT_MIN=0 $B --trees $t
done
echo "== 512 trees: whole-search kernel / forced streamed"; $B --trees 512; MZX_SEARCH_STREAMED_MIN=512 $B --trees 512
echo "== 768 trees"; $B --trees 768; MZX_SEARCH_STREAMED_MIN=512 $B --trees 768
} > $OUT/c4_routing.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail
python - <<'PY'
import json
for ln in open("gpurun_out/r04g/c4_routing.log"):
    if ln.startswith("=="): print(ln.strip())
    if ln.startswith("{"):
        j = json.loads(ln)
        print("   ", j["config"]["trees_per_gpu"], round(j["value"]), round(j["ms_per_step"], 2), round(j["roofline"]["frac"], 4), j["config"]["search_kernel"][:40], j["config"].get("instantiations", {}).get("half_shards"))
PY
