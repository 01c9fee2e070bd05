#!/bin/bash
TAG=${1:-r04i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --workload c4 --steps 3 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0"
SB="python muzero-general_amd/tools/streamed_bench.py"
{
for t in 1024 1536; do
  echo "== $t trees: heads kernel"; $B --trees $t
  echo "== $t trees: MZX_RB_HEADS=0"; MZX_RB_HEADS=0 $B --trees $t
done
echo "== c4-large heads"; python bench.py --workload c4-large --steps 1 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0
echo "== c4-large MZX_RB_HEADS=0"; MZX_RB_HEADS=0 python bench.py --workload c4-large --steps 1 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0
} > $OUT/ab.log 2>&1
{
for b in 512 4608; do $SB connect4 $b --mode 3 --iters 20; MZX_RB_HEADS=0 $SB connect4 $b --mode 3 --iters 20; done
} > $OUT/nn.log 2>&1
grep -v amdgpu $OUT/nn.log
python - <<'PY'
import json
for ln in open("gpurun_out/r04i/ab.log"):
    if ln.startswith("=="): print(ln.strip())
    if ln.startswith("{"):
        j = json.loads(ln)
        print("   ", j["config"]["trees_per_gpu"], round(j["value"]), round(j["ms_per_step"], 2), round(j["roofline"]["frac"], 4))
PY
