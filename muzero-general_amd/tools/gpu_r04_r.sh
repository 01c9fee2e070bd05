#!/bin/bash
# Round 4: connect4 by shard size on the final tree (grouped head launches): whole search steps.
TAG=${1:-r04r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --workload c4 --steps 2 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0"
for n in 768 1536 2048 3072; do echo "== $n"; $B --trees $n; done > $OUT/ab.log 2>&1
python - $OUT <<'PY'
import json, sys
for ln in open(sys.argv[1] + "/ab.log"):
    if ln.startswith("{"):
        j = json.loads(ln)
        print("   %5d trees  %.3f M sims/s  %8.2f ms/step  %.4f" % (j["config"]["trees_per_gpu"], j["value"] / 1e6, j["ms_per_step"], j["roofline"]["frac"]))
PY
