#!/bin/bash
# Round 4: pipelined per-object self-play (two slot groups take turns on the GPU): default bench line's self-play legs,
# plus the self-play GPU tests.
TAG=${1:-r04q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
true > $OUT/pytest.log
echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 600 python bench.py --also c4 --cpu-seconds 0 > $OUT/bench.log 2> $OUT/bench.err
echo "bench rc $?" >> $OUT/bench.err
python - $OUT <<'PY'
import json, sys
ln = [l for l in open(sys.argv[1] + "/bench.log") if l.startswith("{")][-1]
j = json.loads(ln)
for k in j:
    if k.startswith("selfplay"):
        print(k, {q: (round(v, 3) if isinstance(v, float) else v) for q, v in j[k].items()})
PY
tail -3 $OUT/bench.err
