#!/bin/bash
# Round 4, check of the tree with grouped head launches: the -m gpu suite, smoke, the default bench line, and a kernel
# trace of connect4 recurrent_inference at 512 / 4608 samples (per-kernel durations of tower + the two grouped launches).
TAG=${1:-r04final4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc $?" >> $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
echo "bench rc $?" >> $OUT/bench_default.err
SB="python muzero-general_amd/tools/streamed_bench.py"
for b in 512 4608; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$b -o run -- $SB connect4 $b --mode 3 --iters 20 > $OUT/rocprof$b.log 2>&1
  echo "== connect4 recurrent_inference, $b samples" >> $OUT/rocprof_heads.txt
  python muzero-general_amd/tools/rocprof_summary.py $OUT/stats$b rb_ 2>&1 | head -8 >> $OUT/rocprof_heads.txt
done
rm -rf $OUT/stats512 $OUT/stats4608
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu.log | tail -20
tail -2 $OUT/smoke.log
cat $OUT/rocprof_heads.txt
