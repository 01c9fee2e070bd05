#!/bin/bash
# Round 4 final evidence: the -m gpu suite, smoke, the default bench line, rocprofv3 kernel stats of the bench command,
# per-kernel stats + PMC passes (separate --pmc runs) of the tower launches for gomoku alone and connect4 alone,
# connect4 by shard size on the streamed engine.
TAG=${1:-r04final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc $?" >> $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
echo "bench rc $?" >> $OUT/bench_default.err
BENCH="python bench.py --cpu-seconds 0 --selfplay-moves 0"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench/stats -o run -- $BENCH > $OUT/rocprof_bench_stats.log 2>&1
python muzero-general_amd/tools/rocprof_summary.py $OUT/bench > $OUT/summary_bench.txt 2>&1
prof() {   # game batch mode tag
  local CMD="python muzero-general_amd/tools/streamed_bench.py $1 $2 --mode $3 --iters 5"
  local D=$OUT/$4
  mkdir -p $D
  $CMD > $D/bench.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o run -- $CMD > $D/rocprof_stats.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -o run -- $CMD > $D/rocprof_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -o run -- $CMD > $D/rocprof_write.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $D/pmc_mfma -o run -- $CMD > $D/rocprof_mfma.log 2>&1
  python muzero-general_amd/tools/rocprof_summary.py $D rb_ > $D/summary.txt 2>&1
}
prof gomoku 512 1 gomoku512
prof connect4 4608 3 c4_4608
{
for t in 1024 2048 3072; do
  python bench.py --workload c4 --trees $t --net-mode streamed --steps 2 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0
  python bench.py --workload c4 --trees $t --steps 2 --warmup 1 --also none --cpu-seconds 0 --selfplay-moves 0
done
} > $OUT/c4_by_shard.log 2>&1
find $OUT -size +4M -delete
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu.log | tail -20
tail -3 $OUT/smoke.log
