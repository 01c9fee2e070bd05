#!/bin/bash
# Round 4, first GPU call: at-size parity of the streamed engine, per-kernel rocprofv3 of the FINAL rb_gemm_kernel<8,1>
# for gomoku alone and for connect4 (large shard) alone, the default bench line.
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_streamed_at_size.py -q -s -x > $OUT/pytest_at_size.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_at_size.log
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
timeout 600 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
echo "bench rc $?" >> $OUT/bench_default.err
find $OUT -size +4M -delete
tail -5 $OUT/pytest_at_size.log
