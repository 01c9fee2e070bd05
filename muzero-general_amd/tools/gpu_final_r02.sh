#!/bin/bash
# Runs on the GPU box (through gpurun): the round's final evidence on the final tree.
#   part "tests":    full device suite, default bench line (driver contract), smoke, C3 / C5 bench lines
#   part "profiles": rocprofv3 kernel stats + PMC passes (separate --pmc runs) of the C2 kernel, kernel stats + HBM
#                    traffic of the C3 wave-per-tree kernel, kernel stats of C5 / C4
# Outputs under gpurun_out/$TAG/; copy what should be judged into profiles/.
PART=${1:-tests}
TAG=${2:-r02final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ "$PART" = "tests" ]; then
  timeout 1000 python -m pytest tests -m gpu -q -s --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  timeout 500 python bench.py > $OUT/bench_default.log 2>&1
  timeout 200 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1
  for w in c3 c5; do
    timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --cpu-seconds 0 --selfplay-moves 0 > $OUT/bench_$w.log 2>&1
  done
else
  BENCH="python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --also none --selfplay-moves 0"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_stats -o run -- $BENCH > $OUT/rocprof_c2_stats.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c2_pmc_fetch -o run -- $BENCH > $OUT/rocprof_c2_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c2_pmc_write -o run -- $BENCH > $OUT/rocprof_c2_write.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/c2_pmc_sq -o run -- $BENCH > $OUT/rocprof_c2_sq.log 2>&1
  B3="python bench.py --workload c3 --steps 20 --warmup 3 --cpu-seconds 0 --also none --selfplay-moves 0"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_stats -o run -- $B3 > $OUT/rocprof_c3_stats.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c3_pmc_fetch -o run -- $B3 > $OUT/rocprof_c3_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c3_pmc_write -o run -- $B3 > $OUT/rocprof_c3_write.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/c3_pmc_sq -o run -- $B3 > $OUT/rocprof_c3_sq.log 2>&1
  B5="python bench.py --workload c5 --steps 10 --warmup 3 --cpu-seconds 0 --also none --selfplay-moves 0"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -o run -- $B5 > $OUT/rocprof_c5_stats.log 2>&1
  B4="python bench.py --workload c4 --steps 3 --warmup 1 --cpu-seconds 0 --also none --selfplay-moves 0"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_stats -o run -- $B4 > $OUT/rocprof_c4_stats.log 2>&1
  timeout 120 python muzero-general_amd/tools/fused_phase_profile.py --workload c3 > $OUT/phase_c3.txt 2>&1
  timeout 120 python muzero-general_amd/tools/fused_phase_profile.py --workload c5 > $OUT/phase_c5.txt 2>&1
  timeout 120 python muzero-general_amd/tools/fused_phase_profile.py --workload c2 > $OUT/phase_c2.txt 2>&1
fi
find $OUT -size +8M -delete
ls -laR $OUT > $OUT/files_$PART.txt
