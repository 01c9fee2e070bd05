#!/bin/bash
# GPU jobs of this repository, one parameterised script (run through gpurun from the repository root):
#   gpu_job.sh <job> [tag]      results under gpurun_out/<tag>/ (default tag = the job's name)
# Jobs
#   rt-first     tests of rt_search_kernel, then connect4 whole steps by shard size on the three routes
#   c4-shards    connect4 whole steps by shard size (routes: rt_search_kernel, per-simulation launches, rz_search_kernel)
#   tests        the -m gpu suite + smoke
#   bench        the default bench line (as the driver runs it) + rocprofv3 kernel stats of the same command
#   pmc          PMC passes (separate --pmc runs: FETCH_SIZE, WRITE_SIZE, MFMA-busy) of every default workload's search
#   final        tests + bench + pmc
JOB=${1:?job}
TAG=${2:-$JOB}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
Q="--also none --cpu-seconds 0 --selfplay-moves 0"

c4_shards() {   # trees...
  for t in "$@"; do
    for w in c4 c4-rows; do
      timeout 300 python bench.py --workload $w --trees $t --steps 3 --warmup 1 $Q 2>> $OUT/c4_by_shard.err |
        python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('$w', $t, 'trees: %.4g sims/s, %.2f ms per step, %.4f of the FP32 MFMA peak, kernel %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'][:90]))"
    done
  done
  timeout 300 python bench.py --workload c4-ws --steps 3 --warmup 1 $Q 2>> $OUT/c4_by_shard.err |
    python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('c4-ws 1024 trees: %.4g sims/s, %.2f ms per step, %.4f of the FP32 MFMA peak' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
}

run_tests() {
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
  echo "smoke rc $?" >> $OUT/smoke.log
  timeout ${SUITE_S:-1700} python -m pytest tests -m gpu -q -s -x > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc $?" >> $OUT/pytest_gpu.log
  grep -E "passed|failed|^FAILED|^ERROR|rc " $OUT/pytest_gpu.log | tail -20
  tail -3 $OUT/smoke.log
}

run_bench() {
  timeout 900 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
  echo "bench rc $?" >> $OUT/bench_default.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench/stats -o run -- python bench.py --cpu-seconds 0 --selfplay-moves 0 > $OUT/rocprof_bench_stats.log 2>&1
  python muzero-general_amd/tools/rocprof_summary.py $OUT/bench > $OUT/summary_bench.txt 2>&1
  tail -c 1500 $OUT/bench_default.log
}

run_pmc() {   # one workload per directory: HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes), matrix-pipe occupancy, L2 hit rate
  for w in "$@"; do
    if [ -n "$PMC_DEADLINE" ] && [ $(date +%s) -gt $PMC_DEADLINE ]; then echo "pmc: out of time before $w" >> $OUT/pmc_err.log; continue; fi
    local N=2
    case $w in gomoku|atari|c4-large) N=1 ;; esac
    local CMD="python bench.py --workload $w --steps $N --warmup 0 --repeats 1 $Q"
    local D=$OUT/pmc_$w
    mkdir -p $D
    timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -o run -- $CMD > $D/rocprof_fetch.log 2>&1
    timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -o run -- $CMD > $D/rocprof_write.log 2>&1
    timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $D/pmc_mfma -o run -- $CMD > $D/rocprof_mfma.log 2>&1
    case $w in c4|c4-large|gomoku|atari)     # the tower workloads: L2 hit rate of the weight-fragment stream
      timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $D/pmc_l2 -o run -- $CMD > $D/rocprof_l2.log 2>&1 ;;
    esac
    local TAG=$(timeout 200 $CMD 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.readline())['roofline']['kernel'].replace('whole step: ', ''))")
    python muzero-general_amd/tools/pmc_traffic.py $w $N $D "$TAG" >> $OUT/pmc_entries.jsonl 2>> $OUT/pmc_err.log
    python muzero-general_amd/tools/rocprof_summary.py $D mzx > $D/summary.txt 2>&1
  done
  cat $OUT/pmc_entries.jsonl
}

one() {   # label, bench arguments...: one line per run
  local label=$1; shift
  timeout 300 python bench.py --steps 3 --warmup 1 $Q "$@" 2>> $OUT/err.log |
    python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('%-40s %.4g sims/s, %.2f ms per step, %.4f of peak' % ('$label', d['value'], d['ms_per_step'], d['roofline']['frac']))"
}

case $JOB in
  bench-only)      # the default bench line as the driver runs it; size of the line
    timeout 900 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
    echo "bench rc $?" >> $OUT/bench_default.err
    tail -3 $OUT/bench_default.err
    wc -c $OUT/bench_default.log
    python -c "
import json
d = json.loads(open('$OUT/bench_default.log').read().strip().splitlines()[-1])
print('C2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['weights'], d['config']['mean_leaf_depth'])
for w in d['workloads']: print(w['w'], w['v'], w['ms'], w['roofline']['frac'], w['L'], w['k'])
for k in d:
    if k.startswith('selfplay'): print(k, round(d[k]['steps_per_sec']), d[k].get('search_share'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['kind'], d['cpu_baseline']['cores'])
"
    ;;
  rt-quick)      # the rt tests and the planner's shapes around the BASELINE shard
    timeout 600 python -m pytest tests/test_gpu_tower_search.py -m gpu -q -s -x -k "bit_identical or same_trees or routing" > $OUT/pytest_rt.log 2>&1
    grep -E "passed|failed|^FAILED|^ERROR|Error" $OUT/pytest_rt.log | tail
    {
    for t in 512 768 1024 1536 2048; do
      one "rt $t (planner)" --workload c4 --trees $t
    done
    one "rt 1024 K loops only" --workload c4 --tuning rt_dbg=30
    one "rt 1024 no epilogues" --workload c4 --tuning rt_dbg=2
    one "rt 1024 no tree phases" --workload c4 --tuning rt_dbg=4
    one "rt 1024 no staging / tails" --workload c4 --tuning rt_dbg=8
    one "rt 1024 no heads" --workload c4 --tuning rt_dbg=16
    one "rt 1024 every wave multiplies MT tiles" --workload c4 --tuning rt_short=0
    } > $OUT/rt_quick.txt 2>&1
    cat $OUT/rt_quick.txt
    ;;
  variants)      # library variants built beside the product one (mzx/libmzx_<v>.so): whole steps of the tower workloads
    {
    for v in "" _vA _vB; do
      L=$PWD/muzero-general_amd/mzx/libmzx$v.so
      MZX_LIB=$L one "lib$v: rt 1024" --workload c4
      MZX_LIB=$L one "lib$v: rt 1024 K loops only" --workload c4 --tuning rt_dbg=30
      MZX_LIB=$L one "lib$v: rt 1536" --workload c4 --trees 1536
      MZX_LIB=$L one "lib$v: gomoku" --workload gomoku --steps 1
      MZX_LIB=$L one "lib$v: atari" --workload atari --steps 1
    done
    } > $OUT/variants.txt 2>&1
    cat $OUT/variants.txt
    ;;
  tests-host)      # the whole -m gpu suite + smoke, then the host profile of the batched self-play protocol
    run_tests
    timeout 300 python muzero-general_amd/tools/selfplay_host_profile.py > $OUT/host_profile_batched.txt 2>&1
    head -45 $OUT/host_profile_batched.txt
    ;;
  refweights)      # the at-size parity tests on the reference constructor's weights (strict gates) + the new pipelined-shard test
    timeout 1400 python -m pytest tests/test_gpu_streamed_at_size.py -m gpu -q -s -k "reference" > $OUT/pytest_ref.log 2>&1
    echo "pytest rc $?" >> $OUT/pytest_ref.log
    grep -E "passed|failed|^FAILED|^ERROR|Error|identical to the oracle|loose rows|worst error|diverges" $OUT/pytest_ref.log | cut -c1-260 | tail -60
    timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "pipelined_shard or routed_to_their or full_size_residual" > $OUT/pytest_par.log 2>&1
    grep -E "passed|failed|^FAILED|^ERROR|Error|identical|trees on" $OUT/pytest_par.log | cut -c1-200 | tail -20
    ;;
  rt-sweep)      # forced (trees per workgroup, waves) at the BASELINE shard and around it
    {
    for cfg in "4 8" "2 4" "2 8" "1 4" "3 8" "3 4"; do
      set -- $cfg
      one "1024: $1 trees x $(($2 * 64)) threads" --workload c4 --tuning rt_trees=$1,rt_waves=$2
    done
    for cfg in "3 8" "3 4" "2 4" "1 4" "6 8"; do
      set -- $cfg
      one "768: $1 trees x $(($2 * 64)) threads" --workload c4 --trees 768 --tuning rt_trees=$1,rt_waves=$2
    done
    for cfg in "2 8" "2 4" "1 4" "4 8"; do
      set -- $cfg
      one "512: $1 trees x $(($2 * 64)) threads" --workload c4 --trees 512 --tuning rt_trees=$1,rt_waves=$2
    done
    for cfg in "6 8" "3 4" "4 8" "2 4"; do
      set -- $cfg
      one "1280: $1 trees x $(($2 * 64)) threads" --workload c4 --trees 1280 --tuning rt_trees=$1,rt_waves=$2
    done
    one "1024 planner, no tree phases" --workload c4 --tuning rt_dbg=4
    one "1024 planner, no heads" --workload c4 --tuning rt_dbg=16
    one "1024 planner, no staging / tails" --workload c4 --tuning rt_dbg=8
    one "1024 planner, no epilogues" --workload c4 --tuning rt_dbg=2
    one "1024 planner, nothing but K loops" --workload c4 --tuning rt_dbg=30
    } > $OUT/rt_sweep.txt 2>&1
    cat $OUT/rt_sweep.txt
    ;;
  kloop2)      # the K loop with incremental cursors and requests / scalar work interleaved with the MFMAs
    timeout 600 python -m pytest tests/test_gpu_tower_search.py -m gpu -q -s -x -k "bit_identical or same_trees" > $OUT/pytest_rt.log 2>&1
    grep -E "passed|failed|^FAILED|^ERROR|Error" $OUT/pytest_rt.log | tail
    {
    for t in 512 1024 1536 2048; do
      one "rt $t (planner)" --workload c4 --trees $t
    done
    one "rt 1024 K loops only" --workload c4 --tuning rt_dbg=30
    one "rt 1024, 2 trees x 512 threads <3,1>" --workload c4 --tuning rt_trees=2,rt_waves=8
    one "rt 1024, 2 trees x 512 threads <3,1>, K loops only" --workload c4 --tuning rt_trees=2,rt_waves=8,rt_dbg=30
    one "rt 256, 1 tree x 256 threads <3,1> ONE wave per SIMD, K loops only" --workload c4 --trees 256 --tuning rt_trees=1,rt_waves=4,rt_dbg=30
    one "rt 1536, 3 trees x 512 threads <4,1>" --workload c4 --trees 1536 --tuning rt_trees=3,rt_waves=8
    one "launches 1024" --workload c4-rows
    one "launches 9216" --workload c4-rows --trees 9216
    one "gomoku as shipped, launches" --workload gomoku --steps 1 --tuning rt_search=0
    one "atari as shipped" --workload atari --steps 1
    } > $OUT/kloop2.txt 2>&1
    cat $OUT/kloop2.txt
    timeout 900 python -m pytest tests/test_gpu_streamed.py -m gpu -q -x -k "tower_kernel_layer_by_layer" > $OUT/pytest_tower.log 2>&1
    grep -E "passed|failed|^FAILED|^ERROR|Error" $OUT/pytest_tower.log | tail -5
    ;;
  rt-deal)      # positions dealt to the row tiles for conflict-free LDS reads
    timeout 600 python -m pytest tests/test_gpu_tower_search.py -m gpu -q -s -x -k "bit_identical or same_trees" > $OUT/pytest_rt.log 2>&1
    grep -E "passed|failed|^FAILED|^ERROR|Error" $OUT/pytest_rt.log | tail
    {
    for t in 512 1024 1536 2048; do
      one "rt $t (planner)" --workload c4 --trees $t
    done
    one "rt 1024 K loops only" --workload c4 --tuning rt_dbg=30
    one "rt 1024, 2 trees x 512 threads <3,1>" --workload c4 --tuning rt_trees=2,rt_waves=8
    one "gomoku as shipped, rt_search_kernel" --workload gomoku --steps 1 --tuning rt_search=1
    one "gomoku as shipped, launches" --workload gomoku --steps 1 --tuning rt_search=0
    } > $OUT/rt_deal.txt 2>&1
    cat $OUT/rt_deal.txt
    CMD="python bench.py --workload c4 --steps 2 --warmup 1 $Q"
    timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_a -o run -- $CMD > $OUT/pmc_a.log 2>&1
    python muzero-general_amd/tools/rocprof_summary.py $OUT rt_search > $OUT/summary.txt 2>&1
    grep -E "^SQ_|^GRBM" $OUT/summary.txt | cut -c1-70
    ;;
  rt-plan)      # the planner's choice by shard size, against the per-simulation launches
    timeout 600 python -m pytest tests/test_gpu_tower_search.py -m gpu -q -s -x -k "bit_identical or routing or same_trees" > $OUT/pytest_rt.log 2>&1
    grep -E "passed|failed|^FAILED|^ERROR|Error" $OUT/pytest_rt.log | tail
    {
    for t in 256 512 768 1024 1280 1536 2048 3072 4096 4608 9216; do
      one "rt $t (planner)" --workload c4 --trees $t
      one "launches $t" --workload c4-rows --trees $t
    done
    } > $OUT/rt_plan.txt 2>&1
    cat $OUT/rt_plan.txt
    ;;
  rt-deep)      # fewer, fatter waves: more MFMAs per weight fragment
    {
    one "1024: 2 trees x 256 threads <6,1>, two per CU" --workload c4 --tuning rt_trees=2,rt_waves=4
    one "1024: 2 trees x 256 threads <6,1>, K loops only" --workload c4 --tuning rt_trees=2,rt_waves=4,rt_dbg=30
    one "1024: 4 trees x 512 threads <6,1>, one per CU" --workload c4 --tuning rt_trees=4,rt_waves=8
    one "1024: 4 trees x 512 threads <6,1>, K loops only" --workload c4 --tuning rt_trees=4,rt_waves=8,rt_dbg=30
    one "1536: 2 trees x 256 threads <6,1>, three per CU" --workload c4 --trees 1536 --tuning rt_trees=2,rt_waves=4
    one "1536: 3 trees x 256 threads <8,1>, two per CU" --workload c4 --trees 1536 --tuning rt_trees=3,rt_waves=4
    one "1536: 6 trees x 512 threads <8,1>, one per CU" --workload c4 --trees 1536 --tuning rt_trees=6,rt_waves=8
    one "2048: 4 trees x 512 threads <6,1>" --workload c4 --trees 2048 --tuning rt_trees=4,rt_waves=8
    one "512: 2 trees x 256 threads <6,1>, one per CU" --workload c4 --trees 512 --tuning rt_trees=2,rt_waves=4
    } > $OUT/rt_deep.txt 2>&1
    cat $OUT/rt_deep.txt
    ;;
  kloop)      # what the K loop of <3,1> waits for: variants of the library that leave loads / address arithmetic out (kloop_experiment.sh)
    {
    F="--workload c4 --tuning rt_trees=2,rt_waves=8,rt_dbg=30"
    one "K loops only (product build)" $F
    for n in 1 2 3 4; do
      MZX_LIB=$PWD/muzero-general_amd/mzx/libmzx_exp$n.so one "K loops only, experiment $n" $F
    done
    F1="--workload c4 --trees 256 --tuning rt_trees=1,rt_waves=4,rt_dbg=30"
    one "ONE wave per SIMD: K loops only (product build)" $F1
    for n in 1 2 3 4; do
      MZX_LIB=$PWD/muzero-general_amd/mzx/libmzx_exp$n.so one "ONE wave per SIMD: K loops only, experiment $n" $F1
    done
    } > $OUT/kloop.txt 2>&1
    cat $OUT/kloop.txt
    ;;
  rt-w4)      # 256-thread workgroups, one tree each, four per CU
    {
    for t in 512 1024 2048; do
      one "rt $t, 1 tree x 256 threads" --workload c4 --trees $t --tuning rt_trees=1,rt_waves=4
    done
    one "rt 1024, 1 tree x 256 threads, K loops only" --workload c4 --tuning rt_trees=1,rt_waves=4,rt_dbg=30
    one "rt 1024, 1 tree x 256 threads, no tree phases" --workload c4 --tuning rt_trees=1,rt_waves=4,rt_dbg=4
    one "rt 256, 1 tree x 256 threads" --workload c4 --trees 256 --tuning rt_trees=1,rt_waves=4
    one "rt 768, 1 tree x 256 threads" --workload c4 --trees 768 --tuning rt_trees=1,rt_waves=4
    } > $OUT/rt_w4.txt 2>&1
    cat $OUT/rt_w4.txt
    ;;
  rt-waves)     # 512-thread workgroups (two per CU) against 1024-thread ones (one per CU, twice the trees per tile)
    timeout 600 python -m pytest tests/test_gpu_tower_search.py -m gpu -q -s -x -k "bit_identical or routing" > $OUT/pytest_rt.log 2>&1
    grep -E "passed|failed|^FAILED|^ERROR|Error" $OUT/pytest_rt.log | tail
    {
    for t in 256 512 768 1024 1536 2048 3072 4096 9216; do
      one "rt $t (planner)" --workload c4 --trees $t
    done
    one "rt 1024, 2 trees x 512 threads" --workload c4 --tuning rt_trees=2,rt_waves=8
    one "rt 1024, 4 trees x 1024 threads" --workload c4 --tuning rt_trees=4,rt_waves=16
    one "rt 1024, 4 trees x 1024 threads, K loops only" --workload c4 --tuning rt_trees=4,rt_waves=16,rt_dbg=30
    one "rt 1024, 4 trees x 1024 threads, no heads" --workload c4 --tuning rt_trees=4,rt_waves=16,rt_dbg=16
    one "rt 1024, 4 trees x 1024 threads, no tree phases" --workload c4 --tuning rt_trees=4,rt_waves=16,rt_dbg=4
    one "rt 512, 2 trees x 512 threads" --workload c4 --trees 512 --tuning rt_trees=2,rt_waves=8
    one "rt 512, 2 trees x 1024 threads" --workload c4 --trees 512 --tuning rt_trees=2,rt_waves=16
    one "rt 512, 4 trees x 1024 threads" --workload c4 --trees 512 --tuning rt_trees=4,rt_waves=16
    one "rt 1536, 3 trees x 512 threads" --workload c4 --trees 1536 --tuning rt_trees=3,rt_waves=8
    one "rt 1536, 6 trees x 1024 threads" --workload c4 --trees 1536 --tuning rt_trees=6,rt_waves=16
    one "rt 2048, 4 trees x 1024 threads" --workload c4 --trees 2048 --tuning rt_trees=4,rt_waves=16
    one "rt 2048, 2 trees x 512 threads" --workload c4 --trees 2048 --tuning rt_trees=2,rt_waves=8
    one "rt 3072, 6 trees x 1024 threads" --workload c4 --trees 3072 --tuning rt_trees=6,rt_waves=16
    one "rt 3072, 3 trees x 512 threads" --workload c4 --trees 3072 --tuning rt_trees=3,rt_waves=8
    } > $OUT/rt_waves.txt 2>&1
    cat $OUT/rt_waves.txt
    ;;
  rt-occ)       # is rt_search_kernel bound by the matrix pipes a CU's workgroups share, or does every workgroup run at its own pace?
    {
    one "rt 1024, 2 trees per workgroup, two per CU" --workload c4
    one "rt 1024, 2 trees per workgroup, ONE per CU (LDS padded)" --workload c4 --tuning rt_lds_pad_kb=40
    one "rt 512, 2 trees per workgroup (256 workgroups)" --workload c4 --trees 512 --tuning rt_trees=2
    one "rt 256, 2 trees per workgroup (128 workgroups)" --workload c4 --trees 256 --tuning rt_trees=2
    one "rt 512, 1 tree per workgroup (512 workgroups)" --workload c4 --trees 512 --tuning rt_trees=1
    one "rt 256, 1 tree per workgroup (256 workgroups)" --workload c4 --trees 256 --tuning rt_trees=1
    one "rt 1024 K loops only, ONE per CU" --workload c4 --tuning rt_dbg=30,rt_lds_pad_kb=40
    one "rz 512 (LDS-resident whole-search kernel)" --workload c4-ws --trees 512
    one "rz 256" --workload c4-ws --trees 256
    } > $OUT/rt_occ.txt 2>&1
    cat $OUT/rt_occ.txt
    rocprofv3 -L > $OUT/counters.txt 2>&1
    CMD="python bench.py --workload c4 --steps 2 --warmup 1 $Q"
    timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_a -o run -- $CMD > $OUT/pmc_a.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_b -o run -- $CMD > $OUT/pmc_b.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH --output-format csv -d $OUT/pmc_c -o run -- $CMD > $OUT/pmc_c.log 2>&1
    python muzero-general_amd/tools/rocprof_summary.py $OUT rt_search > $OUT/summary.txt 2>&1
    tail -40 $OUT/summary.txt
    ;;
  rt-stagger)
    {
    for us in 0 20 50 100 150 200 300 500; do
      one "rt 1024 stagger $us us" --workload c4 --tuning rt_stagger_us=$us
    done
    for us in 50 100 200; do
      one "rt 1024 K loops only, stagger $us" --workload c4 --tuning rt_dbg=30,rt_stagger_us=$us
      one "rt 1536 stagger $us us" --workload c4 --trees 1536 --tuning rt_stagger_us=$us
      one "rt 512 stagger $us us" --workload c4 --trees 512 --tuning rt_stagger_us=$us
      one "rt 2048 stagger $us us" --workload c4 --trees 2048 --tuning rt_stagger_us=$us
    done
    } > $OUT/rt_stagger.txt 2>&1
    cat $OUT/rt_stagger.txt
    timeout 600 python -m pytest tests/test_gpu_tower_search.py -m gpu -q -s -x -k "bit_identical or routing" > $OUT/pytest_rt.log 2>&1
    grep -E "passed|failed|^FAILED|^ERROR|Error" $OUT/pytest_rt.log | tail
    ;;
  rt-exp)      # where a simulation of rt_search_kernel goes (knock-outs; results are wrong, timings are not) and the ring K loop
    {
    for t in 1024 512 1536; do
      one "rt $t" --workload c4 --trees $t
      one "rt $t ring" --workload c4 --trees $t --tuning rt_ring=1
    done
    one "rt 1024 no K loops" --workload c4 --tuning rt_dbg=1
    one "rt 1024 no K loops, no epilogues" --workload c4 --tuning rt_dbg=3
    one "rt 1024 no epilogues" --workload c4 --tuning rt_dbg=2
    one "rt 1024 no tree phases" --workload c4 --tuning rt_dbg=4
    one "rt 1024 no staging / tails" --workload c4 --tuning rt_dbg=8
    one "rt 1024 no heads" --workload c4 --tuning rt_dbg=16
    one "rt 1024 K loops only" --workload c4 --tuning rt_dbg=30
    one "rt 1024 K loops only, ring" --workload c4 --tuning rt_dbg=30,rt_ring=1
    one "rt 1024 nothing" --workload c4 --tuning rt_dbg=31
    one "rt 1024 one tree per workgroup" --workload c4 --tuning rt_trees=1
    one "rt 1024 one tree per workgroup, ring" --workload c4 --tuning rt_trees=1,rt_ring=1
    one "rt 1024 three trees per workgroup" --workload c4 --tuning rt_trees=3
    one "rt 1024 three trees per workgroup, ring" --workload c4 --tuning rt_trees=3,rt_ring=1
    } > $OUT/rt_exp.txt 2>&1
    cat $OUT/rt_exp.txt
    timeout 600 python -m pytest tests/test_gpu_tower_search.py -m gpu -q -s -x -k "bit_identical or routing" > $OUT/pytest_rt.log 2>&1
    grep -E "passed|failed|^FAILED|^ERROR|Error" $OUT/pytest_rt.log | tail
    ;;
  rt-first)
    timeout 900 python -m pytest tests/test_gpu_tower_search.py -m gpu -q -s -x > $OUT/pytest_rt.log 2>&1
    echo "pytest rc $?" >> $OUT/pytest_rt.log
    grep -E "passed|failed|^FAILED|^ERROR|Error|rc |rt_search_kernel with|identical" $OUT/pytest_rt.log | tail -40
    c4_shards 1024 512 1536 2048 3072 9216 > $OUT/c4_by_shard.txt 2>&1
    cat $OUT/c4_by_shard.txt
    ;;
  c4-shards) shift; shift; c4_shards ${@:-512 768 1024 1536 2048 3072 4608 9216} > $OUT/c4_by_shard.txt 2>&1; cat $OUT/c4_by_shard.txt ;;
  move)      # round 5: the self-play move behind two library calls -- device tests, then the self-play legs of the bench line
    timeout ${MOVE_TEST_S:-300} python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "${MOVE_TESTS:-refilled or batched_shard_of_4096 or whole_game or pipelined_shard}" > $OUT/pytest_move.log 2>&1
    echo "pytest rc $?" >> $OUT/pytest_move.log
    grep -E "passed|failed|^FAILED|^ERROR|Error|rc " $OUT/pytest_move.log | tail -12
    timeout 400 python bench.py ${MOVE_BENCH_ARGS---also c4 --cpu-seconds 0} > $OUT/bench_selfplay.log 2> $OUT/bench_selfplay.err
    echo "bench rc $?" >> $OUT/bench_selfplay.err
    tail -3 $OUT/bench_selfplay.err
    python -c "
import json
d = json.loads(open('$OUT/bench_selfplay.log').read().strip().splitlines()[-1])
print('C2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
for k in d:
    if k.startswith('selfplay'): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in d[k].items() if a in ('steps_per_sec', 'search_share', 'slot_groups', 'one_group_steps_per_sec', 'separate_calls_steps_per_sec', 'steps_per_sec_with_all_histories_as_lists', 'lockstep_steps_per_sec')})
"
    if [ -z "$MOVE_SKIP_PROFILE" ]; then
      timeout 300 python muzero-general_amd/tools/selfplay_host_profile.py > $OUT/host_profile_batched.txt 2>&1
      head -40 $OUT/host_profile_batched.txt
    fi
    ;;
  last)      # HEAD's evidence in one short call: the default bench line, then smoke + the whole -m gpu suite
    timeout 300 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
    echo "bench rc $?" >> $OUT/bench_default.err
    wc -c $OUT/bench_default.log
    run_tests
    ;;
  tests) run_tests ;;
  bench) run_bench ;;
  pmc) shift; shift; run_pmc ${@:-c2 c2-ckpt c3 c4 c4-ws c4-large c5 c5-512 gomoku atari} ;;
  final)      # tests + bench + pmc; the PMC passes stop starting workloads FINAL_BUDGET_S (default 1500) after the job began
    PMC_DEADLINE=$(( $(date +%s) + ${FINAL_BUDGET_S:-1500} ))
    run_tests; run_bench; run_pmc c4 atari c2 c3 c5 c4-ws c2-ckpt c5-512 c4-large gomoku ;;
  *) echo "unknown job $JOB"; exit 2 ;;
esac
find $OUT -size +4M -delete
