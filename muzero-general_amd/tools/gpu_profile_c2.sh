#!/bin/bash
# Runs on the GPU box (through gpurun): parity tests, bench line, rocprofv3 kernel stats and PMC passes
# for the C2 workload.  Outputs under gpurun_out/$TAG/ ; copy what should be judged into profiles/.
TAG=${1:-c2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 20 --warmup 3 --cpu-seconds 0"
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
python bench.py --steps 50 --warmup 5 --cpu-seconds 10 > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $BENCH > $OUT/rocprof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- $BENCH > $OUT/rocprof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- $BENCH > $OUT/rocprof_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -o run -- $BENCH > $OUT/rocprof_sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o run -- $BENCH > $OUT/rocprof_sq2.log 2>&1
find $OUT -name '*.csv' | head -50 > $OUT/files.txt
# keep what is merged back small: drop per-dispatch traces of the torch helper kernels beyond 2 MB
find $OUT -size +8M -delete
ls -laR $OUT >> $OUT/files.txt
