#!/bin/bash
# Round 4: PMC of a connect4 recurrent_inference at 512 samples (the half-shard of BASELINE config C4 at 1024 trees):
# rb_tower_kernel<3,1> (two boards per workgroup, one workgroup per CU) and the grouped head launches.  Separate --pmc passes.
TAG=${1:-r04s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python muzero-general_amd/tools/streamed_bench.py connect4 512 --mode 3 --iters 20"
timeout 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o run -- $CMD > $OUT/rocprof_mfma.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $OUT/pmc_mem -o run -- $CMD > $OUT/rocprof_mem.log 2>&1
python muzero-general_amd/tools/rocprof_summary.py $OUT rb_ > $OUT/summary.txt 2>&1
rm -rf $OUT/pmc_mfma $OUT/pmc_mem
cat $OUT/summary.txt | cut -c1-170 | head -40
