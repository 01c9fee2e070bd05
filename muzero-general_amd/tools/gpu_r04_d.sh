#!/bin/bash
# Round 4: knock-out timings of the tower kernel (where does a layer's time go?) + PMC of the tower launches.
TAG=${1:-r04d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
SB="python muzero-general_amd/tools/streamed_bench.py"
{
for dbg in 0 1 2 3; do
  echo "== MZX_RB_DBG=$dbg"
  MZX_RB_DBG=$dbg $SB connect4 4608 --mode 3 --iters 10
  MZX_RB_DBG=$dbg MZX_RB_TOWER_T=6 $SB connect4 4608 --mode 3 --iters 10
  MZX_RB_DBG=$dbg $SB gomoku 512 --mode 1 --iters 5
done
} > $OUT/knockout.log 2>&1
CMD="$SB connect4 4608 --mode 3 --iters 5"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4/stats -o run -- $CMD > $OUT/rocprof_stats.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/c4/pmc_mfma -o run -- $CMD > $OUT/rocprof_mfma.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/c4/pmc_lds -o run -- $CMD > $OUT/rocprof_lds.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c4/pmc_fetch -o run -- $CMD > $OUT/rocprof_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c4/pmc_write -o run -- $CMD > $OUT/rocprof_write.log 2>&1
python muzero-general_amd/tools/rocprof_summary.py $OUT/c4 rb_ > $OUT/summary_c4.txt 2>&1
find $OUT -size +4M -delete
grep -v amdgpu.ids $OUT/knockout.log
