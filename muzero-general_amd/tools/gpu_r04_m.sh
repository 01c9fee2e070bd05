export TMPDIR=/tmp
SB="python muzero-general_amd/tools/streamed_bench.py"
for dbg in 0 1 2 3; do echo "== DBG=$dbg"; MZX_RB_DBG=$dbg $SB connect4 512 --mode 3 --iters 20; MZX_RB_DBG=$dbg MZX_RB_TOWER_T=3 $SB connect4 768 --mode 3 --iters 20; done 2>&1 | grep -v amdgpu
