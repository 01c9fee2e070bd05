mkdir -p gpurun_out/r24
for N in 0 1 2 3 5; do MZX_RZ_PHASE_NAPS=$N timeout 300 python bench.py --workload c4 --steps 3 --warmup 1 --cpu-seconds 0 --selfplay-moves 0 > gpurun_out/r24/bench_c4_naps$N.log 2>&1; done
