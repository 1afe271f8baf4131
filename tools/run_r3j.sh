#!/bin/bash
# Round 3, GPU call J: micro-batch lanes on the DiT workloads (BASELINE configs 3 / 4).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3j; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; timeout 600 python bench.py --gpus 1 --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1; echo "$name: $(grep -o '"value": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"mfu_vs_bf16_mfma_peak": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"peak_hbm_gb": [0-9.]*' $O/bench_$name.log | head -1)"; tail -2 $O/bench_$name.log | cut -c1-300 | grep -v '^{"metric' ; }
run flux_l1 --workload flux --steps 6 --warmup 2
run flux_l2 --workload flux --steps 6 --warmup 2 --lanes 2
run flux_l2g4 --workload flux --steps 4 --warmup 2 --lanes 2 --gas 4
run wan_l2 --workload wan --steps 4 --warmup 2 --lanes 2
du -sh $O; date; echo done
