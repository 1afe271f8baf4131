#!/bin/bash
# Round 3, GPU call U: four real lanes (lane 0 on the caller's stream): GAS, lane count and ring-depth sweep.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3u; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; envs=$1; shift; env DPIPE_LANE0_MAIN=1 $envs timeout 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1; echo "$name: $(grep -o '"value": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"frac": [0-9.]*' $O/bench_$name.log | head -2 | tr '\n' ' ')"; }
run l4g8 "A=1" --lanes 4 --gas 8
run l4g12 "A=1" --lanes 4 --gas 12
run l4g4 "A=1" --lanes 4 --gas 4
run l4g6 "A=1" --lanes 4 --gas 6
run l5g10_q8 "GPU_MAX_HW_QUEUES=8" --lanes 5 --gas 10
run l5g10 "A=1" --lanes 5 --gas 10
run l4g8_deep "DPIPE_GEMM_SHALLOW=0" --lanes 4 --gas 8
run l4g8_sh1 "DPIPE_GEMM_SHALLOW=1" --lanes 4 --gas 8
du -sh $O; date; echo done
