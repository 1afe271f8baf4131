#!/bin/bash
# Round 3, GPU call T: is the 4-lane loss a hardware-queue artefact?  Lane 0 on the caller's stream (one queue fewer), 3 and 4 lanes.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3t; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; envs=$1; shift; env $envs timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1; echo "$name: $(grep -o '"value": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.log | head -1)"; }
run l3 "A=1"
run l3_main "DPIPE_LANE0_MAIN=1"
run l4 "A=1" --lanes 4 --gas 8
run l4_main "DPIPE_LANE0_MAIN=1" --lanes 4 --gas 8
run l4_main_q8 "DPIPE_LANE0_MAIN=1 GPU_MAX_HW_QUEUES=8" --lanes 4 --gas 8
du -sh $O; date; echo done
