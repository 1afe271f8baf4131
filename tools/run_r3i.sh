#!/bin/bash
# Round 3, GPU call I: do more lanes lose because HIP streams share hardware queues?  (GPU_MAX_HW_QUEUES sweep)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; envs=$1; shift; env $envs timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1; echo "$name: $(grep -o '"value": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.log | head -1)"; }
run l3_q4 "A=1"
run l3_q8 "GPU_MAX_HW_QUEUES=8"
run l4_q4 "A=1" --lanes 4 --gas 8
run l4_q8 "GPU_MAX_HW_QUEUES=8" --lanes 4 --gas 8
run l6_q8 "GPU_MAX_HW_QUEUES=8" --lanes 6 --gas 6
run l6_q16 "GPU_MAX_HW_QUEUES=16" --lanes 6 --gas 6
run l2_q4 "A=1" --lanes 2 --gas 6
du -sh $O; date; echo done
