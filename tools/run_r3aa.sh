#!/bin/bash
# Round 3, GPU call AA: dispatch rule for the occupancy-style tile (short K, many tiles, K-contiguous A): GEMM / conv tests + A/B in the step.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3aa; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gemm_pipe.py tests/test_gpu_conv.py tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x > $O/tests.txt 2>&1
tail -2 $O/tests.txt | cut -c1-300
run() { name=$1; shift; envs=$1; shift; env $envs timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1; echo "$name: $(grep -o '"value": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"frac": [0-9.]*' $O/bench_$name.log | head -2 | tr '\n' ' ') $(grep -o '"avg_launch_us": [0-9.]*' $O/bench_$name.log | head -1)"; }
run q3on "A=1"
run q3off "DPIPE_GEMM_Q3=0"
run q3on2 "A=1"
run q3off2 "DPIPE_GEMM_Q3=0"
du -sh $O; date; echo done
