#!/bin/bash
# Round 3, GPU call K: occupancy-style 4-wave half-step 128^2 tiles (T128Q3 / T128Q4): parity, per-descriptor ledger, in-step A/B.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
echo "== tests"; date
timeout 900 python -m pytest tests/test_gpu_gemm_pipe.py -q -m gpu -p no:cacheprovider -x -k "test_pipe_gemm_matches_fp32_matmul and (11000 or 12000)" > $O/tests.txt 2>&1
tail -4 $O/tests.txt | cut -c1-300
echo "== ledger"; date
timeout 600 python tools/gemm_desc_timing.py profiles/r2_gemm_trace_sdxl_step.json $O/ledger_q.jsonl --hints=auto,3001,4001,11001,11002,11003,12001,12002 --min-gflop=9 --no-torch 2>&1 | cut -c1-400 | tail -40
echo "== bench"; date
run() { name=$1; shift; envs=$1; shift; env $envs timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1; echo "$name: $(grep -o '"value": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"frac": [0-9.]*' $O/bench_$name.log | head -2 | tr '\n' ' ')"; }
run default "A=1"
run q3 "DPIPE_GEMM_SHALLOW=4"
run q4 "DPIPE_GEMM_SHALLOW=5"
du -sh $O; date; echo done
