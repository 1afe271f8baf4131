#!/bin/bash
# Round 3, GPU call Q: the 8-wave 128^2 tile on 5- / 4-deep rings of half K-steps (T128H5 / T128H4): parity, ledger, in-step A/B.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3q; mkdir -p $O
export TMPDIR=/tmp
echo "== tests"; date
timeout 900 python -m pytest tests/test_gpu_gemm_pipe.py -q -m gpu -p no:cacheprovider -x -k "test_pipe_gemm_matches_fp32_matmul and (13000 or 14000)" > $O/tests.txt 2>&1
tail -3 $O/tests.txt | cut -c1-300
echo "== ledger"; date
timeout 600 python tools/gemm_desc_timing.py profiles/r3_gemm_trace_sdxl_step.json $O/ledger_h.jsonl --hints=auto,3001,4001,13001,13002,13003,14001,14002 --min-gflop=3 --no-torch 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j=json.loads(l)
    if 'us' in j: print(j['ta'],j['tb'],j['M'],j['N'],j['K'],'x',j['count'],j['us'])
    else: print(l.strip())
"
echo "== bench"; date
run() { name=$1; shift; envs=$1; shift; env $envs timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1; echo "$name: $(grep -o '"value": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"frac": [0-9.]*' $O/bench_$name.log | head -2 | tr '\n' ' ')"; }
run default "A=1"
run h5 "DPIPE_GEMM_SHALLOW=6"
run h4 "DPIPE_GEMM_SHALLOW=7"
run deep "DPIPE_GEMM_SHALLOW=0"
du -sh $O; date; echo done
