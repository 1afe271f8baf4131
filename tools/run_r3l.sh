#!/bin/bash
# Round 3, GPU call L: CLIP-L on a forked stream next to CLIP-G (parallel graph branch inside a lane): parity + A/B.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3l; mkdir -p $O
export TMPDIR=/tmp
echo "== tests"; date
DPIPE_PARALLEL_TEXT_ENCODERS=1 timeout 900 python -m pytest tests/test_gpu_sdxl.py -q -m gpu -p no:cacheprovider -x > $O/tests.txt 2>&1
tail -4 $O/tests.txt | cut -c1-300
echo "== bench"; date
run() { name=$1; shift; envs=$1; shift; env $envs timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 "$@" > $O/bench_$name.log 2>&1; echo "$name: $(grep -o '"value": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"loss_rel": [0-9.e-]*' $O/bench_$name.log | head -1) $(grep -o '"grad_norm_rel": [0-9.e-]*' $O/bench_$name.log | head -1)"; }
run default "A=1" --no-cpu-baseline
run pte "DPIPE_PARALLEL_TEXT_ENCODERS=1"
run default2 "A=1" --no-cpu-baseline
run pte2 "DPIPE_PARALLEL_TEXT_ENCODERS=1" --no-cpu-baseline
du -sh $O; date; echo done
