#!/bin/bash
# Round 3, GPU call D: LayerNorm backward row-per-wave policy A/B, shallow-ring lane-overlap experiment, regression tests of the touched kernels.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
echo "== LayerNorm A/B"; date
timeout 200 python tools/norm_timing.py 2>/dev/null | grep layer_norm > $O/norm_auto.jsonl
DPIPE_LNMOD_RW=4 timeout 200 python tools/norm_timing.py 2>/dev/null | grep layer_norm > $O/norm_rw4.jsonl
DPIPE_LNMOD_FUSE=0 timeout 200 python tools/norm_timing.py 2>/dev/null | grep layer_norm > $O/norm_unfused.jsonl
paste -d'|' $O/norm_auto.jsonl $O/norm_rw4.jsonl $O/norm_unfused.jsonl | sed 's/"op": "layer_norm", //g'
echo "== tests"; date
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm_pipe.py tests/test_gpu_fullsize.py tests/test_gpu_sdxl.py -q -m gpu -p no:cacheprovider > $O/tests.txt 2>&1
tail -4 $O/tests.txt
echo "== bench"; date
for cfg in "default:" "shallow3:DPIPE_GEMM_SHALLOW=1" "shallow4:DPIPE_GEMM_SHALLOW=1 LANES=4" "lnunfused:DPIPE_LNMOD_FUSE=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}; lanes=3
  case "$envs" in *LANES=4*) lanes=4; envs=${envs/LANES=4/};; esac
  env $envs timeout 300 python bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu-baseline --lanes $lanes $([ $lanes = 4 ] && echo --gas 8) > $O/bench_$name.log 2>&1
  echo "$name: $(grep -o '"value": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"avg_launch_us": [0-9.]*' $O/bench_$name.log | head -1)"
done
du -sh $O; date; echo done
