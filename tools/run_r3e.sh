#!/bin/bash
# Round 3, GPU call E: batched context K/V projection, multi-tensor 8-bit AdamW, two forward streams per pipeline stage; ring-depth and GAS sweeps.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
echo "== tests"; date
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sdxl.py tests/test_gpu_optim.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_lora.py -q -m gpu -p no:cacheprovider > $O/tests.txt 2>&1
tail -6 $O/tests.txt | cut -c1-300
echo "== optimizer timing"; date
timeout 300 python tools/optim_timing.py 2>/dev/null | tee $O/optim_timing.jsonl
echo "== bench variants"; date
run() { name=$1; shift; envs=$1; shift; env $envs timeout 300 python bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1; echo "$name: $(grep -o '"value": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"avg_launch_us": [0-9.]*' $O/bench_$name.log | head -1) $(grep -o '"launches_per_step": [0-9]*' $O/bench_$name.log | head -1)"; }
run default "A=1"
run nokvbatch "DPIPE_BATCH_CONTEXT_KV=0"
run shallow128 "DPIPE_GEMM_SHALLOW=2"
run shallow64 "DPIPE_GEMM_SHALLOW=3"
run gas9 "A=1" --gas 9
run gas12 "A=1" --gas 12
run gas12l4 "A=1" --gas 12 --lanes 4
echo "== pp=2 on one shared GPU"; date
PORT=29581
for n in 2 1; do
DPIPE_STAGE_FWD_STREAMS=$n DPIPE_BENCH_STALL_S=60 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((PORT+n)) bench.py --gpus 2 --steps 6 --warmup 2 --test-single-device --no-cpu-baseline > $O/bench_pp2_fwd$n.log 2>&1
echo "pp2 fwd streams $n: $(grep -o '"value": [0-9.]*' $O/bench_pp2_fwd$n.log | head -1)"
done
du -sh $O; date; echo done
