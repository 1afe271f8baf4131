#!/bin/bash
# Round 3, GPU call B: full GPU suite (LN fused backward, new GEMM tiles, flat gradient arenas, tightened parity bounds), GEMM ledger of the 4-wave tiles,
# driver bench, pp = 2 on one shared GPU with / without graph packet capture, ATen census, rocprofv3 kernel stats.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
echo "== parity tests"; date
timeout 600 python -m pytest tests/test_gpu_attention_long.py tests/test_gpu_fullsize.py tests/test_gpu_realdims.py -q -m gpu -s -p no:cacheprovider -k "not twelve" > $O/tests_parity.txt 2>&1
tail -3 $O/tests_parity.txt
echo "== rest of the GPU suite"; date
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_attention_long.py --deselect tests/test_gpu_realdims.py -k "not golden" > $O/tests_rest.txt 2>&1
tail -8 $O/tests_rest.txt
echo "== GEMM ledger: 4-wave tiles"; date
timeout 400 python tools/gemm_desc_timing.py profiles/r2_gemm_trace_sdxl_step.json $O/gemm_desc_w4.jsonl --min-gflop=3 --no-torch --hints=auto,11001,11002,12001,12002,13001,13003,14001,14002 > $O/gemm_desc_w4.log 2>&1
tail -1 $O/gemm_desc_w4.log
timeout 300 python tools/gemm_desc_timing.py profiles/r2_gemm_trace_sdxl_step.json $O/gemm_desc_auto.jsonl --no-torch --hints=auto > $O/gemm_desc_auto.log 2>&1
tail -1 $O/gemm_desc_auto.log
echo "== bench (driver command)"; date
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json.log 2> $O/bench.err
python -c "
import json
for l in open('$O/bench.json.log'):
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print({k:d[k] for k in ('value','ms_per_step','mfu_vs_bf16_mfma_peak')}, {k:d['roofline'][k] for k in ('achieved','frac','avg_launch_us','gemm_gpu_ms_per_step')}, d.get('parity'))
"
DPIPE_LNMOD_FUSE=0 timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_lnunfused.json.log 2> $O/bench_lnunfused.err
grep -o '"value": [0-9.]*' $O/bench_lnunfused.json.log | head -1
echo "== pp=2 on one shared GPU"; date
PORT=29571
DPIPE_BENCH_STALL_S=60 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 2 --steps 6 --warmup 2 --test-single-device --no-cpu-baseline > $O/bench_pp2_nodecapture.log 2>&1
grep -o '"value": [0-9.]*' $O/bench_pp2_nodecapture.log | head -1; tail -2 $O/bench_pp2_nodecapture.log | cut -c1-300
DPIPE_PP_PACKET_CAPTURE=1 DPIPE_BENCH_STALL_S=60 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((PORT+1)) bench.py --gpus 2 --steps 12 --warmup 2 --test-single-device --no-cpu-baseline > $O/bench_pp2_packetcapture.log 2>&1
grep -o '"value": [0-9.]*' $O/bench_pp2_packetcapture.log | head -1; tail -2 $O/bench_pp2_packetcapture.log | cut -c1-300
DPIPE_BENCH_STALL_S=60 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((PORT+2)) bench.py --gpus 2 --pp 1 --steps 6 --warmup 2 --test-single-device --no-cpu-baseline > $O/bench_dp2.log 2>&1
grep -o '"value": [0-9.]*' $O/bench_dp2.log | head -1; tail -2 $O/bench_dp2.log | cut -c1-300
echo "== ATen census"; date
timeout 300 python tools/aten_census.py > $O/aten_census.jsonl 2> $O/aten_census.err
tail -1 $O/aten_census.jsonl
echo "== rocprofv3 kernel stats"; date
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > $O/prof_bench.log 2>&1
ls $O/prof | head; find $O/prof -name "*kernel_stats.csv" | head -2
date; echo done
