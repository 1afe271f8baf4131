#!/bin/bash
# Round 3, GPU call V: new defaults (4 lanes with lane 0 on the caller's stream, GAS 8; backward half of a pipeline stage on the caller's stream):
# full-size / SDXL / pipeline tests, the driver's command, pp = 2 on one shared GPU, rocprofv3 kernel stats.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3v; mkdir -p $O
export TMPDIR=/tmp
echo "== tests"; date
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sdxl.py tests/test_gpu_pipeline.py tests/test_gpu_flux.py tests/test_gpu_hv.py -q -m gpu -p no:cacheprovider -s > $O/tests.txt 2>&1
grep "timed path\|passed\|failed" $O/tests.txt | cut -c1-300 | tail -6
echo "== driver command"; date
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2> $O/bench_driver.err
grep '^{"metric"' $O/bench_driver.log | cut -c1-1500
echo "== A/B: previous default"; date
timeout 300 python3 bench.py --gpus 1 --steps 10 --warmup 3 --lanes 3 --gas 6 --no-cpu-baseline > $O/bench_l3g6.log 2>&1; grep -o '"value": [0-9.]*' $O/bench_l3g6.log | head -1
echo "== pp2 shared GPU"; date
for n in 1 2; do
DPIPE_STAGE_FWD_STREAMS=$n DPIPE_BENCH_STALL_S=60 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29681+n)) bench.py --gpus 2 --steps 6 --warmup 2 --test-single-device --no-cpu-baseline > $O/bench_pp2_fwd$n.log 2>&1
echo "pp2 fwd streams $n: $(grep -o '"value": [0-9.]*' $O/bench_pp2_fwd$n.log | head -1)"
done
echo "== rocprofv3 kernel stats"; date
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_bench.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
rm -rf $O/prof
head -4 $O/bench_kernel_stats.csv | cut -c1-200
du -sh $O; date; echo done
