R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
# (1) clocks / power while the bench runs
( while true; do rocm-smi --showclocks --showpower --showuse --json 2>/dev/null | head -c 2000; echo; sleep 1; done ) > $O/r5d_smi_normal.log &
SMI=$!
tools/run_gpu.sh r5d "bench:normal::--steps 30 --warmup 5 --no-cpu-baseline --no-synced-loop"
kill $SMI
( while true; do rocm-smi --showclocks --showpower --showuse --json 2>/dev/null | head -c 2000; echo; sleep 1; done ) > $O/r5d_smi_zero.log &
SMI=$!
tools/run_gpu.sh r5d "bench:zero:DPIPE_BENCH_ZERO_WEIGHTS=1:--steps 30 --warmup 5 --no-cpu-baseline --no-synced-loop"
kill $SMI
# (2) concurrent kernel trace (no --stats): real start / end stamps of the lanes' kernels
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace -f csv -d $O/r5d_trace -o tr -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-synced-loop --light > $O/r5d_trace.log 2>&1 )
f=$(find $O/r5d_trace -name '*kernel_trace.csv' | head -1); ls -la $f
python tools/trace_overlap.py $f $O/r5d_trace_overlap.json --skip-frac 0.5 | tail -60
rm -rf $O/r5d_trace
# (3) the default driver command with everything on
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r5d_bench_driver.log 2>&1; echo "driver rc=$?"; grep '^{"metric"' $O/r5d_bench_driver.log > $O/r5d_bench_driver.json; tail -3 $O/r5d_bench_driver.log | cut -c1-600
