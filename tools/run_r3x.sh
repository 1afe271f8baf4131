#!/bin/bash
# Round 3, GPU call X: lane streams picked by a concurrency probe (engine.concurrent_streams) vs fresh streams, with 0..3 streams created before the engine.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3x; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; envs=$1; shift; env $envs timeout 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1; echo "$name: $(grep -o '"value": [0-9.]*' $O/bench_$name.log | head -1)"; }
for x in 0 1 2 3; do
run probe_x$x "DPIPE_BENCH_EXTRA_STREAMS=$x"
run noprobe_x$x "DPIPE_BENCH_EXTRA_STREAMS=$x DPIPE_LANE_STREAM_PROBE=0"
done
du -sh $O; date; echo done
