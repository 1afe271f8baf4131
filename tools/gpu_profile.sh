#!/bin/bash
# On the GPU box: kernel trace of the default bench + FETCH_SIZE / WRITE_SIZE passes over the step's GEMM launch list.  Outputs -> gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$1" != "pmc-only" ]; then
  rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_bench.log 2>&1
  f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
  find $O/prof -name '*kernel_trace.csv' -delete; find $O/prof -name '*.db' -delete
fi
T=$R/profiles/r2_gemm_trace_sdxl_step.json
if [ -f $T ] && [ "$1" != "trace-only" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -f csv -d $O/pmc_$c -o pmc -- python $R/tools/gemm_replay.py $T 6 > $O/pmc_$c.log 2>&1
    python $R/tools/pmc_agg.py $O/pmc_$c $O/pmc_${c}_agg.csv >> $O/pmc_$c.log 2>&1
  done
fi
