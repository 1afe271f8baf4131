#!/bin/bash
# Round 3, GPU call C (small outputs only): LayerNorm backward A/B (fused partials vs separate column reduction), ledger of the 4-wave GEMM tiles on the big
# shapes, ATen census, parity prints of the full-size golden test, rocprofv3 kernel stats + PMC traffic passes (tools/gpu_profile.sh).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
echo "== LayerNorm A/B"; date
timeout 200 python tools/norm_timing.py > $O/norm_fused.jsonl 2>$O/norm.err
DPIPE_LNMOD_FUSE=0 timeout 200 python tools/norm_timing.py > $O/norm_unfused.jsonl 2>>$O/norm.err
paste -d'|' <(grep layer_norm $O/norm_fused.jsonl) <(grep layer_norm $O/norm_unfused.jsonl)
echo "== GEMM ledger: 4-wave tiles, big shapes"; date
timeout 400 python tools/gemm_desc_timing.py profiles/r2_gemm_trace_sdxl_step.json $O/gemm_desc_w4.jsonl --min-gflop=9 --hints=auto,3001,3003,4001,4002,5001,11001,11002,12001,12002,13001,13003,14001,14002 > $O/gemm_desc_w4.log 2>&1
tail -1 $O/gemm_desc_w4.log
timeout 300 python tools/gemm_desc_timing.py profiles/r2_gemm_trace_sdxl_step.json $O/gemm_desc_w4_small.jsonl --min-gflop=3 --no-torch --hints=auto,2001,3001,13001,14001 > $O/gemm_desc_w4_small.log 2>&1
tail -1 $O/gemm_desc_w4_small.log
echo "== ATen census"; date
timeout 300 python tools/aten_census.py > $O/aten_census.jsonl 2> $O/aten_census.err
tail -1 $O/aten_census.jsonl
echo "== parity prints"; date
timeout 400 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -p no:cacheprovider -k golden > $O/tests_fullsize.txt 2>&1
grep "per-parameter\|rel. error\|passed\|failed" $O/tests_fullsize.txt | cut -c1-1200
echo "== rocprofv3"; date
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_bench.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
rm -rf $O/prof
head -30 $O/bench_kernel_stats.csv | cut -c1-200
T=$R/profiles/r2_gemm_trace_sdxl_step.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -f csv -d $O/pmc_$c -o pmc -- python $R/tools/gemm_replay.py $T 6 > $O/pmc_$c.log 2>&1
  python $R/tools/pmc_agg.py $O/pmc_$c $O/pmc_${c}_agg.csv >> $O/pmc_$c.log 2>&1
  rm -rf $O/pmc_$c
  tail -3 $O/pmc_$c.log | cut -c1-300
done
du -sh $O; date; echo done
