#!/bin/bash
# Round 3, GPU call M (final tree): full GPU test suite, smoke(), the driver's bench command (+ the step's GEMM launch list), rocprofv3 kernel stats of the
# same bench, FETCH_SIZE / WRITE_SIZE passes over the new launch list, SQ / LDS counter passes over the attention and convolution kernels.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3m; mkdir -p $O
export TMPDIR=/tmp
echo "== full GPU suite"; date
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/tests_full.txt 2>&1
tail -5 $O/tests_full.txt | cut -c1-300
echo "== smoke"; date
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt | cut -c1-300
echo "== driver command"; date
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --save-gemm-trace $O/gemm_trace_sdxl_step.json > $O/bench_driver.log 2> $O/bench_driver.err
grep '^{"metric"' $O/bench_driver.log | cut -c1-3000
echo "== rocprofv3 kernel stats"; date
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_bench.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
rm -rf $O/prof
head -12 $O/bench_kernel_stats.csv | cut -c1-220
T=$O/gemm_trace_sdxl_step.json
if [ -f $T ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -f csv -d $O/pmc_$c -o pmc -- python $R/tools/gemm_replay.py $T 6 > $O/pmc_$c.log 2>&1
  python $R/tools/pmc_agg.py $O/pmc_$c $O/pmc_${c}_agg.csv >> $O/pmc_$c.log 2>&1
  rm -rf $O/pmc_$c
done
cd $R; python tools/pmc_traffic_json.py $O/pmc_FETCH_SIZE_agg.csv $O/pmc_WRITE_SIZE_agg.csv $T 6 $O/pmc_gemm_traffic.json; cat $O/pmc_gemm_traffic.json | cut -c1-900; cd /tmp
fi
echo "== SQ counters: attention + convolution"; date
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -f csv -d $O/pmc_sq -o pmc -- python $R/tools/attn_conv_pmc_probe.py > $O/pmc_sq.log 2>&1
(cd $R; python tools/pmc_agg.py $O/pmc_sq $O/pmc_attn_conv_sq_counters.csv >> $O/pmc_sq.log 2>&1); rm -rf $O/pmc_sq
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $O/pmc_inst -o pmc -- python $R/tools/attn_conv_pmc_probe.py > $O/pmc_inst.log 2>&1
(cd $R; python tools/pmc_agg.py $O/pmc_inst $O/pmc_attn_conv_inst_counters.csv >> $O/pmc_inst.log 2>&1); rm -rf $O/pmc_inst
wc -l $O/pmc_attn_conv_*.csv; tail -2 $O/pmc_sq.log | cut -c1-300
du -sh $O; date; echo done
