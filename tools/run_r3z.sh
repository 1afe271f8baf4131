#!/bin/bash
# Round 3, GPU call Z (final tree): full GPU test suite, smoke(), the driver's bench command, a per-step timeline of the four lanes, rocprofv3 kernel stats.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3z; mkdir -p $O
export TMPDIR=/tmp
echo "== full GPU suite"; date
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/tests_full.txt 2>&1
grep -n "passed\|failed" $O/tests_full.txt | tail -3 | cut -c1-300
echo "== smoke"; date
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt | cut -c1-300
echo "== driver command"; date
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2> $O/bench_driver.err
grep '^{"metric"' $O/bench_driver.log | cut -c1-900
echo "== timeline"; date
DPIPE_STEP_TIMELINE=$O/step_timeline_4lanes.json timeout 300 python bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_timeline.log 2>&1
python - <<'PY'
import json
rows=json.load(open('gpurun_out/r3z/step_timeline_4lanes.json'))
r=rows[-1]
print('step',r['step'],'total gpu ms',r['gpu_ms_total'])
for e in r['events']:
    print('   ',e['label'],e['gpu_ms'],e['host_ms'])
PY
echo "== rocprofv3 kernel stats"; date
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_bench.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
rm -rf $O/prof
head -3 $O/bench_kernel_stats.csv | cut -c1-200
du -sh $O; date; echo done
