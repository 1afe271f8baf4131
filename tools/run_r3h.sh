#!/bin/bash
# Round 3, GPU call H: shallow-ring auto policy + concurrent-lanes roofline in the bench line; per-step timeline of the lanes.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
echo "== bench default"; date
DPIPE_STEP_TIMELINE=$O/step_timeline.json timeout 300 python bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_default.log 2>&1
grep -o '"value": [0-9.]*' $O/bench_default.log | head -1; grep -o '"concurrent_lanes": {[^}]*}' $O/bench_default.log; grep -o '"frac": [0-9.]*' $O/bench_default.log | head -1
python - <<'PY'
import json
rows=json.load(open('gpurun_out/r3h/step_timeline.json'))
for r in rows[-2:]:
    print('step',r['step'],'total gpu ms',r['gpu_ms_total'])
    for e in r['events']:
        print('   ',e['label'],e['gpu_ms'],e['host_ms'])
PY
echo "== deep rings"; date
DPIPE_GEMM_SHALLOW=0 timeout 300 python bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_deep.log 2>&1
grep -o '"value": [0-9.]*' $O/bench_deep.log | head -1; grep -o '"concurrent_lanes": {[^}]*}' $O/bench_deep.log; grep -o '"frac": [0-9.]*' $O/bench_deep.log | head -1
du -sh $O; date; echo done
