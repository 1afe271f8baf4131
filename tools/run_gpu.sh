#!/bin/bash
# One parameterised driver for the builder's gpurun calls (replaces the per-experiment tools/run_r3*.sh of round 3).
#   tools/run_gpu.sh <tag> <step> [<step> ...]      outputs -> gpurun_out/<tag>_*
# steps:  tests[:<pytest args>]   bench[:<name>[:<env assignments,comma separated>[:<bench args>]]]   prof[:<name>]   cmd:<shell command>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
TAG=$1; shift
cd $R
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" == "$step" ] && rest=""
  case $kind in
    tests)
      timeout 1500 python -m pytest ${rest:-tests -m gpu} -x -q > $O/${TAG}_tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/${TAG}_tests.log)";;
    bench)
      IFS=':' read -r name envs args <<< "$rest"; name=${name:-default}
      ( for kv in ${envs//,/ }; do export $kv; done; timeout 900 python bench.py ${args:---steps 8 --warmup 3 --no-cpu-baseline} > $O/${TAG}_bench_${name}.log 2>&1 )
      grep '^{"metric"' $O/${TAG}_bench_${name}.log > $O/${TAG}_bench_${name}.json
      python - "$O/${TAG}_bench_${name}.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get('roofline', {}); c = r.get('concurrent_lanes') or {}
    print(f"bench {sys.argv[2]}: {d['value']} {d['unit']}, {d['ms_per_step']} ms/step, frac {r.get('frac')}, launches/step {r.get('launches_per_step')}, avg us {r.get('avg_launch_us')}, concurrent frac {c.get('frac')}, loss {d.get('loss')}, parity {d.get('parity', {}).get('grad_norm_rel')}")
    if d.get('loss') != d.get('loss'): print(f"bench {sys.argv[2]}: *** NON-FINITE LOSS: the throughput above is not a measurement ***")
except Exception as e:
    print(f"bench {sys.argv[2]}: no JSON line ({e})")
PY
      ;;
    prof)
      name=${rest:-bench}
      ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/${TAG}_prof -o $name -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_prof_${name}.log 2>&1 )
      f=$(find $O/${TAG}_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${TAG}_${name}_kernel_stats.csv
      rm -rf $O/${TAG}_prof; grep '^{"metric"' $O/${TAG}_prof_${name}.log > $O/${TAG}_prof_${name}.json; echo "prof done: $(wc -l < $O/${TAG}_${name}_kernel_stats.csv) kernel rows";;
    pmc)      # pmc:<trace.json>:<div>  -- FETCH_SIZE / WRITE_SIZE passes (separate, counters only) over the step's GEMM launch list under the engine's lane policy
      IFS=':' read -r trace div <<< "$rest"
      for c in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && export TMPDIR=/tmp DPIPE_GEMM_SHALLOW=2 DPIPE_GEMM_BIG_TILES=16 && timeout 600 rocprofv3 --pmc $c -f csv -d $O/${TAG}_pmc_$c -o pmc -- python $R/tools/gemm_replay.py $R/$trace $div > $O/${TAG}_pmc_$c.log 2>&1 )
        python tools/pmc_agg.py $O/${TAG}_pmc_$c $O/${TAG}_pmc_gemm_step_$c.csv >> $O/${TAG}_pmc_$c.log 2>&1; rm -rf $O/${TAG}_pmc_$c
      done
      python tools/pmc_traffic_json.py $O/${TAG}_pmc_gemm_step_FETCH_SIZE.csv $O/${TAG}_pmc_gemm_step_WRITE_SIZE.csv $trace $div $O/${TAG}_pmc_gemm_traffic.json | tail -1;;
    sqpmc)    # sqpmc:<name>:<command...>  -- ONE counters-only pass (8 SQ slots + GRBM) over an eager command, aggregated per kernel name
      IFS=':' read -r name pcmd <<< "$rest"
      ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -f csv -d $O/${TAG}_sq_$name -o pmc -- bash -c "cd $R && $pcmd" > $O/${TAG}_sq_$name.log 2>&1 )
      python tools/pmc_agg.py $O/${TAG}_sq_$name $O/${TAG}_pmc_${name}_sq_counters.csv | tail -1; rm -rf $O/${TAG}_sq_$name;;
    census)   # census[:<bench args>]  -- kernel trace (no counters) of the lane bench + tools/trace_overlap.py
      ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace -f csv -d $O/${TAG}_tr -o tr -- python $R/bench.py ${rest:---steps 4 --warmup 2 --no-cpu-baseline --no-synced-loop --light} > $O/${TAG}_census.log 2>&1 )
      python tools/trace_overlap.py $(find $O/${TAG}_tr -name '*kernel_trace.csv' | head -1) $O/${TAG}_overlap.json --skip-frac 0.5 | head -30; rm -rf $O/${TAG}_tr;;
    ablate)   # ablate:<mask,mask,...>[:kdiv]  -- DEBUG sensitivity census: the light four-lane bench with one kernel class's launches SKIPPED per run (results are garbage, only the
              # clock counts; csrc/runtime.hip: 1 GEMM, 2 convolution, 4 attention, 8 LayerNorm, 16 GroupNorm, 32 element-wise, 64 step end), and, with :kdiv, one run with every GEMM's K loop d-fold shorter
      IFS=':' read -r masks kdiv <<< "$rest"
      for m in ${masks//,/ }; do
        ( export DPIPE_DEBUG_ABLATE=$m; timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-synced-loop --light > $O/${TAG}_ablate_$m.log 2>&1 )
        python - "$O/${TAG}_ablate_$m.log" "$m" >> $O/${TAG}_ablation.jsonl <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')]
d = json.loads(line[-1]) if line else {}
print(json.dumps({'ablate_mask': int(sys.argv[2]), 'gemm_kdiv': 1, 'ms_per_step': d.get('ms_per_step'), 'images_per_s': d.get('value')}))
PY
        tail -1 $O/${TAG}_ablation.jsonl
      done
      if [ -n "$kdiv" ]; then
        ( export DPIPE_DEBUG_GEMM_KDIV=$kdiv; timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-synced-loop --light > $O/${TAG}_ablate_kdiv$kdiv.log 2>&1 )
        python - "$O/${TAG}_ablate_kdiv$kdiv.log" "$kdiv" >> $O/${TAG}_ablation.jsonl <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')]
d = json.loads(line[-1]) if line else {}
print(json.dumps({'ablate_mask': 0, 'gemm_kdiv': int(sys.argv[2]), 'ms_per_step': d.get('ms_per_step'), 'images_per_s': d.get('value')}))
PY
        tail -1 $O/${TAG}_ablation.jsonl
      fi;;
    cmd)      # every cmd step gets its own log: <tag>_cmd1.log, <tag>_cmd2.log, ...
      NCMD=$((${NCMD:-0} + 1))
      bash -c "$rest" > $O/${TAG}_cmd${NCMD}.log 2>&1; echo "cmd${NCMD} rc=$? $(tail -2 $O/${TAG}_cmd${NCMD}.log)";;
  esac
done
