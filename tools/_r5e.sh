R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
free -g > $O/r5e_free.txt; nproc >> $O/r5e_free.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs > $O/r5e_bench_headline.log 2>&1; echo "headline rc=$?"
grep '^{"metric"' $O/r5e_bench_headline.log > $O/r5e_bench_headline.json; tail -2 $O/r5e_bench_headline.log | cut -c1-300
free -g >> $O/r5e_free.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5e_bench_headline.json').read().strip().splitlines()[-1])
print('value',d['value'],'synced',d.get('value_synced_loop'),'frac',d['roofline']['frac'],'conc',d['roofline'].get('concurrent_lanes',{}).get('frac'))
p=d['parity']; print({k:p[k] for k in ('samples','grad_norm_rel_signed','grad_norm_rel_mean','grad_norm_rel_sigma','grad_norm_rel_max','loss_rel_max')}); print(p.get('fp32_kernels'))
print(d['cpu_baseline']['sample_seconds'], d['cpu_baseline']['cores'])
PY
timeout 600 python -m pytest tests/test_gpu_optim.py -x -q 2>&1 | tail -3
timeout 300 python tools/optim_timing.py 2>&1 | tail -8
