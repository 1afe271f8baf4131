cd ${GRAFT_REPO_ROOT:-/root/repo}
S=$(date +%s)
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5m_bench_driver.log 2>&1; echo "driver rc=$? wall $(( $(date +%s) - S )) s"
grep '^{"metric"' gpurun_out/r5m_bench_driver.log > gpurun_out/r5m_bench_driver.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5m_bench_driver.json').read())
print('value',d['value'],'synced',d.get('value_synced_loop'),'frac',d['roofline']['frac'],'conc',d['roofline'].get('concurrent_lanes',{}).get('frac'))
p=d['parity']; print({k:p[k] for k in ('samples','grad_norm_rel_signed','grad_norm_rel_mean','grad_norm_rel_median','grad_norm_rel_max','loss_rel_max')}); print(p.get('fp32_kernels'))
for k,v in d['other_configs'].items(): print(k, v.get('value'), v.get('error'), v.get('wall_seconds'), (v.get('parity') or {}).get('grad_norm_rel_signed'))
PY
tail -3 gpurun_out/r5m_bench_driver.log | cut -c1-300
