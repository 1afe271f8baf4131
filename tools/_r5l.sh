cd ${GRAFT_REPO_ROOT:-/root/repo}
tools/run_gpu.sh r5l "tests:tests/test_gpu_sdxl.py -k 'stacked'" \
 "bench:s2l4::--steps 20 --warmup 5 --no-cpu-baseline --no-synced-loop --stack 2 --lanes 4" \
 "bench:s4l2::--steps 20 --warmup 5 --no-cpu-baseline --no-synced-loop --stack 4 --lanes 2" \
 "bench:s8l1::--steps 20 --warmup 5 --no-cpu-baseline --no-synced-loop --stack 8 --lanes 1" \
 "bench:s2l2::--steps 20 --warmup 5 --no-cpu-baseline --no-synced-loop --stack 2 --lanes 2" \
 "bench:s4l1::--steps 20 --warmup 5 --no-cpu-baseline --no-synced-loop --stack 4 --lanes 1"
for n in s2l4 s4l2 s8l1 s2l2 s4l1; do python -c "
import json; d=json.loads(open('gpurun_out/r5l_bench_$n.json').read()); print('$n', d['value'], 'loss', d['loss'], 'gn', d['grad_norm'], 'hbm', d['peak_hbm_gb'])"; done
timeout 600 python bench.py --stack 4 --lanes 2 --steps 10 --warmup 3 --light --parity-samples 2 --parity-budget 30 --no-synced-loop > gpurun_out/r5l_stacked_child.log 2>&1; echo "stacked child rc=$?"
grep '^{"metric"' gpurun_out/r5l_stacked_child.log > gpurun_out/r5l_stacked_child.json; python -c "
import json; d=json.loads(open('gpurun_out/r5l_stacked_child.json').read()); p=d['parity']; print('stacked child', d['value'], d['loss'], p['grad_norm_rel_signed'], p['loss_rel_max'])"
