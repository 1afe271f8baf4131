R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_optim.py -x -q 2>&1 | tail -2
timeout 300 python tools/optim_timing.py 2>&1 | tail -4 | tee $O/r5f_optim_timing.jsonl
timeout 600 python bench.py --stack 4 --lanes 2 --steps 10 --warmup 3 --light --parity-samples 2 --parity-budget 30 --no-synced-loop > $O/r5f_stacked.log 2>&1; echo "stacked rc=$?"
grep '^{"metric"' $O/r5f_stacked.log > $O/r5f_stacked.json; python -c "
import json; d=json.loads(open('gpurun_out/r5f_stacked.json').read()); print('stacked', d['value'], d['parity']['grad_norm_rel_signed'], d['parity']['loss_rel_max'], d['peak_hbm_gb'])"
rocm-smi --showmeminfo vram | grep -i used
timeout 420 python bench.py --workload hv --steps 1 --warmup 1 --light > $O/r5f_hv.log 2>&1; echo "hv rc=$?"; grep '^{"metric"' $O/r5f_hv.log > $O/r5f_hv.json; cut -c1-400 $O/r5f_hv.json; tail -2 $O/r5f_hv.log | cut -c1-300
rocm-smi --showmeminfo vram | grep -i used; free -g | head -2
