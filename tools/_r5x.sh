#!/bin/bash
# round-5 call x: the 2x2 block-sum kernel (dpipe_upsample2x_adjoint) on the box: its own test, the convolution tests that go through it, the SDXL step parity tests, one light bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 150 python -m pytest tests/test_gpu_conv.py -x -q > $O/r5x_conv_tests.log 2>&1; echo "conv tests rc=$? $(tail -1 $O/r5x_conv_tests.log)"
timeout 200 python -m pytest tests/test_gpu_sdxl.py -x -q -k "fp32_matches or bf16_close or lanes_match or stores_graphs" > $O/r5x_sdxl_tests.log 2>&1; echo "sdxl tests rc=$? $(tail -1 $O/r5x_sdxl_tests.log)"
timeout 200 python bench.py --steps 10 --warmup 3 --light --parity-samples 2 --parity-budget 30 --no-synced-loop > $O/r5x_bench.log 2>&1; echo "bench rc=$?"
grep '^{"metric"' $O/r5x_bench.log > $O/r5x_bench.json; python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r5x_bench.json').read().strip().splitlines()[-1])
    print(d['value'], d['unit'], d['ms_per_step'], 'loss', d.get('loss'), 'parity', json.dumps(d.get('parity'))[:600])
except Exception as e:
    print('no JSON', e)
PY
