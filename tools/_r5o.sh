cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 400 python tools/conv_timing.py > gpurun_out/r5o_conv_timing.jsonl 2>gpurun_out/r5o_conv_timing.err; echo "conv rc=$?"; python - <<'PY'
import json
for l in open('gpurun_out/r5o_conv_timing.jsonl'):
    d=json.loads(l); print(d['hw'],d['cin'],d['cout'],'k',d['k'],'s',d['stride'],'u',d['ups'],'cnt',d['cnt'],'| auto',d['auto'],'r2',d['t128r2'],'v',d['t128v'],'vk2',d['t128vk2'],'r2k2',d['t128r2k2'],'torch',d['torch_fwd'], d.get('errors',''))
PY
timeout 400 python tools/gemm_mem_sensitivity.py profiles/r4_gemm_trace_sdxl_step.json gpurun_out/r5o_gemm_mem_sensitivity.json 2>&1 | tail -4
tools/run_gpu.sh r5o "bench:pp2l1::--gpus 2 --test-single-device --steps 6 --warmup 2 --no-cpu-baseline --no-synced-loop" \
  "bench:pp2l2:DPIPE_PIPE_LANES=2:--gpus 2 --test-single-device --steps 6 --warmup 2 --no-cpu-baseline --no-synced-loop"
