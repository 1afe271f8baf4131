cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -k "per_sample or residual_extra" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_sdxl.py -x -q -k "stacked" 2>&1 | tail -3
tools/run_gpu.sh r5s "bench:s4l2::--steps 20 --warmup 5 --no-cpu-baseline --no-synced-loop --stack 4 --lanes 2" "bench:base::--steps 20 --warmup 5 --no-cpu-baseline --no-synced-loop"
timeout 600 python tools/gemm_mem_sensitivity.py profiles/r5_gemm_trace_sdxl_step.json gpurun_out/r5s_gemm_mem_sensitivity.json 2>&1 | tail -4
