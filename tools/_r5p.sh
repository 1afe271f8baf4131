cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-synced-loop --save-gemm-trace gpurun_out/r5_gemm_trace_sdxl_step.json > gpurun_out/r5p_trace_bench.log 2>&1; echo "trace rc=$?"; ls -la gpurun_out/r5_gemm_trace_sdxl_step.json
cp gpurun_out/r5_gemm_trace_sdxl_step.json profiles/r5_gemm_trace_sdxl_step.json
tools/run_gpu.sh r5p "pmc:profiles/r5_gemm_trace_sdxl_step.json:8"
