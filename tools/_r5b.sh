tools/run_gpu.sh r5b \
 "tests:tests/test_gpu_gemm_pipe.py -k 'test_pipe_gemm_matches_fp32_matmul'" \
 "cmd:python -m pytest tests/test_gpu_kernels.py -x -q -k 'attn or attention or stores_once' 2>&1 | tail -3" \
 "cmd:DPIPE_ATTN_RING=3 DPIPE_ATTN_RING_DKV=3 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_attention_long.py -x -q -k 'attn or attention' 2>&1 | tail -3" \
 "cmd:DPIPE_ATTN_QB2=1 DPIPE_ATTN_RING=3 python -m pytest tests/test_gpu_kernels.py -x -q -k 'attn or attention' 2>&1 | tail -3" \
 "cmd:DPIPE_ATTN_QB2=2 python -m pytest tests/test_gpu_kernels.py -x -q -k 'attn or attention' 2>&1 | tail -3" \
 "cmd:python tools/kernel_timing.py attn64" \
 "cmd:DPIPE_ATTN_RING=3 python tools/kernel_timing.py attn64" \
 "cmd:DPIPE_ATTN_RING=3 DPIPE_ATTN_RING_DKV=3 python tools/kernel_timing.py attn64" \
 "cmd:DPIPE_ATTN_QB2=1 python tools/kernel_timing.py attn64" \
 "cmd:DPIPE_ATTN_QB2=1 DPIPE_ATTN_RING=3 python tools/kernel_timing.py attn64" \
 "cmd:DPIPE_ATTN_QB2=2 DPIPE_ATTN_RING=3 python tools/kernel_timing.py attn64" \
 "cmd:python tools/gemm_desc_timing.py profiles/r4_gemm_trace_sdxl_step.json gpurun_out/r5b_gemm_ledger.jsonl --hints=auto,3001,4001,12001,13001,14001,15001,7001 --min-gflop=9 | tail -3"
