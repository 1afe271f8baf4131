tools/run_gpu.sh r5c \
 "tests:tests/test_gpu_gemm_pipe.py tests/test_gpu_conv.py" \
 "cmd:DPIPE_GEMM_VS=0 DPIPE_CONV_VS=0 DPIPE_GEMM_GR=8 python tools/gemm_desc_timing.py profiles/r4_gemm_trace_sdxl_step.json gpurun_out/r5c_ledger_old.jsonl --hints=auto --no-torch | tail -1" \
 "cmd:DPIPE_GEMM_VS=0 DPIPE_CONV_VS=0 python tools/gemm_desc_timing.py profiles/r4_gemm_trace_sdxl_step.json gpurun_out/r5c_ledger_gr.jsonl --hints=auto --no-torch | tail -1" \
 "cmd:python tools/gemm_desc_timing.py profiles/r4_gemm_trace_sdxl_step.json gpurun_out/r5c_ledger_new.jsonl --hints=auto,12001,12002,12003 --no-torch | tail -1" \
 "bench:old:DPIPE_GEMM_VS=0,DPIPE_CONV_VS=0,DPIPE_GEMM_GR=8:--steps 20 --warmup 5 --no-cpu-baseline --no-synced-loop" \
 "bench:gr:DPIPE_GEMM_VS=0,DPIPE_CONV_VS=0:--steps 20 --warmup 5 --no-cpu-baseline --no-synced-loop" \
 "bench:vsgemm:DPIPE_CONV_VS=0:--steps 20 --warmup 5 --no-cpu-baseline --no-synced-loop" \
 "bench:all::--steps 20 --warmup 5 --no-cpu-baseline --no-synced-loop" \
 "bench:old2:DPIPE_GEMM_VS=0,DPIPE_CONV_VS=0,DPIPE_GEMM_GR=8:--steps 20 --warmup 5 --no-cpu-baseline --no-synced-loop" \
 "bench:all2::--steps 20 --warmup 5 --no-cpu-baseline --no-synced-loop" \
 "cmd:python tools/conv_timing.py 2>&1 | tail -30" \
 "cmd:DPIPE_CONV_VS=0 python tools/conv_timing.py 2>&1 | tail -30"
