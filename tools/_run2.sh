mkdir -p gpurun_out; rm -f gpurun_out/attn_*.txt
DPIPE_ATTN_FWD_DMA=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | tail -15 > gpurun_out/attn_tests.txt
for d in 0 1; do echo "=== DMA=$d" >> gpurun_out/attn_timing.txt; DPIPE_ATTN_FWD_DMA=$d timeout 400 python tools/kernel_timing.py attn >> gpurun_out/attn_timing.txt 2>&1; done
cat gpurun_out/attn_tests.txt
