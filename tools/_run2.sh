mkdir -p gpurun_out; rm -f gpurun_out/attn_*.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k attention 2>&1 | tail -5 >> gpurun_out/attn_tests.txt
for d in 1 0; do echo "=== SPLIT=$d" >> gpurun_out/attn_timing.txt; DPIPE_ATTN_DKV_SPLIT=$d timeout 400 python tools/kernel_timing.py attn >> gpurun_out/attn_timing.txt 2>&1; done
cat gpurun_out/attn_tests.txt
