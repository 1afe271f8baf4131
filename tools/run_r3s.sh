#!/bin/bash
# Round 3, GPU call S: after the 8-slice rule for very long K on the 64^2 tile: GEMM / SDXL / kernel tests and the driver's command again.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3s; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_gemm_pipe.py tests/test_gpu_sdxl.py tests/test_gpu_kernels.py tests/test_gpu_lora.py -q -m gpu -p no:cacheprovider > $O/tests.txt 2>&1
tail -3 $O/tests.txt | cut -c1-300
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2> $O/bench_driver.err
grep '^{"metric"' $O/bench_driver.log | cut -c1-1200
du -sh $O; date; echo done
