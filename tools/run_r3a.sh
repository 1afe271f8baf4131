#!/bin/bash
# Round 3, GPU call A: parity tests (new fp32 conv / long attention / golden projections), full GPU suite, per-descriptor GEMM ledger (+ A/B libraries),
# the driver's bench command (with the parity object), BASELINE configs 3 / 4 / 5 as real steps.  Everything logs under gpurun_out/r3a/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
echo "== new parity tests"; date
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_attention_long.py tests/test_gpu_fullsize.py tests/test_gpu_realdims.py -q -m gpu -s -p no:cacheprovider --junitxml=$O/junit_parity.xml > $O/tests_parity.txt 2>&1
tail -5 $O/tests_parity.txt
echo "== rest of the GPU suite"; date
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_conv.py --deselect tests/test_gpu_attention_long.py --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_realdims.py > $O/tests_rest.txt 2>&1
tail -5 $O/tests_rest.txt
echo "== GEMM descriptor ledger"; date
timeout 600 python tools/gemm_desc_timing.py profiles/r2_gemm_trace_sdxl_step.json $O/gemm_desc.jsonl > $O/gemm_desc.log 2>&1
tail -1 $O/gemm_desc.log
DPIPE_GEMM_SKINNY=0 timeout 300 python tools/gemm_desc_timing.py profiles/r2_gemm_trace_sdxl_step.json $O/gemm_desc_noskinny.jsonl --hints=auto --no-torch > $O/gemm_desc_noskinny.log 2>&1
tail -1 $O/gemm_desc_noskinny.log
DPIPE_GEMM_SKINNY=0 DPIPE_HIP_LIB=$PWD/diffusion_pipe_amd/csrc/_build/libdpipe_hip_slabplain.so timeout 300 python tools/gemm_desc_timing.py profiles/r2_gemm_trace_sdxl_step.json $O/gemm_desc_slabplain.jsonl --hints=auto --no-torch > $O/gemm_desc_slabplain.log 2>&1
tail -1 $O/gemm_desc_slabplain.log
echo "== bench (driver command)"; date
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json.log 2> $O/bench.err
tail -c 1500 $O/bench.json.log
echo "== DiT workloads, tiny configs"; date
for w in flux wan hv; do timeout 300 python bench.py --workload $w --config tiny --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_${w}_tiny.log 2>&1; tail -c 400 $O/bench_${w}_tiny.log; echo; done
echo "== DiT workloads, full size"; date
timeout 600 python bench.py --workload flux --steps 3 --warmup 1 > $O/bench_flux.log 2>&1; tail -c 600 $O/bench_flux.log; echo
timeout 600 python bench.py --workload wan --steps 2 --warmup 1 > $O/bench_wan.log 2>&1; tail -c 600 $O/bench_wan.log; echo
timeout 900 python bench.py --workload hv --steps 1 --warmup 1 > $O/bench_hv.log 2>&1; tail -c 600 $O/bench_hv.log; echo
date; echo done
