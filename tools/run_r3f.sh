#!/bin/bash
# Round 3, GPU call F: 8-bit multi-tensor test re-run; operand row-pitch probe on the heaviest GEMM descriptors of the step.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
echo "== tests"; date
timeout 600 python -m pytest tests/test_gpu_optim.py -q -m gpu -p no:cacheprovider > $O/tests.txt 2>&1
tail -4 $O/tests.txt | cut -c1-300
echo "== pitch probe"; date
timeout 600 python tools/gemm_pitch_probe.py profiles/r2_gemm_trace_sdxl_step.json $O/gemm_pitch_probe.jsonl --top=16 --pads=0,64,136 2>&1 | cut -c1-700
du -sh $O; date; echo done
