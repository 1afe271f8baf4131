#!/bin/bash
# round-5 call y: second pass over the 8-bit AdamW kernel (ALU diet) -- its tests and its HBM fraction; the first-step distance of two pipeline lanes from one
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 150 python -m pytest tests/test_gpu_optim.py -x -q > $O/r5y_optim_tests.log 2>&1; echo "optim tests rc=$? $(tail -1 $O/r5y_optim_tests.log)"
timeout 100 python tools/optim_timing.py 2>&1 | tail -4 | tee $O/r5y_optim_timing.jsonl
timeout 100 python -m pytest tests/test_gpu_pipeline.py -k two_pipeline_lanes -x -q -s > $O/r5y_pipe_lanes.log 2>&1; echo "pipe lanes rc=$? $(tail -1 $O/r5y_pipe_lanes.log)"; grep "pipe lanes 2 vs 1" $O/r5y_pipe_lanes.log
