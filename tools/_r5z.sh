#!/bin/bash
# round-5 call z (last): the whole GPU suite and smoke() on the final tree
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 345 python -m pytest tests -m gpu -x -q > $O/r5z_gpu_tests.log 2>&1; echo "gpu tests rc=$? $(tail -1 $O/r5z_gpu_tests.log)"
timeout 45 python -c "import __graft_entry__ as g; g.smoke()" > $O/r5z_smoke.log 2>&1; echo "smoke rc=$? $(tail -2 $O/r5z_smoke.log | tr '\n' ' ')"
