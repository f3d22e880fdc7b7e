#!/bin/bash
# gy split once for the data and the weight gradient (nemar_tune 35) on top of the one-copy gy planes (34): kernel tests, step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "split16" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_step_gpu.py -x -q 2>&1 | tail -3
bash tools/gpu_ab.sh $1 "" "35=0" "35=0,34=0" ""
