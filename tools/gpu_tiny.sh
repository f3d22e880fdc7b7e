#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "split_reduction or conv_fwd or conv_bwd_data" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_nets_gpu.py tests/test_step_gpu.py -x -q -k "not pack_plans" 2>&1 | tail -2
grep -E '"H": (2|4|8|16),' profiles/r4_conv_trace.jsonl > /tmp/tiny.jsonl
python tools/microbench_trace.py /tmp/tiny.jsonl 2>/dev/null | head -14
bash tools/gpu_ab.sh $1 "" ""
