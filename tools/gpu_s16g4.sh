#!/bin/bash
# s16g kernels with 16-byte halo loads: kernel tests, the layer table of a step, bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "s16g or k7 or conv" 2>&1 | tail -3
timeout 900 python tools/microbench_trace.py profiles/r4_conv_trace.jsonl > $O/conv_layers.txt 2> $O/layers.err
grep -c . $O/conv_layers.txt; head -3 $O/conv_layers.txt
bash tools/gpu_ab.sh $1 "" ""
