#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -4 > $out/pytest_kernels.txt; cat $out/pytest_kernels.txt
bash tools/gpu_quick.sh $1
