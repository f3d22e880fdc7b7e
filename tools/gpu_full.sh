#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $out/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -2
bash tools/gpu_quick.sh $1 | tail -45
