#!/bin/bash
# round-2 first pass: full GPU test tier (incl. full-width + real-shape cases), per-layer conv table, bench, kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
NEMAR_FULL_REPORT=$O/full_rows.txt timeout 1200 python -m pytest tests -m gpu -q --tb=short -k "${2:-not nothing}" 2>&1 | tail -60 > $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
timeout 300 python tools/microbench_trace.py tools/conv_trace_cfg2.jsonl > $O/layers.txt 2>$O/layers.err
head -30 $O/layers.txt
bash tools/gpu_quick.sh $1 | tail -40
