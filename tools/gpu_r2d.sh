#!/bin/bash
# tests + warp microbench + bench/kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
NEMAR_FULL_REPORT=$O/full_rows.txt timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 > $O/pytest_gpu.txt
tail -12 $O/pytest_gpu.txt
python tools/microbench.py --iters 20 > $O/microbench.jsonl 2>$O/microbench.err
python - <<PY
import json
for l in open('$O/microbench.jsonl'):
    d=json.loads(l)
    if d['op'].startswith('grid_sample'):
        print('%-20s %-32s %-18s sigma=%-10s %8.1f us %7.0f GB/s' % (d['op'], d.get('variant',''), d['shape'], d['sigma'], d['us'], d['GBps']))
PY
bash tools/gpu_quick.sh $1 | tail -32
