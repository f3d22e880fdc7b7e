#!/bin/bash
# A/B of nemar_tune settings on one box: bench.py under each NEMAR_TUNE string given as arguments ("" = defaults)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; shift
cd $R
for t in "$@"; do
  NEMAR_TUNE="$t" timeout 600 python bench.py --no-cpu-baseline --no-extras --graph off > $O/bench_"${t//[=,]/_}".json 2>/dev/null
  python -c "
import json,sys
d = json.load(open('$O/bench_${t//[=,]/_}.json')); print('NEMAR_TUNE=%-12s %.2f img/s  %.2f ms/step' % ('$t', d['value'], d['ms_per_step']))"
done
