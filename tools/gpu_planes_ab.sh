#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
for v in 1 0 1 0; do
  NEMAR_PLANES=$v timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_planes$v.json 2>/dev/null
  python -c "
import json
d = json.load(open('$O/bench_planes$v.json')); print('NEMAR_PLANES=$v %.2f img/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd $R; python tools/prof_summary.py $O/stats $O/kernel_stats.csv > /dev/null 2>&1
grep "instnorm_planes\|dropout\|split_planes" $O/kernel_stats.csv | cut -c1-200
