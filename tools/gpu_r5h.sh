#!/bin/bash
# round 5: full GPU tier (with the all-configuration side-stream test and the route log), then the 7x7-on-side schedule A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
rm -f $O/full_rows.txt
NEMAR_FULL_REPORT=$O/full_rows.txt timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $O/pytest_tail.txt
bs() { python -c "
import json,sys; d = json.load(open('$1')); print('$2  %.2f img/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], d.get('launch','')[:50]))"; }
for i in 1 2 3; do
NEMAR_SIDE_K7=1 python bench.py --no-cpu-baseline --no-extras --graph off > $O/b1.json 2>$O/b1.err; bs $O/b1.json "7x7 on side   "
NEMAR_SIDE_K7=0 python bench.py --no-cpu-baseline --no-extras --graph off > $O/b0.json 2>$O/b0.err; bs $O/b0.json "7x7 on compute"
done 2>&1 | tee $O/ab.txt
