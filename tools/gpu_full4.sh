#!/bin/bash
# round 4: the whole -m gpu tier (with the full-width row report) + the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
export NEMAR_FULL_REPORT=$O/full_rows.txt
rm -f $NEMAR_FULL_REPORT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.txt
unset NEMAR_FULL_REPORT
cat $O/pytest_gpu.txt
timeout 1200 python bench.py $2 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
try:
    d = json.load(open('$O/bench.json'))
    print('bench: %.2f img/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], d['launch']))
    for k in ('exact_route', 'other_configs', 'roofline_grid_sample'):
        print(k, json.dumps(d.get(k))[:900])
except Exception as e:
    print('bench parse failed', e); print(open('$O/bench.err').read()[-3000:])
PY
