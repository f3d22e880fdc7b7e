#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for md in nostem nok7; do
  echo "== mode $md: differing runs of 150"
  NEMAR_SIDE_MODE=$md DIAG_RUNS=151 NEMAR_SIDE_STREAM=1 python tools/diag_hooks.py 2>&1 | grep "^run" | grep -v identical | wc -l
done
for md in all nostem nok7 wide; do
  NEMAR_SIDE_MODE=$md NEMAR_SIDE_STREAM=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --graph off > $O/b.json 2>/dev/null
  python -c "
import json; d = json.load(open('$O/b.json')); print('eager  mode $md  %.2f img/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
done
NEMAR_SIDE_STREAM=0 timeout 600 python bench.py --no-cpu-baseline --no-extras --graph off > $O/b.json 2>/dev/null
python -c "
import json; d = json.load(open('$O/b.json')); print('eager  side off  %.2f img/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
