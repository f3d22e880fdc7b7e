#!/bin/bash
# side-stream weight-gradient branch (NEMAR_SIDE_STREAM): step tests, A/B eager and graph
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_step_gpu.py tests/test_nets_gpu.py -x -q 2>&1 | tail -4
for s in 1 0 1 0; do
  NEMAR_SIDE_STREAM=$s timeout 600 python bench.py --no-cpu-baseline --no-extras --graph off > $O/b.json 2>/dev/null
  python -c "
import json; d = json.load(open('$O/b.json')); print('eager  NEMAR_SIDE_STREAM=$s  %.2f img/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
done
for s in 1 0; do
  NEMAR_SIDE_STREAM=$s timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/b.json 2>$O/b.err
  python -c "
import json; d = json.load(open('$O/b.json')); print('graph  NEMAR_SIDE_STREAM=$s  %.2f img/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], d['launch']))" || tail -5 $O/b.err
done
