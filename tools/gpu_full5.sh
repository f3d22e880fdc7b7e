#!/bin/bash
# full GPU tier + default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_tail.txt
python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
python -c "
import json; d = json.load(open('$O/bench.json')); print('bench: %.2f img/s  %.2f ms/step  %s  probe %s  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['launch'], d.get('launch_probe'), d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
