#!/bin/bash
# round-4 final check: smoke(), the whole GPU tier, the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/pytest_tail.txt
python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json; d = json.load(open('$O/bench.json')); print('bench: %.2f img/s  %.2f ms/step  %s  roofline %.1f TF (%.0f us) frac %.3f' % (d['value'], d['ms_per_step'], d['launch'][:30], d['roofline']['achieved'], d['roofline']['avg_launch_us'], d['roofline']['frac']))"
