#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_step_gpu.py -x -q -k "side_stream or parity or graph" 2>&1 | tail -3
echo "differing runs of 150:"; DIAG_RUNS=151 python tools/diag_hooks.py 2>&1 | grep "^run" | grep -v identical | wc -l
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
python -c "
import json; d = json.load(open('$O/bench.json')); print('bench: %.2f img/s  %.2f ms/step  %s  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['launch'][:40], d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
done
