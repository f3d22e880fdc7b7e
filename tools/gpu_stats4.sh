#!/bin/bash
# bench line (graph replay) + rocprofv3 kernel stats of a short bench run
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
python -c "
import json; d = json.load(open('$O/bench.json')); print('bench: %.2f img/s  %.2f ms/step  %s  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['launch'], d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd $R; python tools/prof_summary.py $O/stats $O/kernel_stats.csv > /dev/null 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open('$O/kernel_stats.csv')))
tot = sum(float(r['total_us']) for r in rows); calls = sum(int(r['calls']) for r in rows)
print('kernels: %.1f ms total over the trace, %d launches' % (tot / 1e3, calls))
for r in rows[:45]:
    print('%6.2f%% %9.1f us x%-5s avg %8.1f  %s' % (float(r['pct']), float(r['total_us']), r['calls'], float(r['avg_us']), r['name'][:110]))
PY
