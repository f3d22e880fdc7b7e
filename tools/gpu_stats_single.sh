#!/bin/bash
# rocprofv3 kernel stats of the bench step on ONE stream (NEMAR_SIDE_STREAM=0, eager): per-kernel durations undisturbed by overlap
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
NEMAR_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 6 --warmup 2 --graph off --no-cpu-baseline --no-extras > $O/bench_line.json 2>/dev/null
cd $R; python tools/prof_summary.py $O/stats $O/kernel_stats.csv > /dev/null 2>&1
python - <<PY
import csv, json
rows = list(csv.DictReader(open('$O/kernel_stats.csv')))
tot = sum(float(r['total_us']) for r in rows); calls = sum(int(r['calls']) for r in rows)
d = json.load(open('$O/bench_line.json'))
print('bench under rocprofv3: %.2f ms/step; kernels %.1f ms over 10 steps (8 + the 2 steps of the roofline pass) = %.2f ms/step, %d launches = %d per step' % (d['ms_per_step'], tot / 1e3, tot / 1e4, calls, calls // 10))
print('roofline kernel, event-timed in bench.py: %.1f us; in this trace:' % d['roofline']['avg_launch_us'])
for r in rows[:3]:
    print('   %6.2f%% x%-5s avg %8.1f us  %s' % (float(r['pct']), r['calls'], float(r['avg_us']), r['name'][:90]))
PY
rm -rf $O/stats
