#!/bin/bash
# Round-5 end state (after the product / measurement split and the register-staged wide weight gradient): the default bench line and
# rocprofv3 kernel stats of the same command on two streams and on one.  (PMC passes, layer table: tools/profile_round5.sh, unchanged kernels.)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 6 --warmup 2 --graph off --no-cpu-baseline --no-extras > $O/bench_line_two_streams.json 2>/dev/null
python $R/tools/prof_summary.py $O/stats $O/kernel_stats.csv > /dev/null 2>&1
rm -rf $O/stats
NEMAR_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d $O/stats1 -- python $R/bench.py --steps 6 --warmup 2 --graph off --no-cpu-baseline --no-extras > $O/bench_line_single_stream.json 2>/dev/null
python $R/tools/prof_summary.py $O/stats1 $O/kernel_stats_single_stream.csv > /dev/null 2>&1
rm -rf $O/stats1
python - <<PY
import csv, json
for tag in ('kernel_stats', 'kernel_stats_single_stream'):
    rows = list(csv.DictReader(open('$O/%s.csv' % tag)))
    tot = sum(float(r['total_us']) for r in rows); calls = sum(int(r['calls']) for r in rows)
    print('%s: kernels %.1f ms over 11 steps (8 + the 3 steps of the roofline / memory pass) = %.2f ms/step, %d launches = %d per step' % (tag, tot / 1e3, tot / 1.1e4, calls, calls // 11))
    for r in rows[:14]:
        print('   %6.2f%% x%-5s avg %8.1f us  %s' % (float(r['pct']), r['calls'], float(r['avg_us']), r['name'][:100]))
d = json.load(open('$O/bench.json'))
print('bench: %.1f img/s, %.2f ms/step, roofline %.3f (%.1f us per launch), operator %.3f, peak memory %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline_operator']['all_three']['frac'], d['peak_memory_GB']))
PY
