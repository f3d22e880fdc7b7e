#!/bin/bash
# quick A/B of the general 16-bit-pipe route: per-layer table at batch 8 / 16 (+ optional bench)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
fmt='
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print("%-34s fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF" % (d["layer"], d["fwd_us"], d["fwd_TF"], d["dgrad_us"], d["dgrad_TF"], d["wgrad_us"], d["wgrad_TF"]))
'
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "s16g" 2>&1 | tail -3
for b in 8 16; do
  echo "== s16g route, batch $b" >> $O/mb.txt
  timeout 300 python tools/microbench_conv.py --iters 20 --batch $b 2>/dev/null | python -c "$fmt" >> $O/mb.txt
done
cat $O/mb.txt
if [ "$2" == "bench" ]; then
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python -c "
import json
d = json.load(open('$O/bench.json')); print('bench %.2f img/s  %.2f ms/step' % (d['value'], d['ms_per_step']))
"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd $R; python tools/prof_summary.py $O/stats $O/kernel_stats.csv > /dev/null 2>&1
head -45 $O/kernel_stats.csv | cut -c1-150
fi
