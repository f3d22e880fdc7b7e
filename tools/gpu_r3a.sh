#!/bin/bash
# round 3, first GPU pass of the general 16-bit-pipe route: kernel tests, real shapes, per-layer A/B table, bench A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_conv_real_shapes_gpu.py -q -x -k "s16g or real_shape" > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
fmt='
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print("%-34s fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF" % (d["layer"], d["fwd_us"], d["fwd_TF"], d["dgrad_us"], d["dgrad_TF"], d["wgrad_us"], d["wgrad_TF"]))
'
for b in 8 16; do
  echo "== s16g route, batch $b" >> $O/mb.txt
  timeout 300 python tools/microbench_conv.py --iters 20 --batch $b 2>/dev/null | python -c "$fmt" >> $O/mb.txt
  echo "== exact-fp32 route, batch $b" >> $O/mb.txt
  timeout 300 python tools/microbench_conv.py --iters 20 --batch $b --tune 24 0 2>/dev/null | python -c "$fmt" >> $O/mb.txt
done
cat $O/mb.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
NEMAR_TUNE="24=0" timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_exact.json 2> $O/bench_exact.err
python -c "
import json
for f in ('bench.json','bench_exact.json'):
    try:
        d = json.load(open('$O/'+f)); print(f, '%.2f img/s  %.2f ms/step' % (d['value'], d['ms_per_step']))
    except Exception as e: print(f, 'failed', e)
"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd $R; python tools/prof_summary.py $O/stats $O/kernel_stats.csv > /dev/null 2>&1
head -40 $O/kernel_stats.csv | cut -c1-170
