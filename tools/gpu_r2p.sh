#!/bin/bash
# round 2, split-bf16 convolution bring-up: kernel tests -> accuracy report -> layer timings -> full suite -> bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "split_bf16" --tb=short 2>&1 | tail -15 > $O/bf6_kernels.txt; tail -5 $O/bf6_kernels.txt
NEMAR_BF6_REPORT=$O/bf6_accuracy.txt timeout 600 python -m pytest tests/test_conv_real_shapes_gpu.py -q -k "split_bf16" --tb=short 2>&1 | tail -25 > $O/bf6_real.txt; tail -8 $O/bf6_real.txt; cat $O/bf6_accuracy.txt
for mode in "" "--arena"; do echo "== microbench $mode"; timeout 300 python tools/microbench_conv.py --iters 30 --only T.resblock $mode 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('%-34s fwd %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF | wgrad %7.1f us %5.1f TF' % (d['layer'], d['fwd_us'], d['fwd_TF'], d['dgrad_us'], d['dgrad_TF'], d['wgrad_us'], d['wgrad_TF']))
"; done | tee $O/microbench.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_mb -- python $R/tools/microbench_conv.py --iters 10 --only T.resblock --arena > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob
for f in glob.glob('$O/prof_mb/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:8]:
        print('%-90s calls %5s avg %9.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
NEMAR_FULL_REPORT=$O/full_rows.txt timeout 1500 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -30 > $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d = json.load(open('$O/bench.json')); print('bench: %.2f img/s  %.2f ms/step  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
