#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "split16" --tb=short 2>&1 | tail -3
for m in "" "--arena"; do echo "== microbench $m"; timeout 200 python tools/microbench_conv.py --iters 30 --only T.resblock $m 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('%-34s fwd %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF | wgrad %7.1f us %5.1f TF' % (d['layer'], d['fwd_us'], d['fwd_TF'], d['dgrad_us'], d['dgrad_TF'], d['wgrad_us'], d['wgrad_TF']))
"; done | tee $O/microbench.txt
NEMAR_SPLIT16_REPORT=$O/split16_accuracy.txt timeout 600 python -m pytest tests/test_conv_real_shapes_gpu.py -q -k "split16" --tb=short 2>&1 | tail -8; cat $O/split16_accuracy.txt
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d = json.load(open('$O/bench.json')); print('bench: %.2f img/s  %.2f ms/step  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_step_full_gpu.py -q --tb=short 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/prof_summary.py $O/stats $O/kernel_stats.csv > /dev/null 2>&1; rm -rf $O/stats
head -16 $O/kernel_stats.csv | cut -c1-140
