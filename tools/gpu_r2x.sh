#!/bin/bash
# per-kernel profile of the bench step (fp16 x 3 split convolutions default) + full GPU suite + bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/prof_summary.py $O/stats $O/kernel_stats.csv > /dev/null 2>&1; rm -rf $O/stats
head -28 $O/kernel_stats.csv | cut -c1-150
cd $R
NEMAR_FULL_REPORT=$O/full_rows.txt NEMAR_SPLIT16_REPORT=$O/split16_accuracy.txt timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -12 > $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt; cat $O/split16_accuracy.txt
python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d = json.load(open('$O/bench.json')); print('bench: %.2f img/s  %.2f ms/step  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])); print(d['cpu_baseline'])"
