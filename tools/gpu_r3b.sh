#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "split16 or absmax" --tb=short 2>&1 | tail -3
NEMAR_TL_LIB=nemar_amd/lib/libnemar_hip_tl.so timeout 120 python tools/timeline_split16.py 2>&1 | grep -v amdgpu.ids | tee $O/timeline_fp16x3.txt
timeout 200 python tools/microbench_conv.py --iters 30 --only T.resblock --arena 2>/dev/null | cut -c1-260
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d = json.load(open('$O/bench.json')); r = d['roofline']; print('bench: %.2f img/s  %.2f ms/step  roofline %.1f TF frac %.3f (%.0f us) traffic %s' % (d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['avg_launch_us'], r['traffic']))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/prof_summary.py $O/stats $O/kernel_stats.csv > /dev/null 2>&1; rm -rf $O/stats
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats.csv')))
tot=sum(float(r['total_us']) for r in rows)
print("total kernel time per step: %.2f ms" % (tot/9/1e3))
for r in rows[:22]:
    print("%-62s calls/step %5.1f avg %8.1f us  per step %6.2f ms" % (r['name'][:62].replace('(anonymous namespace)::',''), int(r['calls'])/9, float(r['avg_us']), float(r['total_us'])/9/1e3))
PY
