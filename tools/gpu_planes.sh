#!/bin/bash
# fused InstanceNorm -> fp16 x 3 planes producer: tests, step parity, bench A/B (NEMAR_PLANES=0/1), kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "planes or producer or norm" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_step_full_gpu.py tests/test_nets_gpu.py -q -x 2>&1 | tail -3
for v in 1 0 1 0; do
  NEMAR_PLANES=$v timeout 600 python bench.py --no-cpu-baseline --no-extras --graph off > $O/bench_planes$v.json 2>/dev/null
  python -c "
import json
d = json.load(open('$O/bench_planes$v.json')); print('NEMAR_PLANES=$v %.2f img/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd $R; python tools/prof_summary.py $O/stats $O/kernel_stats.csv > /dev/null 2>&1
grep "instnorm\|dropout\|split_planes\|absmax\|igemm_split16_kernel<2, 2, 3>" $O/kernel_stats.csv | cut -c1-200
