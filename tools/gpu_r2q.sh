#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_mb -- python $R/tools/microbench_conv.py --iters 10 --only T.resblock --arena > /dev/null 2>&1
python $R/tools/prof_summary.py $O/prof_mb $O/mb_kernel_stats.csv; head -12 $O/mb_kernel_stats.csv | cut -c1-200
cd $R
timeout 600 python -m pytest tests/test_distributed_gpu.py -q --tb=short 2>&1 | tail -8
rm -rf $O/prof_mb
