#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/bench.py --steps 8 --warmup 3 --graph off --no-cpu-baseline --no-extras > /dev/null 2>&1
python $R/tools/overlap_from_trace.py $O/tr 0.5 | tee $O/overlap.txt
head -2 $(ls $O/tr/*/*kernel_trace.csv | head -1) | cut -c1-300
rm -rf $O/tr
