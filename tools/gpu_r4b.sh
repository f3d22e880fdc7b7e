#!/bin/bash
# round 4, GPU pass B: the 7x7 stem / head kernels (conv_k7.hip) — parity through the C ABI, then per-call time with the route on / off
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "k7" 2>&1 | tail -12 > $O/pytest_k7.txt
cat $O/pytest_k7.txt
for t in "" "--tune 33 0"; do
  echo "== microbench_conv --batch 16 --only k7 $t" | tee -a $O/mb_k7.txt
  timeout 300 python tools/microbench_conv.py --batch 16 --iters 20 --only k7 $t 2>&1 | tail -4 | tee -a $O/mb_k7.txt
done
if [ -n "$2" ]; then
  timeout 600 python tests/step_parity.py $O/step_rows.txt > /dev/null 2>&1
  grep -n "FAIL\|EXCEPTION" $O/step_rows.txt | head -20
fi
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/tools/microbench_conv.py --batch 16 --iters 20 --only k7 > /dev/null 2>&1
cd $R; python tools/prof_summary.py $O/stats $O/k7_kernel_stats.csv > /dev/null 2>&1; head -14 $O/k7_kernel_stats.csv | cut -c1-170
