#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the grid_sample kernels at ONE shape and the sigma = 0 regime (what the training step runs): per-launch
# numbers for bench.py's roofline_grid_sample.traffic.  usage: gpu_pmc_warp.sh OUT SIZE
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for size in 256 1024; do
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/w_${size}_$c -- python $R/tools/microbench.py --iters 4 --size $size --sigma0 > /dev/null 2>&1
  echo "=== grid_sample ${size}x${size} sigma=0 $c"; python $R/tools/pmc_summary.py $O/w_${size}_$c grid_sample; python $R/tools/pmc_summary.py $O/w_${size}_$c far_
done
done > $O/pmc_warp.txt 2>&1
rm -rf $O/w_*
cat $O/pmc_warp.txt
