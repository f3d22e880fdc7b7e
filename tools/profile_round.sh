#!/bin/bash
# Runs on the GPU box (via gpurun): bench JSON, rocprofv3 kernel stats of the same command, and separate PMC passes
# (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950) for the dominant conv kernel and the grid_sample kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_conv_$c -- python $R/tools/pmc_conv.py fwd 4 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_warp_$c -- python $R/tools/microbench.py --iters 3 > /dev/null 2>&1
done
python $R/tools/microbench.py > $O/microbench.jsonl 2>/dev/null
cat $O/bench.json
