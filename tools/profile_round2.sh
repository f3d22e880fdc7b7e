#!/bin/bash
# Round-2 evidence run (on the GPU box, via gpurun): bench JSON, rocprofv3 kernel stats of the same command, SQ + traffic PMC
# passes over the dominant conv kernels (separate passes: FETCH_SIZE / WRITE_SIZE cannot share one on gfx950) and over the
# grid_sample kernels, and the microbenchmark table.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
python $R/tools/prof_summary.py $O/stats $O/kernel_stats.csv > /dev/null 2>&1
G1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
for which in fwd dgrad wgrad; do
  rocprofv3 --pmc $G1 --kernel-trace --output-format csv -d $O/sq_$which -- python $R/tools/pmc_conv.py $which 6 > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${which}_$c -- python $R/tools/pmc_conv.py $which 4 > /dev/null 2>&1
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_warp_$c -- python $R/tools/microbench.py --iters 3 > /dev/null 2>&1
done
{
for which in fwd dgrad wgrad; do echo "=== $which SQ"; python $R/tools/pmc_summary.py $O/sq_$which; for c in FETCH_SIZE WRITE_SIZE; do echo "=== $which $c"; python $R/tools/pmc_summary.py $O/pmc_${which}_$c; done; done
for c in FETCH_SIZE WRITE_SIZE; do echo "=== warp $c"; python $R/tools/pmc_summary.py $O/pmc_warp_$c grid_sample; python $R/tools/pmc_summary.py $O/pmc_warp_$c far_; done
} > $O/pmc_summary.txt 2>&1
python $R/tools/microbench.py > $O/microbench.jsonl 2>/dev/null
cat $O/bench.json
rm -rf $O/stats/*/*.db $O/sq_* $O/pmc_fwd_* $O/pmc_dgrad_* $O/pmc_wgrad_* $O/pmc_warp_*
