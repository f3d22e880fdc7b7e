#!/bin/bash
# The round's evidence run (on the GPU box, via gpurun: `tools/profile_evidence.sh <out-name>`): the default bench line (with its extras),
# rocprofv3 kernel stats of the same command on two streams and on one, the traffic / SQ PMC passes of the dominant kernel (separate
# passes, no trace domains next to --pmc), the per-layer convolution table, the grid_sample micro-benchmark.  Copy what is to be judged
# from gpurun_out/<out-name>/ into profiles/ (named per round).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 6 --warmup 2 --graph off --no-cpu-baseline --no-extras > $O/bench_line_two_streams.json 2>/dev/null
python $R/tools/prof_summary.py $O/stats $O/kernel_stats.csv > /dev/null 2>&1
rm -rf $O/stats
NEMAR_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d $O/stats1 -- python $R/bench.py --steps 6 --warmup 2 --graph off --no-cpu-baseline --no-extras > $O/bench_line_single_stream.json 2>/dev/null
python $R/tools/prof_summary.py $O/stats1 $O/kernel_stats_single_stream.csv > /dev/null 2>&1
rm -rf $O/stats1
G1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
pmc() {  # tag shape which
  for c in "$G1" FETCH_SIZE WRITE_SIZE; do
    t=${c%% *}
    NEMAR_PMC_SHAPE=$2 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$1_$3_$t -- python $R/tools/pmc_conv.py $3 4 > /dev/null 2>&1
    echo "=== $1 $3 ($2) $t"; python $R/tools/pmc_summary.py $O/pmc_$1_$3_$t
  done
}
{
pmc resblock16 16,256,256,64,3,1,1,1 fwd
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_warp_$c -- python $R/tools/microbench.py --iters 3 > /dev/null 2>&1
  echo "=== warp $c"; python $R/tools/pmc_summary.py $O/pmc_warp_$c grid_sample; python $R/tools/pmc_summary.py $O/pmc_warp_$c far_
  echo "=== instnorm $c"; python $R/tools/pmc_summary.py $O/pmc_warp_$c instnorm
done
} > $O/pmc_summary.txt 2>&1
# the one-node ResnetBlock (forward + backward, batch 16, dropout on): HBM bytes per kernel
{
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_block_$c -- python $R/tools/pmc_block.py 4 1 > /dev/null 2>&1
  echo "=== one-node ResnetBlock $c (KiB per launch; FETCH_SIZE x 2 = bytes)"; python $R/tools/pmc_summary.py $O/pmc_block_$c | grep -v "distribution\|fillBuffer"
done
} > $O/pmc_block.txt 2>&1
rm -rf $O/pmc_block_*
python $R/tools/microbench.py > $O/microbench.jsonl 2>/dev/null
cd $R
timeout 600 python tools/trace_convs.py > $O/conv_trace.jsonl 2> $O/trace.err
timeout 900 python tools/microbench_trace.py $O/conv_trace.jsonl > $O/conv_layers.txt 2> $O/layers.err
python - <<PY
import csv, json
# steps of the profiled process: 2 warm-up + 6 timed + 3 of the roofline / memory pass on one stream (+ 2 of the in-step roofline pass when the side stream is on)
for tag, ns in (('kernel_stats', 13), ('kernel_stats_single_stream', 11)):
    rows = list(csv.DictReader(open('$O/%s.csv' % tag)))
    tot = sum(float(r['total_us']) for r in rows); calls = sum(int(r['calls']) for r in rows)
    print('%s: kernels %.1f ms over %d steps = %.2f ms/step, %d launches = %d per step' % (tag, tot / 1e3, ns, tot / 1e3 / ns, calls, calls // ns))
    setup = [r for r in rows if 'at::native' in r['name'] or '__amd_rocclr' in r['name']]
    sc = sum(int(r['calls']) for r in setup)
    print('   of which %d launches (%.1f ms) are the one-time set-up of the process (ATen weight initialisation, parameter copies into the flat buffers, runtime '
          'fills): the steps themselves launch (%d - %d) / %d = %d kernels each' % (sc, sum(float(r['total_us']) for r in setup) / 1e3, calls, sc, ns, (calls - sc) // ns))
    for r in rows[:12]:
        print('   %6.2f%% x%-5s avg %8.1f us  %s' % (float(r['pct']), r['calls'], float(r['avg_us']), r['name'][:100]))
PY
cat $O/bench.json
rm -rf $O/pmc_resblock16_* $O/pmc_warp_*
