#!/bin/bash
# split-16 kernels: kernel + real-shape tests, layer timings with / without the arena, bench, step parity suites
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "split16 or absmax" --tb=short 2>&1 | tail -3
timeout 600 python -m pytest tests/test_conv_real_shapes_gpu.py -q -k "split16" --tb=short 2>&1 | tail -4
for m in "" "--arena"; do timeout 200 python tools/microbench_conv.py --iters 30 --only "D.l4" $m 2>/dev/null | cut -c1-250; done
timeout 200 python tools/microbench_conv.py --iters 30 --only T.resblock --arena 2>/dev/null | cut -c1-260
python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err; python -c "
import json; d = json.load(open('$O/bench.json')); r = d['roofline']; print('bench: %.2f img/s  %.2f ms/step  roofline %.1f TF frac %.3f (%.0f us) traffic %s' % (d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['avg_launch_us'], r['traffic']))"
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_step_full_gpu.py tests/test_nets_gpu.py -q --tb=short 2>&1 | tail -5
