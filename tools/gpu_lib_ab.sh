#!/bin/bash
# same-box A/B of two builds of the product library: tools/gpu_lib_ab.sh <out-name> <libA.so> <libB.so> [rounds]   (bench.py through tools/bench_with_lib.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for i in $(seq 1 ${4:-3}); do
  for lib in $2 $3; do
    DIAG_LIB=$lib timeout 600 python tools/bench_with_lib.py --no-cpu-baseline --no-extras --graph off > $O/b.json 2>/dev/null
    python -c "
import json; d = json.load(open('$O/b.json')); print('%-50s %.2f img/s  %.2f ms/step  dominant kernel %.1f us' % ('$lib', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))"
  done
done
