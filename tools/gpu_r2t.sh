#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "split16" --tb=short 2>&1 | tail -3
fmt='
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print("%-34s fwd %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF" % (d["layer"], d["fwd_us"], d["fwd_TF"], d["dgrad_us"], d["dgrad_TF"]))
'
for t in "21 3" "21 4"; do
  echo "== tune $t"; timeout 200 python tools/microbench_conv.py --iters 30 --only T.resblock --arena --tune $t 2>/dev/null | python -c "$fmt"
done | tee $O/microbench.txt
NEMAR_TUNE="21=4" NEMAR_SPLIT16_REPORT=$O/bf6_accuracy_v4.txt timeout 600 python -m pytest tests/test_conv_real_shapes_gpu.py -q -k "split16" --tb=short 2>&1 | tail -5; cat $O/bf6_accuracy_v4.txt
for v in 4; do
NEMAR_TUNE="21=$v" python bench.py --no-cpu-baseline > $O/bench_v$v.json 2> $O/bench.err; python -c "
import json; d = json.load(open('$O/bench_v$v.json')); print('bench variant $v: %.2f img/s  %.2f ms/step  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
done
NEMAR_TUNE="21=4" timeout 900 python -m pytest tests/test_step_gpu.py tests/test_step_full_gpu.py -q --tb=short 2>&1 | tail -6
