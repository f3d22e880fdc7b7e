#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
NEMAR_FULL_REPORT=$O/full_rows.txt timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 > $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
python tools/trace_convs.py > $O/conv_trace_cfg2.jsonl 2>/dev/null
timeout 300 python tools/microbench_trace.py $O/conv_trace_cfg2.jsonl > $O/layers.txt 2>$O/layers.err
head -34 $O/layers.txt
for kv in "17 0" "17 1" "17 2" "0 7"; do echo "== tune $kv"; python tools/microbench_conv.py --iters 30 --only T.resblock --tune $kv 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('%-34s fwd %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF | wgrad %7.1f us %5.1f TF' % (d['layer'], d['fwd_us'], d['fwd_TF'], d['dgrad_us'], d['dgrad_TF'], d['wgrad_us'], d['wgrad_TF']))
"; done
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d = json.load(open('$O/bench.json')); print('bench: %.2f img/s  %.2f ms/step  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
