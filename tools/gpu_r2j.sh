#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
NEMAR_FULL_REPORT=$O/full_rows.txt timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 > $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
for kv in "18 3" "18 4" "18 5"; do echo "== tune $kv"; python tools/microbench_conv.py --iters 30 --only T.resblock --tune $kv 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('%-34s fwd %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF | wgrad %7.1f us %5.1f TF' % (d['layer'], d['fwd_us'], d['fwd_TF'], d['dgrad_us'], d['dgrad_TF'], d['wgrad_us'], d['wgrad_TF']))
"; done
timeout 300 python tools/microbench_trace.py tools/conv_trace_cfg2.jsonl > $O/layers.txt 2>$O/layers.err
head -30 $O/layers.txt
python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d = json.load(open('$O/bench.json')); print('bench: %.2f img/s  %.2f ms/step  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])); print(d['cpu_baseline'])"
