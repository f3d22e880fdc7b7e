#!/bin/bash
# tests + warp microbench + conv A/B of tuning switches on the resblock shape
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
NEMAR_FULL_REPORT=$O/full_rows.txt timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 > $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
python tools/microbench.py --iters 20 > $O/microbench.jsonl 2>$O/microbench.err
python - <<PY
import json
for l in open('$O/microbench.jsonl'):
    d=json.loads(l)
    if d['op'].startswith('grid_sample') and 'gather' in d.get('variant',''):
        print('%-20s %-32s %-18s sigma=%-10s %8.1f us %7.0f GB/s' % (d['op'], d.get('variant',''), d['shape'], d['sigma'], d['us'], d['GBps']))
PY
for kv in "16 1" "16 0"; do echo "== tune $kv"; python tools/microbench_conv.py --iters 30 --tune $kv 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('%-34s fwd %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF | wgrad %7.1f us %5.1f TF' % (d['layer'], d['fwd_us'], d['fwd_TF'], d['dgrad_us'], d['dgrad_TF'], d['wgrad_us'], d['wgrad_TF']))
" | head -6; done
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d = json.load(open('$O/bench.json')); print('bench: %.2f img/s  %.2f ms/step  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
