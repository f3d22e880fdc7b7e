#!/bin/bash
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -2
run() { echo "== $*"; python tools/microbench_conv.py --iters 30 "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('%-30s fwd %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF | wgrad %7.1f us %5.1f TF' % (d['layer'][:30], d['fwd_us'], d['fwd_TF'], d['dgrad_us'], d['dgrad_TF'], d['wgrad_us'], d['wgrad_TF']))
"; }
run --only D.l4
run --only D.l4 --tune 12 0
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench: %.2f img/s  %.2f ms/step  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
