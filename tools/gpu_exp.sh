#!/bin/bash
run() { echo "== $*"; python tools/microbench_conv.py --iters 30 "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('%-30s fwd %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF | wgrad %7.1f us %5.1f TF' % (d['layer'][:30], d['fwd_us'], d['fwd_TF'], d['dgrad_us'], d['dgrad_TF'], d['wgrad_us'], d['wgrad_TF']))
"; }
run --only down
run --only down --tune 6 200
run --only down --tune 6 120
run --only D.l2
run --only D.l2 --tune 6 200
run --only R.
run --only R. --tune 6 200
