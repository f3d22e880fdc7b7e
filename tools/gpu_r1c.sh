#!/bin/bash
out=gpurun_out/r1c; mkdir -p $out
./tools/probes/glds_unaligned > $out/glds_unaligned.txt 2>&1
run() { echo "== $*"; python tools/microbench_conv.py --iters 20 --only resblock "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('fwd %7.1f us %5.1f TF | dgrad %7.1f us | wgrad %7.1f us' % (d['fwd_us'], d['fwd_TF'], d['dgrad_us'], d['wgrad_us']))
"; }
{
cat $out/glds_unaligned.txt
run
run --tune 0 5
run --tune 2 16
run --tune 2 32
run --tune 2 48
run --tune 0 5 --tune 2 16
run --tune 0 5 --tune 2 48
run --tune 2 2
run --tune 2 50
} > $out/summary.txt 2>&1
cat $out/summary.txt
