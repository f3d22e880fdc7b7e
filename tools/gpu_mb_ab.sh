#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for t in "40 1" "40 4"; do echo "== tune $t"; python tools/microbench_conv.py --iters 30 --batch 16 --tune $t 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('%-34s fwd %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF' % (d['layer'], d['fwd_us'], d['fwd_TF'], d['dgrad_us'], d['dgrad_TF']))
"; done
