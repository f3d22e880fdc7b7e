#!/bin/bash
# ablations of the wave-specialised igemm on the resblock shape (timing only; results are wrong under ablation)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for d in 0 1536 1540 5632 5636; do echo "== dbg $d"; python tools/microbench_conv.py --iters 30 --only T.resblock --tune 2 $d 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('fwd %7.1f us | dgrad %7.1f us | wgrad %7.1f us' % (d['fwd_us'], d['dgrad_us'], d['wgrad_us']))
"; done
