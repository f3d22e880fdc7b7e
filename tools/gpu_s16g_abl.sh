#!/bin/bash
# ablations of s16g_kernel (nemar_tune(2, bits)): 1 no tap loop, 2 no source loads, 4 no conversion / LDS stores, 8 no weight DMA,
# 16 no max phase, 128 return before the epilogue.  Timing only (results are wrong under ablation).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for d in ${3:-0 8 128 136 2 10 138 1 139}; do echo "== dbg $d"; for sh in T.down1 "R.res 32" T.down2; do python tools/microbench_conv.py --iters 30 --batch ${2:-16} --only "$sh" --tune 2 $d 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('%-34s fwd %7.1f us | dgrad %7.1f us' % (d['layer'], d['fwd_us'], d['dgrad_us']))
"; done; done
