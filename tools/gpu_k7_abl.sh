#!/bin/bash
# ablations of k7_wgrad_kernel (nemar_tune key 2 bits: 1 no taps, 2 no Big loads, 4 no Small stores, 8 no conversion)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
for d in 0 1 2 4 8 3 15; do
  echo "== dbg $d" | tee -a $O/abl.txt
  timeout 200 python tools/microbench_conv.py --batch 16 --iters 20 --only k7 --tune 2 $d 2>&1 | grep layer | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-28s wgrad %7.1f us' % (d['layer'], d['wgrad_us']))" | tee -a $O/abl.txt
done
