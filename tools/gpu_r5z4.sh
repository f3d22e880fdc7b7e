#!/bin/bash
timeout 200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "split16_weight_gradient" 2>&1 | tail -1
for v in 0 1; do echo "== microbench, nemar_tune(38, $v)"; timeout 100 python tools/microbench_conv.py --batch 16 --iters 30 --only T.resblock --arena --tune 38 $v 2>&1 | grep -i "resblock" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   wgrad call %.1f us' % d['wgrad_us'])"; done
NEMAR_AB_LIBRARY=0 DIAG_OWN_ONLY=1 timeout 100 python tools/diag_wgrad_beside.py 30000 4 64 dgrad_dual 2>&1 | grep "co-runner\|last event\|Error"
