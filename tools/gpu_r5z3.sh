#!/bin/bash
O=gpurun_out/r5z; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "split16_weight_gradient" 2>&1 | tail -2
for v in 0 1; do echo "== microbench, nemar_tune(38, $v)"; timeout 200 python tools/microbench_conv.py --batch 16 --iters 30 --only T.resblock --arena --tune 38 $v 2>&1 | grep -i "resblock" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   wgrad call %.1f us' % d['wgrad_us'])"; done
echo "== product library, no claim for the register-staged kernel"
NEMAR_AB_LIBRARY=0 DIAG_OWN_ONLY=1 timeout 200 python tools/diag_wgrad_beside.py 60000 4 64 dgrad_dual,agg_lds1k 2>&1 | grep "co-runner\|last event\|Error"
b() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print('%-52s %.1f img/s  %.2f ms/step' % (sys.argv[2], d['value'], d['ms_per_step']))
PY
}
for r in 1 2; do for t in "37=1,38=0" "38=1" "37=0,38=0"; do
NEMAR_TUNE=$t timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/b.json 2>$O/b.err; b $O/b.json "NEMAR_TUNE=$t (round $r)"
done; done
