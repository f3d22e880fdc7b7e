#!/bin/bash
# which claim costs what: bench A/B of the claim mask (measurement build), hand-over on
O=gpurun_out/r5v; mkdir -p $O
b() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print('%-46s %.1f img/s  %.2f ms/step' % (sys.argv[2], d['value'], d['ms_per_step']))
PY
}
for r in 1 2; do for m in 3 1 2 0; do
NEMAR_TUNE=37=$m NEMAR_GY_HANDOVER=1 timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/b_$m_$r.json 2>$O/b.err; b $O/b_$m_$r.json "claim mask $m, hand-over on (round $r)"
done; done 2>&1 | tee $O/bench_mask.txt
