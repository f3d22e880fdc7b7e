#!/bin/bash
# whole-CU LDS claim of the split-16 kernels: stand-alone probe, in-situ event rate, bench A/B (hand-over on / off, claim on / off)
O=gpurun_out/r5u; mkdir -p $O
timeout 300 python tools/diag_wgrad_beside.py 60000 4 64 dgrad_dual,agg_lds1k 2>&1 | grep "co-runner\|last event\|Error" | tee $O/probe.txt
DIAG_SECONDS=35 timeout 300 python tools/diag_step_events.py base 2>&1 | grep "experiment\|Error" | tee $O/events.txt
b() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print('%-46s %.1f img/s  %.2f ms/step  (%s)' % (sys.argv[2], d['value'], d['ms_per_step'], d['launch'][:40]))
PY
}
for r in 1 2; do
NEMAR_GY_HANDOVER=1 timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/b_ho1_$r.json 2>$O/b.err; b $O/b_ho1_$r.json "claim on, hand-over on (round $r)"
NEMAR_GY_HANDOVER=0 timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/b_ho0_$r.json 2>$O/b.err; b $O/b_ho0_$r.json "claim on, hand-over off (round $r)"
NEMAR_TUNE=37=0 NEMAR_GY_HANDOVER=0 timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/b_old_$r.json 2>$O/b.err; b $O/b_old_$r.json "claim off, hand-over off = before (round $r)"
done 2>&1 | tee $O/bench_ab.txt
