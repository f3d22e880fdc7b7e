#!/bin/bash
# round 4, GPU pass A: the new full-width fixtures (reference default geometry 288x384, non-square unet 256x384, the config-5
# registration sub-model vs fp64) with the route of every convolution call, the hipGraph tests after the warm-up / stale-loss fixes,
# the matrix-pipe ceiling probe (operand data / type / duration) and the full default bench line with its side measurements
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
export NEMAR_FULL_REPORT=$O/full_rows.txt
rm -f $NEMAR_FULL_REPORT
timeout 900 python -m pytest tests/test_step_full_gpu.py -x -q -k "default_full or c2_256x384 or registration or c2_full" 2>&1 | tail -15 > $O/pytest_full.txt
unset NEMAR_FULL_REPORT
timeout 600 python -m pytest tests/test_step_gpu.py -x -q 2>&1 | tail -8 > $O/pytest_step.txt
timeout 120 tools/probes/_build/mfma_peak_modes > $O/mfma_peak_modes.txt 2>&1
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/pytest_full.txt $O/pytest_step.txt $O/mfma_peak_modes.txt
python - <<PY
import json
try:
    d = json.load(open('$O/bench.json'))
    print('bench: %.2f img/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], d['launch']))
    for k in ('exact_route', 'route_agreement', 'other_configs', 'roofline_grid_sample'):
        print(k, json.dumps(d.get(k))[:1500])
except Exception as e:
    print('bench parse failed', e); print(open('$O/bench.err').read()[-3000:])
PY
