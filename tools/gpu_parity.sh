#!/bin/bash
# step-parity rows (tests/step_parity.py) of one configuration under several nemar_tune settings (and knife bands: "tune@band")
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; shift
name=$1; shift
cd $R
for t in "$@"; do
  tune=${t%@*}; band=${t#*@}; [ "$band" = "$t" ] && band=""
  f=$O/rows_${name}_"${t//[=,@]/_}".txt
  NEMAR_TUNE="$tune" timeout 900 python -c "
import sys; sys.path.insert(0,'tests')
import step_parity
if '$band': step_parity.KNIFE_BAND = float('$band')
rows = step_parity.run('$name', report='$f', check=False)
bad = [r for r in rows if not r[3]]
print('NEMAR_TUNE=%-14s band %s rows %d bad %d' % ('$tune', step_parity.KNIFE_BAND, len(rows), len(bad)))
for r in rows:
    if not r[3] or 'identical fakes' in r[0] or 'elementwise' in r[0] or 'adam' in r[0]: print('   ', 'ok  ' if r[3] else 'FAIL', r[0][:150], '%.3e / %.3e' % (r[1], r[2]))
" 2>/dev/null | grep -v "^initialize\|^model\|^---\|^\[Network"
done
