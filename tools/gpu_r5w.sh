#!/bin/bash
# round-5 end state: the whole GPU tier, smoke(), the default bench line
O=gpurun_out/r5w; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/r5w/bench.json'))
print('%.1f img/s  %.2f ms/step  roofline %.3f  operator %.3f  launch: %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_operator']['all_three']['frac'], d['launch'][:60]))
PY
