#!/bin/bash
# round 5: the producer / consumer form of the in-kernel-split kernel — kernel tests, step fixtures, A/B against the sequential form
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "s16g or instnorm or k7 or transpose" 2>&1 | tail -4
rm -f $O/full_rows.txt
NEMAR_FULL_REPORT=$O/full_rows.txt timeout 1500 python -m pytest tests/test_step_full_gpu.py tests/test_step_gpu.py -x -q 2>&1 | tail -6
bs() { python -c "
import json,sys; d = json.load(open('$1')); print('$2  %.2f img/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], d.get('launch','')[:50]))"; }
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-extras --graph off > $O/b1.json 2>$O/b1.err; bs $O/b1.json "producer/consumer"
NEMAR_TUNE="38=0" python bench.py --no-cpu-baseline --no-extras --graph off > $O/b0.json 2>$O/b0.err; bs $O/b0.json "sequential (38=0)"
done 2>&1 | tee $O/ab.txt
