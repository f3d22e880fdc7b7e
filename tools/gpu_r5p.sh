#!/bin/bash
# round 5: step-level GPU tests with the gy hand-over off beside the side stream; speed of the three schedules on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
rm -f $O/full_rows.txt
NEMAR_FULL_REPORT=$O/full_rows.txt timeout 3000 python -m pytest tests/test_step_full_gpu.py tests/test_step_gpu.py -q 2>&1 | tail -6 | tee $O/pytest_tail.txt
bs() { python -c "
import json,sys; d = json.load(open('$1')); print('$2  %.2f img/s  %.2f ms/step  peak %.2f GB (single-stream order %.2f GB)' % (d['value'], d['ms_per_step'], d['peak_memory_GB']['timed_region'], d['peak_memory_GB']['single_stream_order']))"; }
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-extras --graph off > $O/b1.json 2>$O/b1.err; bs $O/b1.json "side stream, own gy split      "
NEMAR_GY_HANDOVER=1 python bench.py --no-cpu-baseline --no-extras --graph off > $O/b2.json 2>$O/b2.err; bs $O/b2.json "side stream, gy planes handed  "
NEMAR_SIDE_STREAM=0 python bench.py --no-cpu-baseline --no-extras --graph off > $O/b0.json 2>$O/b0.err; bs $O/b0.json "one stream (gy planes handed)  "
done 2>&1 | tee $O/ab.txt
