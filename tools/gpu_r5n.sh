#!/bin/bash
# round 5: the step-level GPU tests (fixtures, routes, trajectory, side-stream equality) with the row report
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
rm -f $O/full_rows.txt
NEMAR_FULL_REPORT=$O/full_rows.txt timeout 3000 python -m pytest tests/test_step_full_gpu.py tests/test_step_gpu.py tests/test_api_gpu.py -q 2>&1 | tail -12 | tee $O/pytest_tail.txt
