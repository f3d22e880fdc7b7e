#!/bin/bash
# round-end evidence: full GPU suite, smoke, then the profile script (bench + rocprof stats + PMC passes)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
NEMAR_FULL_REPORT=$O/full_rows.txt NEMAR_SPLIT16_REPORT=$O/split16_accuracy.txt timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -12 > $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round2.sh $1 2>&1 | tail -3 | cut -c1-600
