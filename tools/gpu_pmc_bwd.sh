#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p_$c -- python $R/tools/pmc_bwd_pair.py 4 > /dev/null 2>&1
  echo "=== $c"; python $R/tools/pmc_summary.py $O/p_$c | grep -v "^void at::\|distribution\|fillBuffer" 
done
rm -rf $O/p_*
