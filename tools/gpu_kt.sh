#!/bin/bash
# per-kernel durations of one conv op on the resblock shape: gpu_kt.sh TAG [fwd|dgrad|wgrad] [key=value...]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/tools/pmc_conv.py "$@" > /dev/null 2>&1
cd $R; python tools/prof_summary.py $O/kt $O/kt.csv > /dev/null 2>&1; cut -c1-140 $O/kt.csv | head -8
