#!/bin/bash
# round 5, lost-store anomaly, batch 2: the stand-alone probe + trigger-kernel variants
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
V=tools/probes/_build
run() { echo "## $*"; env "$@" DIAG_RUNS=${DIAG_RUNS:-61} timeout 300 python tools/diag_lost_stores.py 2>&1 | grep -v "Warning\|Network\|^---" | tail -5; }
{
echo "## stand-alone probe (no torch)"
for n in 2 16; do
timeout 120 $V/side_queue_victim nemar_amd/lib/libnemar_hip.so 100 16 $n 1
done
timeout 120 $V/side_queue_victim nemar_amd/lib/libnemar_hip.so 100 16 2 0
run DIAG_LIB=$V/libnemar_hip_k7shfl.so
run DIAG_LIB=$V/libnemar_hip_k7lb1.so
run A=baseline
} 2>&1 | tee $O/lost_stores2.txt
