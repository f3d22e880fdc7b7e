#!/bin/bash
# round 5, lost-store anomaly: one line of statistics per experimental condition (tools/diag_lost_stores.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
V=tools/probes/_build
run() { echo "## $*"; env "$@" DIAG_RUNS=${DIAG_RUNS:-61} timeout 300 python tools/diag_lost_stores.py 2>&1 | grep -v Warning | tail -5; }
{
run A=baseline
run DIAG_SENTINEL=1
run DIAG_LIB=$V/libnemar_hip_fence.so
run DIAG_LIB=$V/libnemar_hip_swap.so
run DIAG_LIB=$V/libnemar_hip_nt.so
run DIAG_LIB=$V/libnemar_hip_k7env.so K7_SKIP_SMAX=1
run DIAG_LIB=$V/libnemar_hip_k7env.so K7_SKIP_MAIN=1
run DIAG_LIB=$V/libnemar_hip_k7env.so K7_SKIP_SUMS=1
run DIAG_LIB=$V/libnemar_hip_k7env.so K7_SKIP_SMAX=1 K7_SKIP_MAIN=1 K7_SKIP_SUMS=1
run HIP_FORCE_DEV_KERNARG=0
run GPU_MAX_HW_QUEUES=8
run HSA_ENABLE_SDMA=0
run NEMAR_SIDE_K7=0
} 2>&1 | tee $O/lost_stores.txt
