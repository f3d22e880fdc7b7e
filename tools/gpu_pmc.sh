#!/bin/bash
# PMC passes over one conv shape (T resblock) for the igemm / wgrad kernels; counters grouped to fit the SQ slots.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
G1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
G2="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
run() { # tag which tune...
  tag=$1; shift
  rocprofv3 --pmc $G1 --kernel-trace --output-format csv -d $O/${tag}_g1 -- python $R/tools/pmc_conv.py "$@" > /dev/null 2>$O/${tag}_g1.err
  rocprofv3 --pmc $G2 --kernel-trace --output-format csv -d $O/${tag}_g2 -- python $R/tools/pmc_conv.py "$@" > /dev/null 2>$O/${tag}_g2.err
  echo "=== $tag: $*"; python $R/tools/pmc_summary.py $O/${tag}_g1; python $R/tools/pmc_summary.py $O/${tag}_g2; tail -2 $O/${tag}_g1.err $O/${tag}_g2.err | grep -i "error\|invalid\|not found" 
}
rocprofv3 -L 2>/dev/null | grep -o "\b\(SQ\|GRBM\|TCC\|TCP\|TA\|TD\)_[A-Z0-9_]*" | sort -u > $O/counters.txt
{
run fwd_ws2 fwd 6 0=0 | grep -v "^$"
run fwd_ws1 fwd 6 0=4
run fwd_4w fwd 6 0=1
run wgrad2 wgrad 6
run wgrad_old wgrad 6 4=1
run dgrad dgrad 6
} > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
