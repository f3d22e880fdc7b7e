#!/bin/bash
# PMC passes over the T resblock shape on the general 16-bit-pipe kernels (no arena): fwd and wgrad
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
G1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
G2="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
G3="GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR"
run() {
  tag=$1; shift
  for g in 1 2 3; do
    eval GG=\$G$g
    NEMAR_ARENA=0 rocprofv3 --pmc $GG --kernel-trace --output-format csv -d $O/${tag}_g$g -- python $R/tools/pmc_conv.py "$@" > /dev/null 2>$O/${tag}_g$g.err
  done
  echo "=== $tag: $*"; for g in 1 2 3; do python $R/tools/pmc_summary.py $O/${tag}_g$g s16g; done
}
{
run fwd fwd 6
run wgrad wgrad 6
} > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
