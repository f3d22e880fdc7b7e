#!/bin/bash
# instruction mix / wait counters of s16g_kernel on T.down1 forward (batch 16), full and under ablation (nemar_tune(2, bits))
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
GA="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_SMEM"
GB="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_WAIT_INST_LDS"
for t in "2=0"; do
  for g in "$GA" "$GB"; do
    n=${g// /_}; n=${n:16:24}
    NEMAR_PMC_SHAPE=${2:-16,64,128,256,3,2,1,0} rocprofv3 --pmc $g --kernel-trace --output-format csv -d $O/p_${t//=/_}_$n -- python $R/tools/pmc_conv.py fwd 4 $t > /dev/null 2>&1
    echo "=== tune $t"; python $R/tools/pmc_summary.py $O/p_${t//=/_}_$n s16g_kernel
  done
done
rm -rf $O/p_*
