#!/bin/bash
# are the OTHER LDS-DMA staged kernels hit the same way?  forward calls as the victim, claim off (measurement build, nemar_tune 37 = 0) and on
O=gpurun_out/r5x; mkdir -p $O
for v in wide_fwd s16g_fwd; do for m in 0 7; do
  echo "== victim $v, claim mask $m"
  NEMAR_TUNE=37=$m DIAG_VICTIM=$v timeout 200 python tools/diag_wgrad_beside.py ${CALLS:-60000} 4 64 dgrad_dual,agg_lds1k 2>&1 | grep "victim\|co-runner\|last event\|Error"
done; done 2>&1 | tee $O/other_victims.txt
