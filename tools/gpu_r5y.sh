#!/bin/bash
O=gpurun_out/r5y; mkdir -p $O
for v in wgnoclaim wgrot; do for r in 1 2 3 4; do
  echo "== variant $v run $r"
  DIAG_OWN_ONLY=1 DIAG_LIB=tools/probes/_build/libnemar_hip_$v.so timeout 200 python tools/diag_wgrad_beside.py 16000 4 64 dgrad_dual 2>&1 | grep "co-runner\|last event\|Error"
done; done 2>&1 | tee $O/rot.txt
