#!/bin/bash
# the stand-alone repro of the second side-stream difference under variant builds of the wide weight-gradient kernel
O=gpurun_out/r5s; mkdir -p $O
for v in ${VARIANTS:-"" wgdrain wgdeep wgprolog wgsync wgnoxcd}; do
  [ "$v" = product ] && v=""
  if [ -z "$v" ]; then lib=""; else lib=tools/probes/_build/libnemar_hip_$v.so; fi
  echo "== variant ${v:-product}"
  DIAG_OWN_ONLY=1 DIAG_LIB=$lib timeout 200 python tools/diag_wgrad_beside.py ${CALLS:-60000} 4 64 dgrad_dual 2>&1 | grep "co-runner\|last event\|Error"
done
