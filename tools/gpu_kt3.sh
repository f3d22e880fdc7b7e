#!/bin/bash
for t in 2=0 2=1 2=2 2=64 2=65 2=67; do echo "== $t"; bash tools/gpu_kt.sh kt_$t dgrad 10 $t | grep "igemm_kernel" | cut -c1-130; done
