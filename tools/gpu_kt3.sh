#!/bin/bash
for t in 8=1 8=2 8=3 8=4 8=6 8=9; do echo "== $t"; bash tools/gpu_kt.sh kt_$t dgrad 10 $t | grep "igemm_kernel" | cut -c1-130; done
