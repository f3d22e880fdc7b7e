#!/bin/bash
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -2
for t in 8=1 8=2 8=3; do echo "== $t"; bash tools/gpu_kt.sh kt_$t dgrad 10 $t | grep "igemm_kernel" | cut -c1-130; done
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench: %.2f img/s  %.2f ms/step  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
