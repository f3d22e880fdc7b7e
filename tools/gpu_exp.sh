#!/bin/bash
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -x -q 2>&1 | tail -2
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench: %.2f img/s  %.2f ms/step  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
python tools/microbench_trace.py tools/conv_trace_cfg2.jsonl 12=1 2>/dev/null | grep -E " (32x32|16x16|8x8|4x4|2x2)$" | grep -v "K=512\|K=1 " | awk '{s+=$1} END {print "small layers with split:", s, "ms"}'
python tools/microbench_trace.py tools/conv_trace_cfg2.jsonl 12=0 2>/dev/null | grep -E " (32x32|16x16|8x8|4x4|2x2)$" | grep -v "K=512\|K=1 " | awk '{s+=$1} END {print "small layers without split:", s, "ms"}'
