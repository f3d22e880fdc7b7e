#!/bin/bash
timeout 1200 python -m pytest tests/test_step_gpu.py -x -q 2>&1 | tail -8
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench batched: %.2f img/s  %.2f ms/step  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
NEMAR_NO_BATCHED_PASSES=1 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench unbatched: %.2f img/s  %.2f ms/step  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
