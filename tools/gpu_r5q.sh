#!/bin/bash
# round 5, boundary split: the product library in fresh processes vs the measurement build; kernel tier on the measurement build; default bench on the product
O=gpurun_out/r5q; mkdir -p $O
timeout 900 python -m pytest tests/test_product_lib_gpu.py -q -m gpu > $O/product.txt 2>&1; tail -5 $O/product.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_conv_real_shapes_gpu.py -q -m gpu -x > $O/kernels.txt 2>&1; tail -3 $O/kernels.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/r5q/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k: (v if 'error' in v else 'ok') for k, v in d.get('extras', {}).items() if isinstance(v, dict)})
print(d.get('extras', {}).get('exact_route'))
PY
