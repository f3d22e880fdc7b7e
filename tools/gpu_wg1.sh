#!/bin/bash
# one-copy gy planes in the wide weight gradient (nemar_tune 34): kernel tests, stand-alone call time, step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "split16" 2>&1 | tail -3
echo '{"op": "wgrad", "C0": 256, "C1": 0, "K": 256, "R": 3, "stride": 1, "pad": 1, "pad_mode": 1, "H": 64, "W": 64, "N": 16, "act": 0, "count": 18}
{"op": "wgrad", "C0": 256, "C1": 0, "K": 512, "R": 4, "stride": 1, "pad": 1, "pad_mode": 0, "H": 32, "W": 32, "N": 24, "act": 0, "count": 1}' > /tmp/wg.jsonl
for t in "34=1" "34=0"; do echo "== $t"; python tools/microbench_trace.py /tmp/wg.jsonl $t 2>/dev/null; done
bash tools/gpu_ab.sh $1 "" "34=0" ""
