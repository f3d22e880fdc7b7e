#!/bin/bash
# second GPU pass of round 1: new wide weight-gradient kernel + gen-2 wave-specialised forward/dgrad kernel
set -u
out=gpurun_out/r1b; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q > $out/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?" | tee -a $out/summary.txt
tail -3 $out/pytest_kernels.log | tee -a $out/summary.txt
python tools/microbench_conv.py --iters 20 > $out/mb_default.jsonl 2>$out/mb_default.err
for cfg in 4 1 2; do python tools/microbench_conv.py --iters 20 --only resblock --tune 0 $cfg > $out/mb_res_cfg$cfg.jsonl 2>&1; done
python tools/microbench_conv.py --iters 20 --tune 4 1 > $out/mb_oldwgrad.jsonl 2>&1
for tb in 512 768 1536 2048; do python tools/microbench_conv.py --iters 20 --only resblock --tune 5 $tb > $out/mb_res_tb$tb.jsonl 2>&1; done
for f in $out/mb_*.jsonl; do echo "== $f"; python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print("%-34s fwd %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF | wgrad %7.1f us %5.1f TF" % (d['layer'], d['fwd_us'], d['fwd_TF'], d['dgrad_us'], d['dgrad_TF'], d['wgrad_us'], d['wgrad_TF']))
PY
done | tee -a $out/summary.txt
timeout 900 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/summary.txt
cat $out/bench.json | tee -a $out/summary.txt
