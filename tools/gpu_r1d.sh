#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -3
run() { echo "== $*"; python tools/microbench_conv.py --iters 20 --only resblock "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('fwd %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF | wgrad %7.1f us %5.1f TF' % (d['fwd_us'], d['fwd_TF'], d['dgrad_us'], d['dgrad_TF'], d['wgrad_us'], d['wgrad_TF']))
"; }
run
run --tune 4 2
run --tune 2 16
run --tune 2 32
run --tune 2 48
run --tune 2 2
grep '^{' tools/conv_trace_cfg2.jsonl > $out/conv_trace.jsonl 2>/dev/null || python tools/trace_convs.py 2>/dev/null | grep '^{' > $out/conv_trace.jsonl
python tools/microbench_trace.py $out/conv_trace.jsonl > $out/conv_trace_times.txt 2>$out/mbt.err || tail -5 $out/mbt.err
head -40 $out/conv_trace_times.txt
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench: %.2f img/s  %.2f ms/step  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
