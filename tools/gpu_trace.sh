#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
python tools/trace_convs.py > $out/conv_trace.jsonl 2> $out/trace.err || tail -5 $out/trace.err
wc -l $out/conv_trace.jsonl
python tools/microbench_trace.py $out/conv_trace.jsonl > $out/conv_trace_times.txt 2>$out/mbt.err || tail -5 $out/mbt.err
cat $out/conv_trace_times.txt
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench: %.2f img/s  %.2f ms/step  roofline %.1f TF (%.0f us)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us']))"
