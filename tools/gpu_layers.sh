#!/bin/bash
# per-layer convolution table of one bench step: every conv-family C-ABI call (shape, count) timed stand-alone, with its route
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout 600 python tools/trace_convs.py > $O/conv_trace.jsonl 2> $O/trace.err
timeout 900 python tools/microbench_trace.py $O/conv_trace.jsonl > $O/conv_layers.txt 2> $O/layers.err
head -70 $O/conv_layers.txt
