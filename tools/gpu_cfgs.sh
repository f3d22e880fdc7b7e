#!/bin/bash
# robustness: the other BASELINE config shapes at per-GPU size (C3 multi-resolution D, C4 512^2 bilateral, C5 1024^2, C1 affine 128^2)
pr() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1: %.2f img/s  %.2f ms/step  finite=%s' % (d['value'], d['ms_per_step'], d['losses_finite']))"; }
timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --batch 8 --opt=--multi_resolution --opt=2 2>&1 | tail -3 | pr C3
timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --batch 4 --size 512 --opt=--stn_bilateral_alpha --opt=1.5 2>&1 | tail -3 | pr C4
timeout 900 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --batch 1 --size 1024 2>&1 | tail -3 | pr C5
timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --batch 1 --size 128 --opt=--stn_type --opt=affine --opt=--netG --opt=resnet_6blocks 2>&1 | tail -3 | pr C1
