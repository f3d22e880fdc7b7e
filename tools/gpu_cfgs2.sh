#!/bin/bash
# step time of the other BASELINE shapes (parity cases, not bench lines)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
pr() { python -c "
import json,sys
txt=sys.stdin.read().strip().splitlines()
d=json.loads([l for l in txt if l.startswith('{\"metric\"')][-1]); print('$1: %.2f ms/step %.1f img/s' % (d['ms_per_step'], d['value']))"; }
{
python bench.py --batch 1 --size 128 --steps 50 --warmup 10 --no-cpu-baseline --no-extras --opt=--stn_type --opt=affine --opt=--netG --opt=resnet_6blocks 2>/dev/null | pr "C1 (affine, resnet_6blocks, 128x128, batch 1)"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --opt=--multi_resolution --opt=2 2>/dev/null | pr "C3 (C2 + multi-resolution D, batch 8)"
python bench.py --batch 4 --size 512 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --opt=--stn_bilateral_alpha --opt=1.5 --opt=--stn_multires_reg --opt=2 2>/dev/null | pr "C4 (512x512, bilateral, multires reg, batch 4)"
python bench.py --batch 1 --size 1024 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --opt=--stn_cfg --opt=deep 2>/dev/null | pr "C5 (1024x1024, deep cfg, batch 1)"
} | tee $O/other_configs.txt
