#!/bin/bash
# step as a captured hipGraph vs eager launches: BASELINE config 1 (launch-bound), config 2 (GPU-bound) and a batch-1 256x256 run
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
pr() { python -c "
import json,sys
txt=sys.stdin.read().strip().splitlines()
d=json.loads([l for l in txt if l.startswith('{\"metric\"')][-1]); print('$1: %.2f ms/step %.1f img/s' % (d['ms_per_step'], d['value']))"; }
{
for g in "--graph off" "--graph on"; do
python bench.py --batch 1 --size 128 --steps 100 --warmup 10 --no-cpu-baseline --no-extras $g --opt=--stn_type --opt=affine --opt=--netG --opt=resnet_6blocks 2>$O/err1.txt | pr "C1 (affine, resnet_6blocks, 128x128, batch 1) $g"
python bench.py --batch 1 --steps 50 --warmup 5 --no-cpu-baseline --no-extras $g 2>$O/err2.txt | pr "C2 shape at batch 1 $g"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $g 2>$O/err3.txt | pr "C2 (batch 8) $g"
done
} | tee $O/graph_vs_eager.txt
tail -3 $O/err1.txt
