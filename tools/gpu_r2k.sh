#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_warp_$c -- python $R/tools/microbench.py --iters 3 --size 256 --sigma0 > /dev/null 2>&1
done
{ for c in FETCH_SIZE WRITE_SIZE; do echo "=== warp 8x3x256x256 sigma=0 $c"; python $R/tools/pmc_summary.py $O/pmc_warp_$c grid_sample; python $R/tools/pmc_summary.py $O/pmc_warp_$c far_; done; } > $O/pmc_warp.txt 2>&1
cat $O/pmc_warp.txt
rm -rf $O/pmc_warp_*
cd $R
echo "== C1 (affine, resnet_6blocks, 128x128, batch 1)"
python bench.py --batch 1 --size 128 --steps 50 --warmup 10 --no-cpu-baseline --no-extras --opt=--stn_type --opt=affine --opt=--netG --opt=resnet_6blocks 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C1: %.2f ms/step %.1f img/s' % (d['ms_per_step'], d['value']))"
echo "== C3"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --opt=--multi_resolution --opt=2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C3: %.2f ms/step %.1f img/s' % (d['ms_per_step'], d['value']))"
echo "== C4 (512, bilateral, batch 4)"; python bench.py --batch 4 --size 512 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --opt=--stn_bilateral_alpha --opt=1.5 --opt=--stn_multires_reg --opt=2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C4: %.2f ms/step %.1f img/s' % (d['ms_per_step'], d['value']))"
echo "== C5 (1024, deep cfg, batch 1)"; python bench.py --batch 1 --size 1024 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --opt=--stn_cfg --opt=deep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C5: %.2f ms/step %.1f img/s' % (d['ms_per_step'], d['value']))"
