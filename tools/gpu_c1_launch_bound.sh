#!/bin/bash
# BASELINE config 1 (affine STN, resnet_6blocks, 128x128, batch 1): wall time per step vs the sum of its kernels, eager and as a hipGraph
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
for g in on off; do
  mkdir -p $O/$g
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$g/stats -- python $R/bench.py --batch 1 --size 128 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --graph $g --opt=--stn_type --opt=affine --opt=--netG --opt=resnet_6blocks > $O/$g/bench.json 2>/dev/null
  cd $R; python tools/prof_summary.py $O/$g/stats $O/$g/kernel_stats.csv > /dev/null 2>&1
  python - <<PY
import csv, json
rows = list(csv.DictReader(open("$O/$g/kernel_stats.csv")))
d = json.loads([l for l in open("$O/$g/bench.json") if l.startswith('{"metric"')][-1])
steps = 50 + 5 + 2 + (3 if "$g" == "on" else 0)
tot = sum(float(r["total_us"]) for r in rows); n = sum(int(r["calls"]) for r in rows)
print("C1 --graph $g: wall %.2f ms/step (under rocprofv3 --kernel-trace); kernels %.2f ms/step in %.0f launches/step (%.1f us average)" % (d["ms_per_step"], tot / steps / 1e3, n / steps, tot / n))
PY
done | tee $O/c1.txt
