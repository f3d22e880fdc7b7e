#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
fmt='
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print("%-34s fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF" % (d["layer"], d["fwd_us"], d["fwd_TF"], d["dgrad_us"], d["dgrad_TF"]))
'
for t in "" "--tune 27 2" "--tune 28 1" "--tune 27 2 --tune 28 1" "--tune 27 1"; do
  echo "== $t" >> $O/mb.txt
  for only in "T.down1" "T.down2" "R.res 64" "R.up2" "D.l2" "T.resblock"; do
    timeout 300 python tools/microbench_conv.py --iters 20 --batch 16 $t --only "$only" 2>/dev/null | python -c "$fmt" >> $O/mb.txt
  done
done
cat $O/mb.txt
