#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
fmt='
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print("%-34s wgrad %7.1f us %6.1f TF" % (d["layer"], d["wgrad_us"], d["wgrad_TF"]))
'
for dbg in 0 2 31; do
  echo "== dbg $dbg" >> $O/abl.txt
  for only in "T.resblock" "T.down1"; do
    timeout 300 python tools/microbench_conv.py --iters 20 --batch 8 --tune 2 $dbg --only "$only" 2>/dev/null | python -c "$fmt" >> $O/abl.txt
  done
done
cat $O/abl.txt
