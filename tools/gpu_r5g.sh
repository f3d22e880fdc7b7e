#!/bin/bash
# round 5: the library without packed-FP32 instructions — the anomaly must be gone with EVERY weight gradient on the side queue; speed A/B on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
V=tools/probes/_build
run() { echo "## $*"; env "$@" DIAG_RUNS=${DIAG_RUNS:-101} timeout 300 python tools/diag_lost_stores.py 2>&1 | grep -v "Warning\|Network\|^---\|was created\|initialize" | tail -5; }
{
echo "## probe, library victim: new library, then the round-4 build"
timeout 120 $V/side_queue_victim nemar_amd/lib/libnemar_hip.so 100 16 16 1
timeout 120 $V/side_queue_victim $V/libnemar_hip_pk.so 100 16 16 1
run A=new_lib
run DIAG_LIB=$V/libnemar_hip_pk.so
bs() { python -c "
import json,sys; d = json.load(open('$1')); print('$2  %.2f img/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], d.get('launch','')[:50]))"; }
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extras --graph off > $O/b_new.json 2>$O/b_new.err; bs $O/b_new.json "new lib           "
DIAG_LIB=$V/libnemar_hip_pk.so python tools/bench_with_lib.py --no-cpu-baseline --no-extras --graph off > $O/b_old.json 2>$O/b_old.err; bs $O/b_old.json "round-4 flags     "
NEMAR_SIDE_MODE=all python bench.py --no-cpu-baseline --no-extras --graph off > $O/b_all.json 2>$O/b_all.err; bs $O/b_all.json "new lib, 7x7 on side"
done
} 2>&1 | tee $O/summary.txt
