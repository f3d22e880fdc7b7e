#!/bin/bash
# split-bf16 kernel: what bounds the global->LDS copies?  chunk-order rotation, A-only / B-only copies, L2 counters
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "split_bf16" --tb=short 2>&1 | tail -3
fmt='
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print("%-34s fwd %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF" % (d["layer"], d["fwd_us"], d["fwd_TF"], d["dgrad_us"], d["dgrad_TF"]))
'
for t in "21 0" "21 1" "21 1 --tune 22 2" "21 1 --tune 22 4" "21 1 --tune 22 8" "21 1 --tune 22 16" "21 1 --tune 2 524288" "21 1 --tune 2 1048576" "21 1 --tune 22 16 --tune 15 0"; do
  echo "== tune $t"; timeout 200 python tools/microbench_conv.py --iters 30 --only T.resblock --arena --tune $t 2>/dev/null | python -c "$fmt"
done | tee $O/microbench.txt
cd /tmp && export TMPDIR=/tmp
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/tools/microbench_conv.py --iters 4 --only T.resblock --arena > /dev/null 2>&1
  python $R/tools/pmc_summary.py $O/pmc_$n igemm_bf6
done 2>&1 | tee $O/pmc.txt
rm -rf $O/pmc_*
cd $R
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_api_gpu.py -q --tb=short 2>&1 | tail -8
