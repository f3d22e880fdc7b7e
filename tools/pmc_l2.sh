R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_l2; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d $O/a -- python $R/tools/pmc_conv.py fwd 6 > /dev/null 2>$O/a.err
python $R/tools/pmc_summary.py $O/a igemm; tail -3 $O/a.err
