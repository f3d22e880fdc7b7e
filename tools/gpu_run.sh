#!/bin/bash
# One parameterised GPU runner (replaces the per-round tools/gpu_r*.sh one-offs).  Usage on the GPU box:
#   tools/gpu_run.sh <out-name> <step> [<step> ...]
# steps:
#   test:<pytest -k expression>     pytest -m gpu subset (expression "all" = whole gpu tier)
#   bench[:ENV=V,ENV=V]             python bench.py --no-cpu-baseline --no-extras under the given environment -> bench_<tag>.json
#   fullbench                       python bench.py (the driver's line, cpu baseline and extras included)
#   stats[:ENV=V,...]               rocprofv3 --kernel-trace --stats of 6 eager steps (NEMAR_SIDE_STREAM=0 unless given) -> kernel_stats_<tag>.csv
#   layers                          tools/microbench_conv.py per-layer table
#   py:<script.py args>             any tools/ script
#   pmc:<script.py args>            rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (SEPARATE passes, kernel trace only) over a tools/ script ->
#                                   pmc_<tag>.txt: per kernel, KiB per launch (FETCH_SIZE counts 64 B per 128-B request on gfx950: x 2 for bytes)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; shift
cd $R
for step in "$@"; do
  kind=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  tag=$(echo "$arg" | tr -c 'A-Za-z0-9_\n' '_'); [ -z "$tag" ] && tag=default
  envs=$(echo "$arg" | tr ',' ' ')
  case $kind in
    test)
      if [ "$arg" == "all" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.txt 2>&1
      else timeout 1200 python -m pytest tests -m gpu -x -q -k "$arg" > $O/pytest_$tag.txt 2>&1; fi
      tail -15 $O/pytest_*.txt | tail -25 ;;
    bench)
      env $envs timeout 900 python bench.py --no-cpu-baseline --no-extras --graph off > $O/bench_$tag.json 2> $O/bench_$tag.err
      python -c "
import json; d = json.load(open('$O/bench_$tag.json')); print('bench[%s]: %.2f img/s  %.2f ms/step  roofline %.3f (%.0f us)' % ('$arg', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_us', 0)))" || tail -5 $O/bench_$tag.err ;;
    fullbench)
      timeout 1500 python bench.py > $O/bench_full.json 2> $O/bench_full.err; cut -c1-600 $O/bench_full.json ;;
    stats)
      ( cd /tmp && export TMPDIR=/tmp && env NEMAR_SIDE_STREAM=0 $envs rocprofv3 --kernel-trace --stats -d $O/stats_$tag -- python $R/bench.py --steps 6 --warmup 2 --graph off --no-cpu-baseline --no-extras > $O/stats_line_$tag.json 2>/dev/null )
      python tools/prof_summary.py $O/stats_$tag $O/kernel_stats_$tag.csv > /dev/null 2>&1
      python - <<PY
import csv
rows = list(csv.DictReader(open('$O/kernel_stats_$tag.csv')))
tot = sum(float(r['total_us']) for r in rows); calls = sum(int(r['calls']) for r in rows)
print('stats[$arg]: kernels %.2f ms/step, %d launches/step (10 steps incl. the roofline pass)' % (tot / 1e4, calls // 10))
for r in rows[:28]:
    print('   %5.2f%% x%-5s avg %8.1f us  %s' % (float(r['pct']), r['calls'], float(r['avg_us']), r['name'][:100]))
PY
      rm -rf $O/stats_$tag ;;
    layers)
      python tools/microbench_conv.py --iters 20 > $O/layers.jsonl 2>/dev/null; tail -3 $O/layers.jsonl | cut -c1-300 ;;
    py)
      timeout 1500 python $arg > $O/py_$tag.txt 2>&1; tail -30 $O/py_$tag.txt ;;
    pmc)
      ( cd /tmp && export TMPDIR=/tmp
        for c in FETCH_SIZE WRITE_SIZE; do
          rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p_$c -- python $R/$arg > /dev/null 2>&1
          echo "=== $c (KiB per launch)"; python $R/tools/pmc_summary.py $O/p_$c | grep -v "distribution\|fillBuffer"
        done ) > $O/pmc_$tag.txt 2>&1
      rm -rf $O/p_FETCH_SIZE $O/p_WRITE_SIZE; grep -A1 "planes\|igemm\|wgrad_split\|sum_partials" $O/pmc_$tag.txt | head -60 ;;
  esac
done
