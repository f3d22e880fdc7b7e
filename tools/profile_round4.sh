#!/bin/bash
# Round-4 evidence run (on the GPU box, via gpurun): the default bench line (with its extras), rocprofv3 kernel stats of the same
# command, SQ + traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate passes, no trace domains next to --pmc) over
#   * the dominant kernel at the step's own batch (16): igemm_split16_kernel<2,2,3> forward
#   * the 7x7 stem (3 -> 64) and head (64 -> 3) at 256x256, batch 16: forward, data gradient, weight gradient (conv_k7.hip)
#   * grid_sample forward / backward (tools/microbench.py shapes)
# and the per-layer convolution table of one step.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
python $R/tools/prof_summary.py $O/stats $O/kernel_stats.csv > /dev/null 2>&1
G1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
pmc() {  # tag shape which
  for c in "$G1" FETCH_SIZE WRITE_SIZE; do
    t=${c%% *}
    NEMAR_PMC_SHAPE=$2 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$1_$3_$t -- python $R/tools/pmc_conv.py $3 4 > /dev/null 2>&1
    echo "=== $1 $3 ($2) $t"; python $R/tools/pmc_summary.py $O/pmc_$1_$3_$t
  done
}
{
pmc resblock16 16,256,256,64,3,1,1,1 fwd
for which in fwd dgrad wgrad; do pmc stem 16,3,64,256,7,1,3,1 $which; done
for which in fwd dgrad wgrad; do pmc head 16,64,3,256,7,1,3,1 $which; done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_warp_$c -- python $R/tools/microbench.py --iters 3 > /dev/null 2>&1
  echo "=== warp $c"; python $R/tools/pmc_summary.py $O/pmc_warp_$c grid_sample; python $R/tools/pmc_summary.py $O/pmc_warp_$c far_
done
} > $O/pmc_summary.txt 2>&1
python $R/tools/microbench.py > $O/microbench.jsonl 2>/dev/null
cd $R
timeout 600 python tools/trace_convs.py > $O/conv_trace.jsonl 2> $O/trace.err
timeout 900 python tools/microbench_trace.py $O/conv_trace.jsonl > $O/conv_layers.txt 2> $O/layers.err
timeout 300 python tools/overlap_timeline.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|c10d_logger\|return func" > $O/overlap_timeline.txt
{ echo '# tools/timeline_s16g.py 64 128 256 3 2 16 (T.down1 forward, batch 16, NEMAR_TIMELINE build): s_memtime cycles per 16-channel chunk'; NEMAR_TL_LIB=$R/nemar_amd/lib/libnemar_hip_tl.so timeout 120 python tools/timeline_s16g.py 64 128 256 3 2 16 2>&1 | grep -v amdgpu.ids; echo '# tools/timeline_s16g.py 32 32 256 3 1 8 (R.res 32->32)'; NEMAR_TL_LIB=$R/nemar_amd/lib/libnemar_hip_tl.so timeout 120 python tools/timeline_s16g.py 32 32 256 3 1 8 2>&1 | grep -v amdgpu.ids; } > $O/s16g_timeline.txt 2>&1
bash tools/gpu_pmc_s16g4.sh $1_sq 2>&1 | grep -v "^\[" > $O/pmc_s16g_sq.txt
timeout 200 python tools/overlap_probe.py 2>&1 | grep -v amdgpu.ids > $O/overlap_probe.txt
tools/probes/_build/load_width > $O/load_width_raw.txt 2>&1
cat $O/bench.json
rm -rf $O/stats/*/*.db $O/pmc_resblock16_* $O/pmc_stem_* $O/pmc_head_* $O/pmc_warp_*
