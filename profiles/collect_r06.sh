#!/bin/bash
# Round-6 profiles on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect_r06.sh 4096        headline (radio_ml_conv.yaml, 16x16, batch 4096)
#   bash profiles/collect_r06.sh ref         BASELINE config 5 (radio_ml_conv_ref.yaml, int8 weights through the ABI, batch 4096)
#   bash profiles/collect_r06.sh plane128    the 128x128 argparse-default plane, batch 64 (PMC of the current k_lif_seq_c32t)
#   bash profiles/collect_r06.sh t1024       T = 1024 (the reference's n_iters default and script setting) at batch 512
# kernel-trace statistics and the PMC passes are SEPARATE rocprofv3 runs (one counter group per pass, never combined with a
# trace domain); the summaries to commit land under gpurun_out/r06prof_*/ and are copied to profiles/ by hand.
set -e
export TMPDIR=/tmp
MODE=${1:-4096}
if [ "$MODE" = "ref" ]; then
    OUT=$PWD/gpurun_out/r06prof_ref; ARGS="--network ref --batch 4096 --validate 0"; TSTEPS=2
elif [ "$MODE" = "t1024" ]; then
    OUT=$PWD/gpurun_out/r06prof_t1024; ARGS="--batch 512 --steps-T 1024 --cpu-windows 0 --per-step 0 --config5 0 --batch-sweep 0 --trained 0 --live-traffic 0 --t1024 0"; TSTEPS=2
elif [ "$MODE" = "plane128" ]; then
    OUT=$PWD/gpurun_out/r06prof_plane128; ARGS="--plane 128 --batch 64 --cpu-windows 0 --per-step 0 --config5 0 --batch-sweep 0 --trained 0 --live-traffic 0 --t1024 0"; TSTEPS=2
else
    OUT=$PWD/gpurun_out/r06prof_b$MODE; ARGS="--batch $MODE --cpu-windows 0 --per-step 0 --config5 0 --batch-sweep 0 --trained 0 --live-traffic 0 --t1024 0 --plane128 0"; TSTEPS=4
fi
mkdir -p "$OUT"
python3 bench.py $ARGS --steps 5 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 bench.py $ARGS --steps $TSTEPS --warmup 1 > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -o p -- python3 bench.py $ARGS --steps 1 --warmup 0 > /dev/null 2> "$OUT/pmc$i.err"
    echo "pmc pass $i done" >&2
done
python3 experiments/pmc_summary.py "$OUT/pmc.json" "$OUT/pmc1" "$OUT/pmc2" "$OUT/pmc3" "$OUT/pmc4" > "$OUT/pmc_summary.txt"
rm -rf "$OUT"/pmc[1-4] "$OUT/trace"
if [ "$MODE" = "4096" ]; then
    # the per-timestep protocol (net.test / net.learn at batch 512): kernel-trace statistics of 64 + 64 timesteps
    DCLL_GRAPH_LEARN=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_ps" -o t -- python3 experiments/per_step_timing.py 512 > "$OUT/per_step.txt" 2> "$OUT/trace_ps.err"
    find "$OUT/trace_ps" -name "*kernel_stats.csv" -exec cp {} "$OUT/learn_step_b512_kernel_stats.csv" \;
    rm -rf "$OUT/trace_ps"
fi
