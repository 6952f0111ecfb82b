#!/bin/bash
# Round-2 profiles on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect_r02.sh B        B = batch per GPU (4096 = headline, 8192 = BASELINE config 3)
# kernel-trace statistics and the PMC passes are separate rocprofv3 runs (one counter group per pass); the summaries that
# get committed are written under gpurun_out/r02prof_b$B/ and copied to profiles/ by hand.
set -e
if [ "$1" = "ref" ]; then
    OUT=$PWD/gpurun_out/r02prof_ref
    mkdir -p "$OUT"
    export TMPDIR=/tmp
    python3 bench.py --network ref --steps 3 --warmup 1 --batch 4096 > "$OUT/bench.json" 2> "$OUT/bench.err"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 bench.py --network ref --steps 2 --warmup 1 --batch 4096 > /dev/null 2> "$OUT/trace.err"
    find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
    rm -rf "$OUT/trace"
    exit 0
fi
B=${1:-4096}
OUT=$PWD/gpurun_out/r02prof_b$B
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py --steps 10 --warmup 2 --batch $B > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 bench.py --steps 4 --warmup 1 --batch $B --cpu-windows 0 --per-step 0 > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -o p -- python3 bench.py --steps 1 --warmup 0 --batch $B --cpu-windows 0 --per-step 0 > /dev/null 2> "$OUT/pmc$i.err"
    echo "pmc pass $i done" >&2
done
python3 experiments/pmc_summary.py "$OUT/pmc.json" "$OUT/pmc1" "$OUT/pmc2" "$OUT/pmc3" "$OUT/pmc4" > "$OUT/pmc_summary.txt"
rm -rf "$OUT"/pmc[1-4] "$OUT/trace"

# BASELINE config 5 (radio_ml_conv_ref.yaml): bash profiles/collect_r02.sh ref   -> gpurun_out/r02prof_ref/
