#!/bin/bash
# A second build of libdcll_hip.so with compile-time switches, for A/B runs in ONE session (DCLL_HIP_SO=<path> python ...):
#   bash experiments/build_variant.sh split2 dcll_hip.hip -DWG32_SPLIT2_MAX_BATCH=512
#   bash experiments/build_variant.sh epi0   dcll_seq_w3.hip -DW3_EPI=0
# -> experiments/_variants/libdcll_hip_<name>.so (git-ignored; travels to the GPU box with the snapshot).  Only the named
# translation unit is recompiled; the others come from the product's csrc/_build (run `make -C csrc` first).
set -e
NAME=$1; TU=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/snn_modulation_classification_amd/csrc
OUT=$ROOT/experiments/_variants
mkdir -p "$OUT"
make -s -C "$CSRC"
EXTRA=""
[ "$TU" = "dcll_seq_tiled.hip" ] && EXTRA="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function $EXTRA "$@" \
    -c -o "$OUT/${TU%.hip}_$NAME.o" "$CSRC/$TU"
OBJS=""
for f in dcll_hip dcll_seq_tiled dcll_readout dcll_learn dcll_seq_w3 dcll_dense; do
    if [ "$f.hip" = "$TU" ]; then OBJS="$OBJS $OUT/${f}_$NAME.o"; else OBJS="$OBJS $CSRC/_build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libdcll_hip_$NAME.so" $OBJS
echo "$OUT/libdcll_hip_$NAME.so"
