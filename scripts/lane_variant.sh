#!/bin/bash
# Builds a variant of libpdmphip.so that differs in the lane kernel's compile flags only (for scripts/ab.sh):
#   scripts/lane_variant.sh pdmpflux.jl_amd/csrc/abtmp/libB.so -DLANE_STORE_AUX=2
set -e
OUT=$1; shift
cd "$(dirname "$0")/../pdmpflux.jl_amd/csrc"
mkdir -p abtmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -DLANE_WAVES=1 "$@" -Rpass-analysis=kernel-resource-usage -c inst_lane_zigzag.hip -o abtmp/lane_variant.o 2> abtmp/lane_variant.remarks
grep -E "VGPRs Spill: [1-9]" abtmp/lane_variant.remarks && { echo "variant spills"; [ -n "$ALLOW_SPILL" ] || exit 1; }
OBJS=$(ls *.o | grep -v inst_lane_zigzag.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "../../$OUT" $OBJS abtmp/lane_variant.o -L/opt/rocm/lib -lrocblas -lhiprtc -ldl
echo built $OUT
