#!/bin/bash
# usage: scripts/check_lane_asm.sh [TU ...]  -- device assembly of the one-wave-per-SIMD kernels (the only ones that hold values in AGPRs),
# checked for the allocator's AGPR copies in front of an exec restore (scripts/check_exec_prologue.py)
cd "$(dirname "$0")/../pdmpflux.jl_amd/csrc" || exit 1
TUS=${@:-inst_lane_zigzag inst_lane_zigzag_a inst_lane_zigzag_b inst_tape_lane inst_lane2_zigzag inst_lane2_zigzag_a inst_lane2_zigzag_b}
mkdir -p /tmp/laneasm
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DLANE_WAVES=1 --cuda-device-only -S"
echo $TUS | tr ' ' '\n' | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc $FL {}.hip -o /tmp/laneasm/{}.s 2>/dev/null"
rc=0
for t in $TUS; do python3 ../../scripts/check_exec_prologue.py /tmp/laneasm/$t.s || rc=1; done
exit $rc
