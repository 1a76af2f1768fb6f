#!/bin/bash
# rocprofv3 evidence for the register forms added at the end of round 4 (StickyZigZag, SpeedUpZigZag) and the dense time horizon
set -e
for spec in "r4_sticky_banana_reg --sampler StickyZigZagAD --target banana --n-sk 500" \
            "r4_suzz_banana_reg --sampler SpeedUpZigZagAD --target banana --n-sk 500" \
            "r4_suzz_iso_reg --sampler SpeedUpZigZagAD --target iso --n-sk 500"; do
  set -- $spec; tag=$1; shift
  bash profiles/collect.sh $tag "$@" --placement-survey off --alloc hipmalloc
done
