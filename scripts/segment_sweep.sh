#!/bin/bash
# (measured at the end of round 3: the defaults are within 1 % of the best) segments per chain (PDMP_FORCE_SEGMENTS) on the VALU-bound configs at full size (through gpurun)
for S in default 2 4 8 16 32; do
  if [ $S = default ]; then unset PDMP_FORCE_SEGMENTS; else export PDMP_FORCE_SEGMENTS=$S; fi
  ms=$(python bench.py --sampler BPS --target banana --steps 4 --warmup 1 --cpu-seconds 0 --pmc off 2>/dev/null | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  echo "SEG BPS banana segments $S: $ms ms"
done
for S in default 2 4 8 16 32; do
  if [ $S = default ]; then unset PDMP_FORCE_SEGMENTS; else export PDMP_FORCE_SEGMENTS=$S; fi
  ms=$(python bench.py --sampler Boomerang --dim 100 --chains 50000 --refresh-rate 0.5 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off 2>/dev/null | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  echo "SEG Boomerang segments $S: $ms ms"
done
