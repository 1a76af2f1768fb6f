#!/bin/bash
# occupancy probe of the pair form: 128 / 256 / 384 / 512 blocks of four waves (through gpurun)
for C in 16384 32768 49152 65536; do
  ms=$(PDMP_LANE=2 python bench.py --chains $C --steps 8 --warmup 2 --cpu-seconds 0 --pmc off 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  echo "OCC pair chains $C: $ms ms"
done
for C in 32768 65536; do
  ms=$(PDMP_LANE=1 python bench.py --chains $C --steps 8 --warmup 2 --cpu-seconds 0 --pmc off 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  echo "OCC lane1 chains $C: $ms ms"
done
