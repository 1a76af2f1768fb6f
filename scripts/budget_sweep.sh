#!/bin/bash
# PDMP_CELL_BUDGET on the lane kernel's forms at the sizes they serve (through gpurun)
for C in 50000 100000; do for B in 3 5 9; do
  ms=$(PDMP_CELL_BUDGET=$B python bench.py --chains $C --steps 10 --warmup 2 --cpu-seconds 0 --pmc off 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  echo "LANEBUDGET chains $C budget $B: $ms ms"
done; done
for B in 3 5 9; do
  ms=$(PDMP_CELL_BUDGET=$B python bench.py --dim 100 --steps 6 --warmup 2 --cpu-seconds 0 --pmc off 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  echo "LANEBUDGET d100 budget $B: $ms ms"
done
for C in 12500 16384 25000; do ms=$(python bench.py --chains $C --steps 10 --warmup 2 --cpu-seconds 0 --pmc off 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.3f ms' % r['roofline']['kernel_ms'], r['roofline']['kernel'][:40])"); echo "DEFAULT chains $C: $ms"; done
