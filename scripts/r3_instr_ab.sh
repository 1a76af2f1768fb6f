#!/bin/bash
# instruction counts per event (live PMC children of bench.py) of the round-2 tree and the working tree, same box
iv() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=r['roofline'].get('valu') or {}; print('%.2f ms  VALU/ev %.1f  SALU/ev %.1f  busy %.3f' % (r['roofline']['kernel_ms'], v.get('instr_per_event') or 0, v.get('salu_instr_per_event') or 0, v.get('busy') or 0))"; }
for args in "--sampler Boomerang --dim 100 --chains 50000 --refresh-rate 0.5" "--sampler BPS --target banana" "--chains 12500"; do
  echo "[$args]"
  (cd scratch/r2tree && python bench.py $args --cpu-seconds 0 --steps 4 --warmup 1 2>/dev/null | iv | sed 's/^/   r2  /')
  python bench.py $args --cpu-seconds 0 --steps 4 --warmup 1 2>/dev/null | iv | sed 's/^/   now /'
done
