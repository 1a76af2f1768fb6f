#!/bin/bash
# In-box A/B of one build under two environments (run through gpurun): scripts/ab_env.sh "VAR=1" [bench args]
# Alternates default / VAR three times on the SAME GPU (boxes of the pool differ by several %).
E=$1; shift
for rep in 1 2 3; do
  r=$(timeout -k 10 200 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --pmc off "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])")
  echo "default $r"
  r=$(env $E timeout -k 10 200 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --pmc off "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])")
  echo "$E $r"
done
