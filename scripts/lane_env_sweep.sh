#!/bin/bash
# ms per step of bench.py under a list of environment settings (run through gpurun): scripts/lane_env_sweep.sh "A=1 B=2" "A=3" -- [bench args]
SETS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do SETS+=("$1"); shift; done
shift
for rep in 1 2; do
  for S in "${SETS[@]}"; do
    r=$(env $S timeout -k 10 200 python bench.py --cpu-seconds 0 --pmc off --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])")
    echo "$S: $r"
  done
done
