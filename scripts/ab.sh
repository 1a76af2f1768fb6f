#!/bin/bash
# In-box A/B of two builds of libpdmphip.so (run through gpurun): scripts/ab.sh scratch/libA.so scratch/libB.so [bench args]
# Alternates the two libraries three times on the SAME GPU (boxes of the pool differ by several %).
A=$1; B=$2; shift 2
cp pdmpflux.jl_amd/libpdmphip.so /tmp/lib_orig.so
trap 'cp /tmp/lib_orig.so pdmpflux.jl_amd/libpdmphip.so' EXIT
for rep in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then cp $A pdmpflux.jl_amd/libpdmphip.so; else cp $B pdmpflux.jl_amd/libpdmphip.so; fi
    r=$(timeout -k 10 200 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --pmc off "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])")
    echo "$v $r"
  done
done
cp /tmp/lib_orig.so pdmpflux.jl_amd/libpdmphip.so
