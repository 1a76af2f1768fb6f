#!/bin/bash
# In-box comparison of several builds of libpdmphip.so (run through gpurun): scripts/abn.sh "libA.so libB.so ..." [bench args]
# Alternates the libraries three times on the SAME GPU (boxes of the pool differ by several %).
LIBS=$1; shift
cp pdmpflux.jl_amd/libpdmphip.so /tmp/lib_orig.so
trap 'cp /tmp/lib_orig.so pdmpflux.jl_amd/libpdmphip.so' EXIT
for rep in 1 2 3; do
  for L in $LIBS; do
    cp $L pdmpflux.jl_amd/libpdmphip.so
    r=$(timeout -k 10 200 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --pmc off "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])")
    echo "$(basename $L) $r"
  done
done
cp /tmp/lib_orig.so pdmpflux.jl_amd/libpdmphip.so
