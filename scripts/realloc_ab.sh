#!/bin/bash
# scripts/realloc_series.py under several builds of the library, alternating (run through gpurun): scripts/realloc_ab.sh "libA.so libB.so" [chains]
LIBS=$1; shift
cp pdmpflux.jl_amd/libpdmphip.so /tmp/lib_orig.so
trap 'cp /tmp/lib_orig.so pdmpflux.jl_amd/libpdmphip.so' EXIT      # the shipped library comes back whatever happens
for rep in 1 2; do
  for L in $LIBS; do
    cp $L pdmpflux.jl_amd/libpdmphip.so
    echo "== $(basename $L)"
    timeout -k 10 200 python scripts/realloc_series.py "$@" 2>/dev/null | awk '{print $1, $2, $7, $8}'
  done
done
