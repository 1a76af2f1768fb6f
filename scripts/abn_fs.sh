#!/bin/bash
# like abn.sh for scripts/bench_from_skeleton.py
LIBS=$1; shift
cp pdmpflux.jl_amd/libpdmphip.so /tmp/lib_orig.so
trap 'cp /tmp/lib_orig.so pdmpflux.jl_amd/libpdmphip.so' EXIT
for rep in 1 2 3; do
  for L in $LIBS; do
    cp $L pdmpflux.jl_amd/libpdmphip.so
    r=$(timeout -k 10 200 python scripts/bench_from_skeleton.py "$@" 2>/dev/null | tail -1)
    echo "$(basename $L) $r" | cut -c1-300
  done
done
cp /tmp/lib_orig.so pdmpflux.jl_amd/libpdmphip.so
