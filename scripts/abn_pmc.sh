#!/bin/bash
# like abn.sh, with the live PMC passes: prints ms per step and the HBM bytes written per launch
LIBS=$1; shift
cp pdmpflux.jl_amd/libpdmphip.so /tmp/lib_orig.so
trap 'cp /tmp/lib_orig.so pdmpflux.jl_amd/libpdmphip.so' EXIT
for rep in 1 2; do
  for L in $LIBS; do
    cp $L pdmpflux.jl_amd/libpdmphip.so
    r=$(timeout -k 10 400 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline'].get('traffic_write'), r['roofline']['valu'])")
    echo "$(basename $L) $r"
  done
done
cp /tmp/lib_orig.so pdmpflux.jl_amd/libpdmphip.so
