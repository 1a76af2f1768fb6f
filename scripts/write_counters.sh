#!/bin/bash
# Second-counter check of the bytes the headline launch writes (verdict round 2, weakness 4: "the nt over-count is asserted, not shown"):
# TCC_EA0_WRREQ (all write requests to the fabric), TCC_EA0_WRREQ_64B (the 64-byte ones), WRITE_SIZE -- for the shipped library (streaming
# stores, buffer_store ... nt) and for a build with plain stores (-DLANE_STORE_AUX=0).  usage (through gpurun): scripts/write_counters.sh "libA.so libB.so"
LIBS=$1
cp pdmpflux.jl_amd/libpdmphip.so /tmp/lib_orig.so
trap 'cp /tmp/lib_orig.so pdmpflux.jl_amd/libpdmphip.so' EXIT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for L in $LIBS; do
  cp $L pdmpflux.jl_amd/libpdmphip.so
  n=$(basename $L .so)
  O=gpurun_out/wrc_$n; rm -rf $O; mkdir -p $O
  rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/a -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --pmc off > /dev/null 2> $O/a.log || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/b -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --pmc off > /dev/null 2> $O/b.log || exit 1
  rocprofv3 --pmc TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WR_UNCACHED_32B_sum --output-format csv -d $O/c -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --pmc off > /dev/null 2> $O/c.log || true
  python3 - "$O" "$n" <<'PY'
import csv,glob,sys,collections
O,n=sys.argv[1],sys.argv[2]
agg=collections.defaultdict(float)
for f in glob.glob(O+"/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "lane_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
w,w64=agg.get("TCC_EA0_WRREQ_sum",0),agg.get("TCC_EA0_WRREQ_64B_sum",0)
print(n, dict(agg))
print(n, "bytes by request size = 64*W64 + 32*(W-W64) = %.2f GB; WRITE_SIZE = %.2f GB; requests %.3e of which 64-byte %.3e"%((64*w64+32*(w-w64))/1e9, agg.get("WRITE_SIZE",0)*1024/1e9, w, w64))
PY
done
