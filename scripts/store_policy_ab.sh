#!/bin/bash
# Cache-policy bits of the lane kernel's event stores (LANE_STORE_AUX: sc0 = 1, nt = 2, sc1 = 16): time per step and write requests to the
# fabric of each build, alternating on one box.  usage (through gpurun): scripts/store_policy_ab.sh "libA.so libB.so ..."
LIBS=$1
cp pdmpflux.jl_amd/libpdmphip.so /tmp/lib_orig.so
trap 'cp /tmp/lib_orig.so pdmpflux.jl_amd/libpdmphip.so' EXIT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for L in $LIBS; do
    cp $L pdmpflux.jl_amd/libpdmphip.so
    r=$(timeout -k 10 200 python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --pmc off 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['roofline']['kernel_ms'])")
    echo "time $(basename $L .so) $r"
  done
done
for L in $LIBS; do
  cp $L pdmpflux.jl_amd/libpdmphip.so
  n=$(basename $L .so); O=gpurun_out/spa_$n; rm -rf $O; mkdir -p $O
  rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/a -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --pmc off > /dev/null 2> $O/a.log || continue
  python3 - "$O" "$n" <<'PY'
import csv,glob,sys,collections
O,n=sys.argv[1],sys.argv[2]
agg=collections.defaultdict(float)
for f in glob.glob(O+"/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "lane_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
w,w64=agg.get("TCC_EA0_WRREQ_sum",0),agg.get("TCC_EA0_WRREQ_64B_sum",0)
print("writes", n, "%.2f GB in %.3e requests (%.3e of 64 bytes)"%((64*w64+32*(w-w64))/1e9, w, w64))
PY
done
