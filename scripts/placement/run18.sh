#!/bin/bash
# call 18: the same per-dispatch counters in rocprofv3's JSON output, which keeps the counter dimensions (16 TCC instances x 8 XCCs)
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 280 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_DRAM_CREDIT_STALL --output-format json -d $O/c18_chan -- $P malloc 8 > $O/c18_chan.jsonl 2> $O/c18_chan.log; echo "rc=$?"
ls -la $O/c18_chan/*/
python3 - <<'P'
import json,glob
f=glob.glob('gpurun_out/r4/c18_chan/*/*results.json')[0]
d=json.load(open(f))
r=d['rocprofiler-sdk-tool'][0]
print(list(r.keys()))
print(list(r['callback_records'].keys()) if 'callback_records' in r else None)
cc=r['callback_records']['counter_collection']
print(len(cc)); print(json.dumps(cc[5])[:1500])
print(json.dumps(r.get('counters',[])[:2])[:1500])
P
