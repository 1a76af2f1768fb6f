#!/bin/bash
# Round-4 placement investigation, call 17: PER-CHANNEL write-request counters (the raw TCC counters have 16 instances x 8 XCCs) for
# fast and slow hipMalloc placements of one process: is a slow placement a channel imbalance?
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  timeout -k 10 280 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL --output-format csv -d $O/c17_chan$i -- $P malloc 8 > $O/c17_chan$i.jsonl 2> $O/c17_chan$i.log; echo "chan$i rc=$?"
done
ls -la $O/c17_chan1/*/ | head; head -5 $O/c17_chan1/*/*counter_collection.csv; wc -l $O/c17_chan1/*/*counter_collection.csv; cat $O/c17_chan1.jsonl | cut -c1-160
