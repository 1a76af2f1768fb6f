#!/bin/bash
# Round-4 placement investigation, call 8: probe mode `pairs` over a pool of 2 1/2 sets (three sets as allocated, the worst one re-paired).
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 280 $P pairs 1024 160 16 > $O/c8_pairs1.jsonl 2>&1; echo "pairs1 rc=$?"
timeout -k 10 280 $P pairs 1024 160 16 > $O/c8_pairs2.jsonl 2>&1; echo "pairs2 rc=$?"
echo done
