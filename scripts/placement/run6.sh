#!/bin/bash
# Round-4 placement investigation, call 6: does the VIRTUAL address of a chunked mapping matter (probe mode `valign`)?  Two processes.
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 280 $P valign 1024 > $O/c6_valign1.jsonl 2>&1; echo "valign1 rc=$?"
timeout -k 10 280 $P valign 1024 > $O/c6_valign2.jsonl 2>&1; echo "valign2 rc=$?"
cat $O/c6_valign1.jsonl $O/c6_valign2.jsonl
