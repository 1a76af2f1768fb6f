#!/bin/bash
# Round-4 placement investigation, call 10: is the process-wide constant the physical offset the pool starts at (probe mode `spacer`)?
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 $P spacer 1024 > $O/c10_spacer1.jsonl 2>&1; echo "spacer1 rc=$?"
timeout -k 10 400 $P spacer 1024 > $O/c10_spacer2.jsonl 2>&1; echo "spacer2 rc=$?"
cat $O/c10_spacer1.jsonl $O/c10_spacer2.jsonl
