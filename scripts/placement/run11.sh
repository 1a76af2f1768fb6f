#!/bin/bash
# Round-4 placement investigation, call 11: is the process-wide constant the VIRTUAL distance between X and V (probe mode `vadelta`)?
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 $P vadelta > $O/c11_vadelta1.jsonl 2>&1; echo "vadelta1 rc=$?"
timeout -k 10 300 $P vadelta > $O/c11_vadelta2.jsonl 2>&1; echo "vadelta2 rc=$?"
cat $O/c11_vadelta1.jsonl $O/c11_vadelta2.jsonl
