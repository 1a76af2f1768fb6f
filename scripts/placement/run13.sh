#!/bin/bash
# Round-4 placement investigation, call 13: the launch time over the virtual address of X, V and t across several 512-GiB windows (`vascan`).
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 $P vascan > $O/c13_vascan1.jsonl 2>&1; echo "vascan1 rc=$?"
timeout -k 10 400 $P vascan > $O/c13_vascan2.jsonl 2>&1; echo "vascan2 rc=$?"
tail -3 $O/c13_vascan1.jsonl
