#!/bin/bash
# Round-4 placement investigation, call 12: the launch time over the virtual placement of X and V (probe mode `vamap`).
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 $P vamap > $O/c12_vamap1.jsonl 2>&1; echo "vamap1 rc=$?"
timeout -k 10 400 $P vamap > $O/c12_vamap2.jsonl 2>&1; echo "vamap2 rc=$?"
tail -3 $O/c12_vamap1.jsonl
