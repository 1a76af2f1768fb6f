#!/bin/bash
# Round-4 placement investigation, call 9: which allocation decides the launch time -- t, X / V, or the engine's own buffers (probe mode `tswap`).
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 280 $P tswap 1024 > $O/c9_tswap1.jsonl 2>&1; echo "tswap1 rc=$?"
timeout -k 10 280 $P tswap 1024 > $O/c9_tswap2.jsonl 2>&1; echo "tswap2 rc=$?"
cat $O/c9_tswap1.jsonl $O/c9_tswap2.jsonl
