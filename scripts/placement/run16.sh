#!/bin/bash
# Round-4 placement investigation, call 16: the re-mapping experiments again, now that a re-mapped address is made valid for kernels
# (a hipMalloc + hipFree after every mapping: probe.hip, flush_translations) -- stale2 as the check, then `pairs` (three sets as
# allocated, the worst one re-paired) and `map` (every chunk alone; engine on the fastest / slowest / first / last chunks).
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 100 $P stale2 > $O/c16_stale2.jsonl 2>&1; echo "stale2 rc=$?"; grep remap $O/c16_stale2.jsonl
timeout -k 10 400 $P pairs 1024 160 16 > $O/c16_pairs1.jsonl 2>&1; echo "pairs1 rc=$?"
timeout -k 10 400 $P pairs 1024 160 16 > $O/c16_pairs2.jsonl 2>&1; echo "pairs2 rc=$?"
echo done
