#!/bin/bash
# Round-4 placement investigation, call 4: is the speed a property of WHERE in the VRAM the memory lies (per-chunk map over all free
# memory, engine on the fastest / slowest / first / last chunks), and what do several sets alive at once look like (bench.py's protocol).
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 280 $P map 1024 > $O/c4_map1.jsonl 2>&1; echo "map1 rc=$?"
timeout -k 10 280 $P sets 1024 3 > $O/c4_sets1.jsonl 2>&1; echo "sets1 rc=$?"
timeout -k 10 280 $P map 1024 > $O/c4_map2.jsonl 2>&1; echo "map2 rc=$?"
timeout -k 10 280 $P sets 1024 3 > $O/c4_sets2.jsonl 2>&1; echo "sets2 rc=$?"
timeout -k 10 200 $P map 2048 > $O/c4_map_2g.jsonl 2>&1; echo "map 2g rc=$?"
echo done
