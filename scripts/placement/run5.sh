#!/bin/bash
# Round-4 placement investigation, call 5: is "slow" a relation between the X chunk and the V chunk written together, and can the
# allocator arrange the pairing?  (probe mode `pairs`: pair-time matrix, then the engine on the pairing as allocated / all pairs fast /
# all pairs slow, from one pool of 1-GiB chunks.)  Two processes.
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 280 $P pairs 1024 44 24 > $O/c5_pairs1.jsonl 2>&1; echo "pairs1 rc=$?"
timeout -k 10 280 $P pairs 1024 44 24 > $O/c5_pairs2.jsonl 2>&1; echo "pairs2 rc=$?"
echo done
