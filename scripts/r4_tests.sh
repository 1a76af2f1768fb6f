#!/bin/bash
# the whole GPU suite in ONE process, log under gpurun_out/r4 (progress lines keep the call alive)
O=gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1150 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tee $O/tests_full.log | grep -E "passed|failed|error|^tests/.*(FAILED|ERROR)|%\]" | tail -40
echo "pytest rc=${PIPESTATUS[0]}"
