#!/bin/bash
# usage: scripts/r3_tests.sh <tag> <pytest args...>   -- runs pytest on the GPU box, log under gpurun_out/<tag>/tests.log
O=gpurun_out/$1; shift; mkdir -p $O
python -m pytest "$@" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $O/tests.log
grep -E "passed|failed|FAILED|ERROR" $O/tests.log | tail -15
exit $rc
