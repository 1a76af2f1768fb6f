#!/bin/bash
# rocprofv3 evidence for an arbitrary python command (profiles/collect.sh is the bench.py form): kernel trace + stats, then WRITE_SIZE and
# FETCH_SIZE in their own --pmc passes.  Usage: profiles/collect_cmd.sh <tag> <script.py> [args...]; summarise with profiles/summarise.py
set -e
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/bench_traced.json 2> $OUT/trace.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_wr -- python3 "$@" > /dev/null 2> $OUT/pmc_wr.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_rd -- python3 "$@" > /dev/null 2> $OUT/pmc_rd.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 "$@" > /dev/null 2> $OUT/pmc_sq.log
echo collected $OUT
