#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's default workload on the GPU box (run via gpurun):
#   pass 1: --kernel-trace --stats           (per-kernel durations; must agree with bench.py's hipEvent timing)
#   pass 2: --pmc VALU / wave counters       (own run, no trace domains combined with --pmc)
#   pass 3: --pmc WRITE_SIZE   pass 4: --pmc FETCH_SIZE   (TCC slots do not fit in one pass)
# Usage: profiles/collect.sh <tag> [bench args...]; writes gpurun_out/prof_<tag>/ ; summarise with profiles/summarise.py
set -e
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --pmc off "$@" > $OUT/bench_traced.json 2> $OUT/trace.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --pmc off "$@" > /dev/null 2> $OUT/pmc_sq.log
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --pmc off "$@" > /dev/null 2> $OUT/pmc_sq2.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_wr -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --pmc off "$@" > /dev/null 2> $OUT/pmc_wr.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_rd -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --pmc off "$@" > /dev/null 2> $OUT/pmc_rd.log
echo collected $OUT
