#!/bin/bash
# Round-4 placement investigation, call 1 (GPU box): the probe without counters (does the mode exist with hipMalloc / with
# physical chunks of several sizes, does it follow the chunks or their order), then two counter passes over fresh hipMalloc
# allocations (translation misses; write-request stalls).  Output: gpurun_out/r4/placement_*.
set -e
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 240 $P malloc 8        > $O/placement_malloc.jsonl 2> $O/placement_malloc.err
timeout -k 10 300 $P synth 2048 4 4  > $O/placement_vmm2048.jsonl 2> $O/placement_vmm2048.err
timeout -k 10 240 $P vmm 32 2 1      > $O/placement_vmm32.jsonl 2> $O/placement_vmm32.err
timeout -k 10 240 $P vmm 2 1 0       > $O/placement_vmm2.jsonl 2> $O/placement_vmm2.err || echo "vmm 2 MiB: failed or timed out" >> $O/placement_vmm2.err
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_utcl -- $P malloc 6 > $O/pmc_utcl.jsonl 2> $O/pmc_utcl.log
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_wrreq -- $P malloc 6 > $O/pmc_wrreq.jsonl 2> $O/pmc_wrreq.log
echo done
