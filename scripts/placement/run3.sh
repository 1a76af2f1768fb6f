#!/bin/bash
# Round-4 placement investigation, call 3: (a) the fragmented hipMalloc under the translation counters, (b) plain hipMalloc in fresh
# processes after churn, WITHOUT counters and with the write-request stall counters, (c) physically contiguous allocations,
# (d) per-chunk timing of arrays built from 1-GiB chunks, slowest chunks exchanged.
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PMC="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_GUI_ACTIVE"
timeout -k 10 120 $P malloc 4 > $O/c3_malloc_first.jsonl 2>&1
timeout -k 10 280 rocprofv3 --pmc $PMC --output-format csv -d $O/c3_pmc_utcl_frag4 -- $P frag 4 > $O/c3_pmc_utcl_frag4.jsonl 2> $O/c3_pmc_utcl_frag4.log
for i in 1 2 3 4; do timeout -k 10 120 $P malloc 3 > $O/c3_malloc_after_churn$i.jsonl 2>&1; done
for i in 1 2; do timeout -k 10 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum --output-format csv -d $O/c3_pmc_wrreq$i -- $P malloc 3 > $O/c3_pmc_wrreq$i.jsonl 2> $O/c3_pmc_wrreq$i.log; done
timeout -k 10 120 $P contig 3 > $O/c3_contig.jsonl 2>&1
timeout -k 10 280 $P chunks 1024 3 16 > $O/c3_chunks.jsonl 2>&1
timeout -k 10 120 $P malloc 4 > $O/c3_malloc_last.jsonl 2>&1
echo done
