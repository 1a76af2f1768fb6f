#!/bin/bash
# Round-4 placement investigation, call 2: launch time against the size of the physical fragments (VMM chunks of 2..64 MiB; hipMalloc
# out of a VRAM whose free space was turned into holes), translation misses for a fragmented and an unfragmented allocation, and
# several fresh processes of plain hipMalloc with the translation counters on (to catch a naturally slow allocation).
O=gpurun_out/r4; mkdir -p $O
P=scripts/placement/probe
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in 4 8 16 64 512; do timeout -k 10 200 $P vmm $s 1 0 >> $O/placement_vmm_sizes.jsonl 2>> $O/placement_vmm_sizes.err; done
for h in 64 16 8 4; do timeout -k 10 280 $P frag $h >> $O/placement_frag.jsonl 2>> $O/placement_frag.err; echo "frag $h rc=$?" >> $O/placement_frag.err; done
PMC="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum GRBM_GUI_ACTIVE"
timeout -k 10 200 rocprofv3 --pmc $PMC --output-format csv -d $O/pmc_utcl_vmm2 -- $P vmm 2 1 0 > $O/pmc_utcl_vmm2.jsonl 2> $O/pmc_utcl_vmm2.log
timeout -k 10 200 rocprofv3 --pmc $PMC --output-format csv -d $O/pmc_utcl_vmm8 -- $P vmm 8 1 0 > $O/pmc_utcl_vmm8.jsonl 2> $O/pmc_utcl_vmm8.log
for i in 1 2 3 4 5 6; do timeout -k 10 120 rocprofv3 --pmc $PMC --output-format csv -d $O/pmc_utcl_fresh$i -- $P malloc 3 > $O/pmc_utcl_fresh$i.jsonl 2> $O/pmc_utcl_fresh$i.log; done
timeout -k 10 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum --output-format csv -d $O/pmc_wrreq -- $P malloc 2 > $O/pmc_wrreq.jsonl 2> $O/pmc_wrreq.log
timeout -k 10 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum --output-format csv -d $O/pmc_wrreq_vmm2 -- $P vmm 2 1 0 > $O/pmc_wrreq_vmm2.jsonl 2> $O/pmc_wrreq_vmm2.log
echo done
