#!/bin/bash
# scripts/probes/store_probe.hip on the GPU box: rows of 1000 doubles (config 4: 62.5 L2 lines) against rows of 1008 (63 whole lines)
cd scripts/probes && (timeout -k 10 200 ./store_probe_all 10000 | grep -E "MODE (1|3|6) +W +(0|60) +LDS 48") && echo "--- rows of 1008 doubles" && (timeout -k 10 200 ./store_probe_1008 10000 | grep -E "MODE (1|3|6) +W +(0|60) +LDS 48")
