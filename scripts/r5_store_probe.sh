#!/bin/bash
# scripts/probes/store_probe.hip on the GPU box: what the row stores of config 4 cost beside W x 16 FMAs per event
cd scripts/probes && timeout -k 10 200 ./store_probe_all 10000
