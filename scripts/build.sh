#!/bin/bash
# builds libpdmphip.so (and fails loudly); usage: scripts/build.sh
cd "$(dirname "$0")/../pdmpflux.jl_amd/csrc" || exit 1
out=$(make -j8 -s 2>&1); rc=$?
echo "$out" | grep -E "error|Spill: [1-9]|Error|scratch memory" | head -20
ls -la ../libpdmphip.so
exit $rc
