#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out/r5d; mkdir -p $o
timeout 900 ./tools/kernel_check gemmpk > $o/gemmpk.log 2>&1; echo "gemmpk rc=$?"; grep -cE "PASS" $o/gemmpk.log; grep -E "FAIL" $o/gemmpk.log | head; grep "ratio" $o/gemmpk.log
