#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out/r5i; mkdir -p $o
LD_LIBRARY_PATH=$(pwd)/tools/exp_lib:$LD_LIBRARY_PATH timeout 600 ./tools/kernel_check gemmcyc > $o/gemmcyc_nt.log 2>&1; echo "gemmcyc rc=$?"; grep -E "form" $o/gemmcyc_nt.log | cut -c1-200
