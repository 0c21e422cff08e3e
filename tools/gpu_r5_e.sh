#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out/r5e; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_sp.py tests/test_gpu_fp8.py -x -q > $o/pytest_sp_fp8.log 2>&1; echo "pytest sp+fp8 rc=$?"; tail -30 $o/pytest_sp_fp8.log
