#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out/r5j; mkdir -p $o
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -6 $o/pytest_gpu.log
timeout 600 python tools/bench_gemm_yardstick.py > $o/gemm_yardstick.log 2>&1; cat $o/gemm_yardstick.log | grep -v amdgpu
