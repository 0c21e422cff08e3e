#!/bin/bash
# round 3: the fp8 QK^T attention variant -- kernel exactness, error statement, speed, model-level tests
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
( timeout 600 tools/kernel_check attnq8 ) > gpurun_out/attnq8.log 2>&1
echo "kernel_check rc=$?" >> gpurun_out/attnq8.log
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -s -k "qk8 or rope_fp8 or small_model or error_statement" > gpurun_out/q8_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/q8_tests.log
tail -40 gpurun_out/attnq8.log
tail -30 gpurun_out/q8_tests.log
