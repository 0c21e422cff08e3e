#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out/r5n; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_vae.py -q > $o/pytest_vae.log 2>&1; echo "pytest vae rc=$?"; tail -3 $o/pytest_vae.log
for mi in 0 32 0 32; do WAN_CONV_MFMA=$mi timeout 300 python tools/bench_vae.py --iters 2 2>/dev/null | grep workload | cut -c60-260; done
