#!/bin/bash
# Round 5, call N: VAE suite with the frame-count-independent MFMA form rule + VAE bench line
set -x
mkdir -p gpurun_out/r05n
timeout 900 python -m pytest tests/test_gpu_vae.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/r05n/pytest_vae.log
timeout 600 python tools/bench_vae.py 2>&1 | tail -12 | tee gpurun_out/r05n/bench_vae.log
