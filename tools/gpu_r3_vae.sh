#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vae.py -q -x 2>&1 | tail -4
for n in 1 4 5; do
  WAN_VAE_DECODE_CHUNK=$n timeout 600 python tools/bench_vae.py > gpurun_out/r3_vae_chunk$n.json 2>/dev/null; cat gpurun_out/r3_vae_chunk$n.json
done
