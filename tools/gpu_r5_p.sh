#!/bin/bash
# Round 5, call P: ragged batches in one attention launch (wan_attention_fwd_varlen) -- kernel tests + in-situ cost check
set -x
mkdir -p gpurun_out/r05p
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attention" 2>&1 | tail -8 | tee gpurun_out/r05p/pytest_attention.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-verify --no-e2e --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r05p/bench_2steps.json
