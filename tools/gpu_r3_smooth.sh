#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -s > gpurun_out/fp8_tests_smooth.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/fp8_tests_smooth.log
timeout 600 python bench.py --no-cpu-baseline --fp8 --fp8-layers qkv,ffn,o,cross,attn > gpurun_out/bench_fp8_all_attn_smooth.json 2> gpurun_out/bench_fp8_all_attn_smooth.err; echo "rc=$?"; tail -c 700 gpurun_out/bench_fp8_all_attn_smooth.json; python -c "
import json; d=json.loads(open('gpurun_out/bench_fp8_all_attn_smooth.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['parity'])"
