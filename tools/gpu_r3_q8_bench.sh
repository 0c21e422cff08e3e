#!/bin/bash
# round 3: bench lines of the lossy modes that include the fp8 QK^T attention variant
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 600 python bench.py --no-cpu-baseline --fp8 --fp8-layers attn > gpurun_out/bench_fp8_attn_only.json 2> gpurun_out/bench_fp8_attn_only.err
echo "rc=$?"; tail -c 1800 gpurun_out/bench_fp8_attn_only.json
timeout 600 python bench.py --no-cpu-baseline --fp8 --fp8-layers qkv,ffn,o,cross,attn > gpurun_out/bench_fp8_all_attn.json 2> gpurun_out/bench_fp8_all_attn.err
echo "rc=$?"; tail -c 1800 gpurun_out/bench_fp8_all_attn.json
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_bf16_same_box.json 2> gpurun_out/bench_bf16_same_box.err
echo "rc=$?"; tail -c 600 gpurun_out/bench_bf16_same_box.json
