#!/bin/bash
# round 3: the fp8 QK^T + fp8 P.V attention variant -- tests, bench lines of the lossy modes on one box
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -s > gpurun_out/fp8_tests_f8.log 2>&1; echo "pytest rc=$?"; tail -32 gpurun_out/fp8_tests_f8.log
for arm in "attn,attn_pv" "qkv,ffn,o,cross,attn,attn_pv"; do
  name=$(echo "$arm" | tr -c 'a-zA-Z0-9' '_')
  timeout 600 python bench.py --no-cpu-baseline --fp8 --fp8-layers $arm > gpurun_out/bench_fp8_$name.json 2> gpurun_out/bench_fp8_$name.err; echo "bench [$arm] rc=$?"
  python -c "
import json; d=json.loads(open('gpurun_out/bench_fp8_$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], d['mfma_frac_whole_step'], r['avg_ms'], r['achieved'], r['frac'], r['variant_code'], r['attn_flagged_wgs'], d['parity'])"
done
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_bf16_same_box_f8.json 2> gpurun_out/bench_bf16_same_box_f8.err; echo "bench bf16 rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/bench_bf16_same_box_f8.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_ms'], r['achieved'])"
