#!/bin/bash
# round 4: the whole GPU suite + the headline bench line (+ optional extra bench arms given as arguments)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r4/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; grep -E "passed|failed|error" gpurun_out/r4/pytest_gpu.log | tail -3; grep -E "FAILED|Error" gpurun_out/r4/pytest_gpu.log | head -20
timeout 900 python bench.py --steps 4 --warmup 1 > gpurun_out/r4/bench_14b.json 2> gpurun_out/r4/bench_14b.err; echo "bench rc=$?"; cat gpurun_out/r4/bench_14b.json
for arm in "$@"; do
  name=$(echo "$arm" | tr -c 'a-zA-Z0-9' '_')
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $arm > gpurun_out/r4/bench_$name.json 2> gpurun_out/r4/bench_$name.err; echo "bench [$arm] rc=$?"; cat gpurun_out/r4/bench_$name.json
done
