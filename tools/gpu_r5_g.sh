#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out/r5g; mkdir -p $o
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -8 $o/pytest_gpu.log
timeout 300 python bench.py --workload 1.3b-cof --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $o/bench_13bcof.json 2> $o/bench_13bcof.err; echo "1.3b-cof rc=$?"; cut -c1-250 $o/bench_13bcof.json
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $o/bench_14b.json 2> $o/bench_14b.err; echo "14b rc=$?"; cut -c1-250 $o/bench_14b.json
timeout 200 ./tools/kernel_check attnarms > $o/attnarms.log 2>&1; echo "attnarms rc=$?"; tail -2 $o/attnarms.log
