#!/bin/bash
# Round 5, call O: the final tree -- full GPU suite, smoke(), default bench line
set -x
mkdir -p gpurun_out/r05o
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r05o/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r05o/smoke.log
timeout 900 python bench.py 2>&1 | tail -2 | tee gpurun_out/r05o/bench_default.log
