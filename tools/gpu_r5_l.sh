#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out/r5l; mkdir -p $o
timeout 1500 python -m pytest tests -q -m gpu > $o/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -4 $o/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
