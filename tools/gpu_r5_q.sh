#!/bin/bash
# Round 5, call Q: does the box slow down as it warms up?  The same bench command four times back to back on one box.
mkdir -p gpurun_out/r05q
for i in 1 2 3 4; do
  timeout 600 python bench.py --steps 4 --warmup 1 --no-verify --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05q/run$i.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05q/run$i.json").read())
print("run $i", d["ms_per_step"], d["roofline"]["avg_ms"], d["roofline"]["frac"])
PY
  rocm-smi --showtemp --showpower --showclocks 2>/dev/null | grep -E "Temperature \(Sensor (junction|memory)|Power|sclk" | head -6
done 2>&1 | tee gpurun_out/r05q/summary.log
