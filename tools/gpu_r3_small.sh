#!/bin/bash
# round 3: (1) the self-spawning multi-rank bench on one shared GPU (gloo, host-staged: exercises the launcher + the N > 1
# JSON fields, the numbers mean nothing), (2) per-kernel profile of the BASELINE configs[0] shape.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for n in 2 4 8; do
  timeout 600 python bench.py --gpus $n --backend gloo --share-gpu --workload 1.3b-small --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_selfspawn_$n.json 2> gpurun_out/r3_selfspawn_$n.err; echo "self-spawn N=$n rc=$?"; cut -c1-600 gpurun_out/r3_selfspawn_$n.json
done
timeout 300 python bench.py --workload 1.3b-small --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r3_small_eager.json 2>/dev/null; cut -c1-300 gpurun_out/r3_small_eager.json
timeout 300 python bench.py --workload 1.3b-small --steps 8 --warmup 2 --no-cpu-baseline --graph > gpurun_out/r3_small_graph.json 2>/dev/null; cut -c1-300 gpurun_out/r3_small_graph.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_small -o small -- python $GRAFT_REPO_ROOT/bench.py --workload 1.3b-small --steps 8 --warmup 2 --no-cpu-baseline --no-verify --no-kernel-events > /dev/null 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_small -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -30 {}'
timeout 600 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_fullsize.py -q -s 2>&1 | grep -E "passed|failed|1.3B block|FAILED" | tail -5
