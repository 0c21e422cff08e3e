#!/bin/bash
# SQ-counter passes over one bench.py step (per-kernel matrix-pipe occupancy and wait cycles IN SITU).
# Usage on the GPU box: [BENCH_ARGS="--fp8 --fp8-layers ..."] bash tools/profile_bench_sq.sh <tag>
set -u
tag=${1:-sq}
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$repo/gpurun_out/profsq_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cmd="python $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-e2e ${BENCH_ARGS:-}"
i=0; args=""
for grp in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -- $cmd > "$out/p$i.log" 2>&1
  args="$args p$i=$out/p$i"
done
python "$repo/tools/pmc_summary.py" "$out/sq_summary.json" $args > "$out/sq_summary.txt" 2>&1
rm -rf "$out"/p1 "$out"/p2
cat "$out/sq_summary.txt"
