#!/bin/bash
# SQ-counter passes over tools/conv_lin.py (the causal 3x3x3 convolution kernels): matrix-pipe occupancy, waits, LDS conflicts.
# Usage on the GPU box: bash tools/profile_conv_sq.sh <tag>
set -u
tag=${1:-conv}
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$repo/gpurun_out/profsq_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cmd="python $repo/tools/conv_lin.py --profile"
i=0; args=""
for grp in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -- $cmd > "$out/p$i.log" 2>&1
  args="$args p$i=$out/p$i"
done
python "$repo/tools/pmc_summary.py" "$out/sq_summary.json" $args > "$out/sq_summary.txt" 2>&1
rm -rf "$out"/p1 "$out"/p2 "$out"/p3
cat "$out/sq_summary.txt"
