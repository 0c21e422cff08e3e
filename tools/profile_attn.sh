#!/bin/bash
# PMC passes on the self-attention kernel alone (tools/kernel_check attnprof: L = 67 080, 40 heads, pre-scaled q).
# One group of SQ counters per pass, no tracing domains.  Usage on the GPU box: bash tools/profile_attn.sh <tag>
set -u
tag=${1:-final}
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$repo/gpurun_out/attnpmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
args=""
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -- "$repo/tools/kernel_check" attnprof > "$out/p$i.log" 2>&1
  args="$args p$i=$out/p$i"
done
python "$repo/tools/pmc_summary.py" "$out/attn_pmc.json" $args > "$out/attn_pmc.txt" 2>&1
rm -rf "$out"/p1 "$out"/p2 "$out"/p3 "$out"/p4
cat "$out/attn_pmc.txt"
