#!/bin/bash
# Round 5, call B: the attention questions of the round-4 review, measured in one process / on one box:
#  (1) 40 vs 12 vs 5 heads in process (is the 5-head shard really 10 % behind?), XCD pinning on / off, max-free vs lazy
#  (2) the mfma_power probe with half / none of the exponentials (the bound of a cheaper exponential)
#  (3) clock and fabric traffic of the self-attention launch with every K / V tile L2-resident (attn_exp = 1) vs the product
#  (4) XCD pinning in situ (bench, 2 steps)
cd "$(dirname "$0")/.." || exit 1
repo=$(pwd)
o=$repo/gpurun_out/r5b; mkdir -p $o
./tools/kernel_check attnx 40 "" "attn_xcd_map=0" "attn_fast=0" > $o/attnx_40.log 2>&1; cat $o/attnx_40.log
./tools/kernel_check attnx 12 "" "attn_fast=0" > $o/attnx_12.log 2>&1; cat $o/attnx_12.log
./tools/kernel_check attnx 5 "" "attn_fast=0" "attn_tail=0" > $o/attnx_5.log 2>&1; cat $o/attnx_5.log
./tools/probe/mfma_power > $o/mfma_power.log 2>&1; grep random $o/mfma_power.log
cd /tmp && export TMPDIR=/tmp
for arm in 0 1; do
  i=0; args=""
  for grp in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    WAN_ATTN_EXP=$arm timeout 300 rocprofv3 --pmc $grp --output-format csv -d $o/exp$arm/p$i -- $repo/tools/kernel_check attnprof > $o/exp${arm}_p$i.log 2>&1
    args="$args p$i=$o/exp$arm/p$i"
  done
  python $repo/tools/pmc_summary.py $o/attn_tilemask${arm}_pmc.json $args > $o/attn_tilemask${arm}_pmc.txt 2>&1
  rm -rf $o/exp$arm
  cat $o/attn_tilemask${arm}_pmc.txt | head -40
done
cd $repo
for x in 1 0; do
  WAN_ATTN_XCD_MAP=$x timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-verify > $o/bench14b_xcd$x.json 2> $o/bench14b_xcd$x.err; echo "14b xcd_map=$x rc=$?"; cut -c1-240 $o/bench14b_xcd$x.json
done
timeout 300 python tools/probe/attn_head_groups.py > $o/attn_head_groups.log 2>&1; cat $o/attn_head_groups.log
