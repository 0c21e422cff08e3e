#!/bin/bash
# fabric traffic of the self-attention launch at 5 / 12 / 40 heads (L = 67 080), FETCH_SIZE and WRITE_SIZE in separate PMC passes
cd "$(dirname "$0")/.." || exit 1
repo=$(pwd); o=$repo/gpurun_out/r5k; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
for H in 5 12 40; do
  args=""
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $o/h$H/$c -- $repo/tools/kernel_check attnx $H "" > $o/h${H}_$c.log 2>&1
    args="$args $c=$o/h$H/$c"
  done
  python $repo/tools/pmc_summary.py $o/attn_h${H}_pmc.json $args > $o/attn_h${H}_pmc.txt 2>&1
  rm -rf $o/h$H
  echo "== H=$H"; head -4 $o/attn_h${H}_pmc.txt
done
