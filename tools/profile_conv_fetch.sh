#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) of the causal 3x3x3 convolution kernels (tools/conv_lin.py --profile).
set -u
tag=${1:-conv}
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$repo/gpurun_out/proffetch_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cmd="python $repo/tools/conv_lin.py --profile"
args=""
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d "$out/$c" -- $cmd > "$out/$c.log" 2>&1
  args="$args $c=$out/$c"
done
python "$repo/tools/pmc_summary.py" "$out/fetch_summary.json" $args > "$out/fetch_summary.txt" 2>&1
rm -rf "$out/FETCH_SIZE" "$out/WRITE_SIZE"
cat "$out/fetch_summary.txt"
