#!/bin/bash
# rocprofv3 kernel statistics of one WanVAE encode + decode (tools/bench_vae.py).  Usage on the GPU box: bash tools/profile_vae.sh <tag>
set -u
tag=${1:-vae}
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$repo/gpurun_out/profvae_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/k" -- python "$repo/tools/bench_vae.py" --iters 1 > "$out/run.log" 2>&1
f=$(find "$out/k" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out/kernel_stats.csv"
rm -rf "$out/k"
head -16 "$out/kernel_stats.csv" | cut -c1-200
