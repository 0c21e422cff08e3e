#!/bin/bash
# rocprofv3 passes behind profiles/: kernel trace + stats, then one --pmc pass per counter
# (never combined with tracing domains).  Usage on the GPU box: bash tools/profile_bench.sh <tag> [extra bench.py arguments ...]
# (e.g. `--workload 14b-cof-720p`); PASSES="trace FETCH_SIZE" limits the passes, PASS_TIMEOUT the seconds allowed per pass.
set -u
tag=${1:-r01}
shift || true
extra="$*"
passes=${PASSES:-"trace FETCH_SIZE WRITE_SIZE"}
pt=${PASS_TIMEOUT:-600}
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$repo/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cmd="python $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-e2e --no-box-probe $extra"
for c in $passes; do
  if [ "$c" = trace ]; then
    timeout $pt rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $cmd > "$out/trace.log" 2>&1
  else
    timeout $pt rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -- $cmd > "$out/pmc_$c.log" 2>&1
  fi
done
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out/kernel_stats.csv"; fi
python "$repo/tools/pmc_summary.py" "$out/pmc_summary.json" fetch="$out/pmc_FETCH_SIZE" write="$out/pmc_WRITE_SIZE" > "$out/pmc_summary.txt" 2>&1
# keep the merged-back directory small: raw traces stay on the box
rm -rf "$out/trace" "$out"/pmc_FETCH_SIZE "$out"/pmc_WRITE_SIZE
ls -la "$out"
