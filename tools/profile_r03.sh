#!/bin/bash
# One pass over the FINAL tree: kernel stats + FETCH/WRITE of the headline bench, SQ counters in situ, VAE kernel stats,
# the headline line itself and the GPU test log.  Results -> gpurun_out/, to be copied into profiles/r03/.
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/prof_r03_fp8
bash tools/profile_bench.sh r03 > gpurun_out/prof_r03.log 2>&1; tail -3 gpurun_out/prof_r03.log
bash tools/profile_bench_sq.sh r03 > gpurun_out/profsq_r03.log 2>&1; tail -8 gpurun_out/profsq_r03.log
bash tools/profile_vae.sh r03 > gpurun_out/profvae_r03.log 2>&1; tail -3 gpurun_out/profvae_r03.log
# the lossy mode with the fp8 QK^T attention kernel: kernel stats of one step (names + durations only; no PMC pass)
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r03_fp8/trace" -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-verify --fp8 --fp8-layers qkv,ffn,o,cross,attn,attn_pv \
    > "$GRAFT_REPO_ROOT/gpurun_out/prof_r03_fp8/trace.log" 2>&1
  f=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof_r03_fp8/trace" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/prof_r03_fp8/kernel_stats.csv"; rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_r03_fp8/trace" )
cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py > gpurun_out/r3_bench_14b_final.json 2> gpurun_out/r3_bench_14b_final.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r3_bench_14b_final.json
