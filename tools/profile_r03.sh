#!/bin/bash
# One pass over the FINAL tree: kernel stats + FETCH/WRITE of the headline bench, SQ counters in situ, VAE kernel stats,
# the headline line itself and the GPU test log.  Results -> gpurun_out/, to be copied into profiles/r03/.
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/profile_bench.sh r03 > gpurun_out/prof_r03.log 2>&1; tail -3 gpurun_out/prof_r03.log
bash tools/profile_bench_sq.sh r03 > gpurun_out/profsq_r03.log 2>&1; tail -8 gpurun_out/profsq_r03.log
bash tools/profile_vae.sh r03 > gpurun_out/profvae_r03.log 2>&1; tail -3 gpurun_out/profvae_r03.log
cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py > gpurun_out/r3_bench_14b_final.json 2> gpurun_out/r3_bench_14b_final.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r3_bench_14b_final.json
