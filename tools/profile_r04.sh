#!/bin/bash
# One pass over the FINAL tree: kernel stats + FETCH/WRITE of the headline bench, SQ counters in situ, VAE kernel stats, the lossy-mode
# lines the review asked for, the headline line itself (with its e2e object) and the GPU test log.  Results -> gpurun_out/, to be copied
# into profiles/r04/.
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4
bash tools/profile_bench.sh r04 > gpurun_out/prof_r04.log 2>&1; tail -3 gpurun_out/prof_r04.log
bash tools/profile_bench_sq.sh r04 > gpurun_out/profsq_r04.log 2>&1; tail -12 gpurun_out/profsq_r04.log
bash tools/profile_vae.sh r04 > gpurun_out/profvae_r04.log 2>&1; tail -3 gpurun_out/profvae_r04.log
cd "$GRAFT_REPO_ROOT"
for arm in "--attn-stress" "--fp8 --fp8-layers attn,attn_pv --attn-stress" "--fp8 --fp8-layers qkv,ffn,o,cross,attn,attn_pv"; do
  name=$(echo "$arm" | tr -c 'a-zA-Z0-9' '_')
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e $arm > gpurun_out/r4/bench_$name.json 2> gpurun_out/r4/bench_$name.err; echo "bench [$arm] rc=$?"; cut -c1-300 gpurun_out/r4/bench_$name.json
done
timeout 900 python bench.py > gpurun_out/r4/bench_14b_final.json 2> gpurun_out/r4/bench_14b_final.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r4/bench_14b_final.json
