#!/bin/bash
# refresh, on the round's FINAL tree, what the last kernel commit (VAE convolution on 16x16x32 MFMAs) touches: VAE kernel stats, the headline line (e2e), the GPU test log
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=$(pwd)
o=gpurun_out/r05b; mkdir -p $o
bash tools/profile_vae.sh r05b > $o/profvae.log 2>&1; cp gpurun_out/profvae_r05b/kernel_stats.csv $o/vae_kernel_stats.csv 2>/dev/null; head -6 $o/vae_kernel_stats.csv | cut -c1-150
timeout 300 python tools/bench_vae.py --iters 2 2>/dev/null | grep workload > $o/vae_bench_81f_480p.json; cat $o/vae_bench_81f_480p.json | cut -c1-260
timeout 900 python bench.py > $o/bench_14b_final.json 2> $o/bench_14b_final.err; echo "bench rc=$?"; cut -c1-300 $o/bench_14b_final.json
timeout 1500 python -m pytest tests -q -m gpu > $o/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $o/pytest_gpu.log
