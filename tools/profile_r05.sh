#!/bin/bash
# ONE pass over the FINAL round-5 tree (results -> gpurun_out/r05/, to be copied into profiles/r05/):
#   * for EACH 1-GPU-runnable BASELINE config -- 14b-cof (configs[2], the headline), 1.3b-cof (configs[1]), 14b-720p (configs[3] shape),
#     1.3b-small (configs[0] shape, whole loop from one hipGraph) -- a bench line and the rocprofv3 --kernel-trace --stats summary of a
#     1-step run of the same command;
#   * for the headline: FETCH_SIZE / WRITE_SIZE PMC passes and the SQ counters in situ (separate --pmc passes, no tracing domains);
#   * the WanVAE kernel stats, the lossy-mode lines, the headline line itself in the driver's form (with cpu_baseline and e2e),
#     the GEMM yardstick against hipBLASLt and the GPU test log.
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
repo=$GRAFT_REPO_ROOT
o=$repo/gpurun_out/r05; mkdir -p $o
bash tools/profile_bench.sh r05 > $o/prof_14b.log 2>&1; tail -3 $o/prof_14b.log
cp gpurun_out/prof_r05/kernel_stats.csv $o/bench14b_kernel_stats.csv 2>/dev/null; cp gpurun_out/prof_r05/pmc_summary.json $o/bench14b_pmc_summary.json 2>/dev/null
bash tools/profile_bench_sq.sh r05 > $o/profsq_14b.log 2>&1; tail -8 $o/profsq_14b.log
cp gpurun_out/profsq_r05/sq_summary.json $o/bench14b_sq_insitu.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for wl in 1.3b-cof 14b-720p 1.3b-small; do
  extra=""; [ "$wl" = "1.3b-small" ] && extra="--graph-loop --steps 8 --warmup 2"
  [ "$wl" != "1.3b-small" ] && extra="--steps 1 --warmup 0"
  name=$(echo $wl | tr '.' 'p' | tr '-' '_')
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace_$name -- python $repo/bench.py --workload $wl $extra --no-cpu-baseline --no-verify --no-e2e > $o/trace_$name.log 2>&1
  f=$(find $o/trace_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $o/bench_${name}_kernel_stats.csv
  rm -rf $o/trace_$name
done
cd $repo
bash tools/profile_vae.sh r05 > $o/profvae.log 2>&1; cp gpurun_out/profvae_r05/kernel_stats.csv $o/vae_kernel_stats.csv 2>/dev/null
timeout 400 python bench.py --workload 1.3b-cof --steps 4 --warmup 1 --no-cpu-baseline --no-e2e > $o/bench_1p3b_cof.json 2> $o/bench_1p3b_cof.err; echo "1.3b-cof rc=$?"; cut -c1-260 $o/bench_1p3b_cof.json
timeout 600 python bench.py --workload 14b-720p --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $o/bench_14b_720p.json 2> $o/bench_14b_720p.err; echo "14b-720p rc=$?"; cut -c1-260 $o/bench_14b_720p.json
timeout 300 python bench.py --workload 1.3b-small --graph-loop --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $o/bench_1p3b_small_graph_loop.json 2> $o/bench_1p3b_small.err; echo "1.3b-small rc=$?"; cut -c1-260 $o/bench_1p3b_small_graph_loop.json
timeout 300 python bench.py --workload 1.3b-small --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $o/bench_1p3b_small_eager.json 2>> $o/bench_1p3b_small.err; cut -c1-260 $o/bench_1p3b_small_eager.json
for arm in "--attn-stress" "--fp8 --fp8-layers attn,attn_pv --attn-stress" "--fp8 --fp8-layers qkv,ffn,o,cross,attn,attn_pv"; do
  name=$(echo "$arm" | tr -c 'a-zA-Z0-9' '_')
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e $arm > $o/bench_14b_$name.json 2> $o/bench_14b_$name.err; echo "bench [$arm] rc=$?"; cut -c1-200 $o/bench_14b_$name.json
done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_14b_driver_like_20_steps.json 2> $o/bench_14b_driver_like.err; echo "driver-like rc=$?"; cut -c1-400 $o/bench_14b_driver_like_20_steps.json
timeout 900 python bench.py > $o/bench_14b_final.json 2> $o/bench_14b_final.err; echo "bench rc=$?"; cut -c1-400 $o/bench_14b_final.json
timeout 600 python tools/bench_gemm_yardstick.py > $o/gemm_yardstick_final_tree.log 2>&1; grep -v amdgpu $o/gemm_yardstick_final_tree.log | tail -24
timeout 300 python tools/probe/attn_head_groups.py > $o/attn_head_groups.log 2>&1; cat $o/attn_head_groups.log | grep -v amdgpu
timeout 1500 python -m pytest tests -q -m gpu > $o/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $o/pytest_gpu.log
