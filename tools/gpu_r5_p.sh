#!/bin/bash
# the N > 1 code path of bench.py after this round's changes to the Ulysses branch: ranks share cuda:0, host-staged gloo exchanges (numbers meaningless)
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out/r5p; mkdir -p $o
timeout 600 python bench.py --gpus 2 --backend gloo --share-gpu --workload 1.3b-cof --steps 1 --warmup 1 --no-cpu-baseline > $o/bench_selfspawn_gloo_shared_gpu_n2_1p3b.json 2> $o/n2.err; echo "n2 rc=$?"; cut -c1-400 $o/bench_selfspawn_gloo_shared_gpu_n2_1p3b.json; tail -3 $o/n2.err
timeout 900 python bench.py --gpus 4 --backend gloo --share-gpu --workload 14b-cof --steps 1 --warmup 1 --no-cpu-baseline --fp8 --fp8-layers attn,attn_pv > $o/bench_selfspawn_gloo_shared_gpu_n4_14b_fp8attn.json 2> $o/n4.err; echo "n4 fp8 rc=$?"; cut -c1-400 $o/bench_selfspawn_gloo_shared_gpu_n4_14b_fp8attn.json; tail -3 $o/n4.err
