#!/bin/bash
# Round 5, call A: baselines that decide the GEMM work -- the persistent kernel on the K = 1536 shapes (gemm_pk = 2) vs the
# dispatcher's current choice, in the yardstick and in the 1.3B bench lines.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r5a
o=gpurun_out/r5a
python tools/bench_gemm_yardstick.py --only "1.3B" > $o/yard_13b_default.log 2>&1; tail -12 $o/yard_13b_default.log
python tools/bench_gemm_yardstick.py --only "1.3B" --tuning gemm_pk=2 > $o/yard_13b_pk2.log 2>&1; tail -12 $o/yard_13b_pk2.log
for pk in 1 2; do
  WAN_GEMM_PK=$pk timeout 300 python bench.py --workload 1.3b-cof --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $o/bench_13bcof_pk$pk.json 2> $o/bench_13bcof_pk$pk.err; echo "1.3b-cof pk=$pk rc=$?"; cut -c1-260 $o/bench_13bcof_pk$pk.json
  WAN_GEMM_PK=$pk timeout 300 python bench.py --workload 1.3b-small --graph-loop --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $o/bench_13bsmall_pk$pk.json 2> $o/bench_13bsmall_pk$pk.err; echo "1.3b-small pk=$pk rc=$?"; cut -c1-260 $o/bench_13bsmall_pk$pk.json
done
timeout 300 python tools/probe/attn_head_groups.py > $o/attn_head_groups_base.log 2>&1; cat $o/attn_head_groups_base.log
