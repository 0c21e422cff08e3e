#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out/r5h; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sp.py -x -q -k "split_k or gemm or graph_capture or compose" > $o/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -6 $o/pytest_sel.log
python tools/bench_gemm_yardstick.py --only "1.3B small" > $o/yard_small_splitk.log 2>&1; tail -5 $o/yard_small_splitk.log
python tools/bench_gemm_yardstick.py --only "1.3B small" --tuning gemm_splitk=0 > $o/yard_small_nosplit.log 2>&1; tail -5 $o/yard_small_nosplit.log
for sk in 1 0; do
  WAN_GEMM_SPLITK=$sk timeout 300 python bench.py --workload 1.3b-small --graph-loop --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $o/bench_13bsmall_sk$sk.json 2> $o/bench_13bsmall_sk$sk.err; echo "1.3b-small splitk=$sk rc=$?"; cut -c1-250 $o/bench_13bsmall_sk$sk.json
done
timeout 600 python -m pytest tests/test_gpu_t5.py tests/test_gpu_vae.py tests/test_gpu_dit.py -x -q > $o/pytest_models.log 2>&1; echo "pytest models rc=$?"; tail -4 $o/pytest_models.log
