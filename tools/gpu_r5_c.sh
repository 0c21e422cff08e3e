#!/bin/bash
# Round 5, call C: the persistent GEMM's row-permuted epilogues (numerics, bitwise vs the round-4 forms, in-process A/B on the 14B and
# 1.3B shapes), the GPU tests that touch the GEMM, the tile-mask clock experiment of the attention kernel, lazy-first vs max-free in situ.
cd "$(dirname "$0")/.." || exit 1
repo=$(pwd)
o=$repo/gpurun_out/r5c; mkdir -p $o
timeout 600 ./tools/kernel_check gemmpk > $o/gemmpk.log 2>&1; echo "gemmpk rc=$?"; grep -E "FAIL|PASSED|mismatch" $o/gemmpk.log | head -20; grep "gemm\[" $o/gemmpk.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > $o/pytest_gemm.log 2>&1; echo "pytest gemm rc=$?"; tail -5 $o/pytest_gemm.log
cd /tmp && export TMPDIR=/tmp
for arm in 0 1; do
  i=0; args=""
  for grp in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    WAN_ATTN_EXP=$arm timeout 300 rocprofv3 --pmc $grp --output-format csv -d $o/exp$arm/p$i -- $repo/tools/kernel_check attnprof > $o/exp${arm}_p$i.log 2>&1
    args="$args p$i=$o/exp$arm/p$i"
  done
  python $repo/tools/pmc_summary.py $o/attn_tilemask${arm}_pmc.json $args > $o/attn_tilemask${arm}_pmc.txt 2>&1
  rm -rf $o/exp$arm
  head -3 $o/attn_tilemask${arm}_pmc.txt
done
cd $repo
./tools/kernel_check attnx 40 "" "attn_exp=1" > $o/attnx_tilemask.log 2>&1; cat $o/attnx_tilemask.log
for f in 1 0 1 0; do
  WAN_ATTN_FAST=$f timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-verify > $o/bench14b_fast${f}_$RANDOM.json 2> $o/bench14b_fast$f.err; echo "14b attn_fast=$f rc=$?"
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob("gpurun_out/r5c/bench14b_fast*.json")):
    d=json.load(open(p)); print(p, d["ms_per_step"], d["roofline"]["avg_ms"], d["roofline"]["achieved"])
PY
