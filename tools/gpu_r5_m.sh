#!/bin/bash
# the VAE's LDS-patch convolution on v_mfma_f32_16x16x32_bf16 (conv_mfma = 16) against the product's 32x32x16 form
cd "$(dirname "$0")/.." || exit 1
repo=$(pwd); o=$repo/gpurun_out/r5m; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_vae.py -x -q -k "lds_patch or g9 or g10 or chunk" > $o/pytest_vae.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_vae.log
timeout 600 python tools/bench_conv.py --ab-mfma > $o/conv_mfma_ab.log 2>&1; grep -v amdgpu $o/conv_mfma_ab.log
for mi in 32 16; do
  WAN_CONV_MFMA=$mi timeout 300 python tools/bench_vae.py --iters 2 > $o/vae_mfma$mi.log 2>&1; echo "== bench_vae conv_mfma=$mi"; grep -v amdgpu $o/vae_mfma$mi.log | tail -4
done
cd /tmp && export TMPDIR=/tmp
for mi in 32 16; do
  WAN_CONV_MFMA=$mi timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $o/sq$mi -- python $repo/tools/bench_conv.py "--only=dec stage3" > $o/sq$mi.log 2>&1
  python $repo/tools/pmc_summary.py $o/conv_mfma${mi}_sq.json sq=$o/sq$mi > $o/conv_mfma${mi}_sq.txt 2>&1
  rm -rf $o/sq$mi
  echo "== SQ conv_mfma=$mi"; grep conv3_patch $o/conv_mfma${mi}_sq.txt | head -2
  python - <<PY
import json
d=json.load(open("$o/conv_mfma${mi}_sq.json")); print({k:v for k,v in d.get("_derived",{}).items() if "conv3_patch" in k})
PY
done
