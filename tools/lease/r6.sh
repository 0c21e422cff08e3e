#!/bin/bash
# Round-6 GPU leases, one stage per gpurun call:   gpurun --timeout N -- 'bash tools/lease/r6.sh <stage>'
# Everything a stage produces goes under gpurun_out/r06_<stage>/ (merged back); what is to be judged is copied to profiles/r06/ by hand.
set -u
stage=${1:?stage}
out=gpurun_out/r06_$stage
mkdir -p "$out"
B="python bench.py"
json() { grep '^{' | tail -1; }
case $stage in
  a)  # the new bench plumbing: box probe, SP parity over gloo on a shared GPU (N = 2, 4), the 33-frame demo configuration, 720p CoF
    timeout 900 python -m pytest tests/test_gpu_sp.py -x -q -k "async_path or self_validating" > $out/pytest_sp.log 2>&1; tail -3 $out/pytest_sp.log
    timeout 600 $B --steps 2 --no-cpu-baseline --no-e2e 2>$out/bench_14b.err | json > $out/bench_14b_2steps.json
    timeout 900 $B --workload 14b-cof-33f --steps 4 --no-cpu-baseline 2>$out/bench_33f.err | json > $out/bench_14b_cof_33f.json
    timeout 900 $B --gpus 2 --backend gloo --share-gpu --workload 14b-cof --steps 1 --warmup 1 2>$out/bench_sp2.err | json > $out/bench_sp2_gloo_shared_gpu_14b.json
    timeout 1200 $B --gpus 4 --backend gloo --share-gpu --workload 14b-cof --steps 1 --warmup 1 2>$out/bench_sp4.err | json > $out/bench_sp4_gloo_shared_gpu_14b.json
    timeout 900 $B --workload 14b-cof-720p --steps 1 --warmup 1 --no-cpu-baseline 2>$out/bench_720p.err | json > $out/bench_14b_cof_720p.json
    ;;
  b)  # GEMM rasterisation sweep (+ FETCH_SIZE per arm), the mixed-MFMA-shape probe, the 321f@720p length-extrapolation config on one GPU, whole GPU suite
    free -g | head -2; df -h /tmp | tail -1; nproc
    timeout 300 tools/probe/mfma_power > $out/mfma_power_mixed.log 2>&1; grep -E "MIXED|softmax VALU" $out/mfma_power_mixed.log
    timeout 600 tools/kernel_check gemmgm > $out/gemm_gm_sweep.log 2>&1; grep "gemm " $out/gemm_gm_sweep.log
    ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$out/pmc_gm -- $OLDPWD/tools/kernel_check gemmgm 0,1,2,3,4,6,8,16 once > $OLDPWD/$out/pmc_gm.log 2>&1 )
    python - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/pmc_gm/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "gemm_pk_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gms = [0, 1, 2, 3, 4, 6, 8, 16]
shapes = ["ffn.0+gelu", "qk proj", "o proj+gate+resid", "ffn.2+resid", "v proj (T)"]
with open(out + "/gemm_gm_fetch.log", "w") as fo:
    for i, r in enumerate(rows):
        line = "%-18s gm=%-2d FETCH_SIZE x2 = %.2f GB  (%.3f ms under the counter pass)" % (shapes[i // len(gms)] if i // len(gms) < len(shapes) else "?", gms[i % len(gms)],
               float(r["Counter_Value"]) * 1024 * 2 / 1e9, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        print(line); fo.write(line + "\n")
PY
    rm -rf $out/pmc_gm
    timeout 1500 $B --workload 14b-cof-321f-720p --steps 1 --warmup 0 --no-cpu-baseline 2>$out/bench_321f.err | json > $out/bench_14b_cof_321f_720p.json
    PASSES="trace FETCH_SIZE" PASS_TIMEOUT=900 bash tools/profile_bench.sh 14b_cof_321f_720p_2layers --workload 14b-cof-321f-720p --layers 2 > $out/prof_321f.log 2>&1
    PASSES="trace FETCH_SIZE WRITE_SIZE" PASS_TIMEOUT=900 bash tools/profile_bench.sh 14b_cof_720p_4layers --workload 14b-cof-720p --layers 4 > $out/prof_720p.log 2>&1
    timeout 900 python tools/bench_ingest.py --layers 40 2>$out/ingest.err | json > $out/ingest_14b.json; cat $out/ingest_14b.json
    timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
    ;;
  *) echo "unknown stage $stage"; exit 2;;
esac
for f in $out/*.json; do echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
except Exception as e:
    print("  (no JSON line)", e); sys.exit(0)
keys = ("value", "value_normalised", "ms_per_step", "n_gpus", "parity", "box", "rank_wall_s", "exposed_comm_ms")
for k in keys:
    v = d.get(k)
    if k == "box" and v: v = {kk: v[kk] for kk in ("mfma_mix_tflops", "copy_tbps", "rel_to_reference")} | {"before": v["before"]["mfma_mix_tflops"], "after": v["after"]["mfma_mix_tflops"]}
    if k == "parity" and v: v = {kk: v.get(kk) for kk in ("ok", "rel_l2", "cosine", "error")}
    if k == "exposed_comm_ms" and v: v = {kk: v.get(kk) for kk in ("per_step", "per_step_by_exchange")}
    print(" ", k, v)
r = d.get("roofline")
if r: print("  roofline", {k: r.get(k) for k in ("achieved", "frac", "avg_ms", "launches", "variant_code", "traffic")})
e = d.get("e2e")
if e: print("  e2e", {k: e.get(k) for k in ("sec_per_video", "stages_s", "error")})
PY
done
for f in $out/*.err; do echo "== $f"; tail -5 "$f"; done
