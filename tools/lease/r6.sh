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
