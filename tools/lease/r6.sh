#!/bin/bash
# Round-6 GPU leases, one stage per gpurun call:   gpurun --timeout N -- 'bash tools/lease/r6.sh <stage>'
# Everything a stage produces goes under gpurun_out/r06_<stage>/ (merged back); what is to be judged is copied to profiles/r06/ by hand.
set -u
stage=${1:?stage}
out=gpurun_out/r06_${stage}${3:-}
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
  c)  # persistent cross-attention: numerics, A/B; the round's kernel changes against the pre-change library on ONE box; GPU suite
    timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "cross_attention_persistent or attention" > $out/pytest_attn.log 2>&1; tail -3 $out/pytest_attn.log
    timeout 600 python tools/bench_cross_attn.py > $out/cross_attn_persistent_ab.log 2>&1; cat $out/cross_attn_persistent_ab.log
    OLD=$PWD/tools/exp_lib/libwan_hip_before_persistent_cross.so
    for i in 1 2; do
      timeout 400 $B --steps 4 --no-cpu-baseline --no-e2e --no-verify 2>/dev/null | json > $out/bench_new_$i.json
      WAN_HIP_LIB=$OLD timeout 400 $B --steps 4 --no-cpu-baseline --no-e2e --no-verify 2>/dev/null | json > $out/bench_old_$i.json
    done
    python - $out <<'PY'
import json, sys
out = sys.argv[1]
with open(out + "/step_ab_same_box.log", "w") as fo:
    for tag in ("new_1", "old_1", "new_2", "old_2"):
        try:
            d = json.load(open(f"{out}/bench_{tag}.json"))
            line = f"{tag}: {d['ms_per_step']:.2f} ms/step  value {d['value']:.0f}  normalised {d['value_normalised']:.0f}  attention {d['roofline']['avg_ms']:.3f} ms/launch  box {d['box']['mfma_mix_tflops']:.1f} TF (before {d['box']['before']['mfma_mix_tflops']:.1f} after {d['box']['after']['mfma_mix_tflops']:.1f})"
        except Exception as e:
            line = f"{tag}: no line ({e})"
        print(line); fo.write(line + "\n")
PY
    PASSES="trace" bash tools/profile_bench.sh 14b_new > $out/prof_new.log 2>&1
    WAN_HIP_LIB=$OLD PASSES="trace" bash tools/profile_bench.sh 14b_old > $out/prof_old.log 2>&1
    python - <<'PY'
import csv
def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(p))}
try:
    n, o = load("gpurun_out/prof_14b_new/kernel_stats.csv"), load("gpurun_out/prof_14b_old/kernel_stats.csv")
    tn, to = sum(v[2] for v in n.values()), sum(v[2] for v in o.values())
    print(f"GPU time per 1-step run: new {tn:.1f} ms, old {to:.1f} ms")
    for k in sorted(set(n) | set(o), key=lambda k: -(n.get(k, (0, 0, 0))[2] + o.get(k, (0, 0, 0))[2]))[:14]:
        print(f"  {k[:110]:110s} new {n.get(k, (0, 0, 0))[1]:9.3f} ms x{n.get(k, (0, 0, 0))[0]:<4d} old {o.get(k, (0, 0, 0))[1]:9.3f} ms x{o.get(k, (0, 0, 0))[0]}")
except Exception as e:
    print("kernel stats comparison failed:", e)
PY
    timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
    ;;
  d)  # early prefetch in the persistent cross-attention kernel; telemetry-carrying bench lines; ingest straight to the device; configs[1]
    timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "cross_attention_persistent or attention" > $out/pytest_attn.log 2>&1; tail -3 $out/pytest_attn.log
    timeout 600 python tools/bench_cross_attn.py > $out/cross_attn_persistent_ab.log 2>&1; cat $out/cross_attn_persistent_ab.log
    timeout 400 $B --steps 4 --no-cpu-baseline --no-e2e 2>$out/bench_1.err | json > $out/bench_14b_run1.json
    timeout 400 $B --steps 4 --no-cpu-baseline --no-e2e 2>$out/bench_2.err | json > $out/bench_14b_run2.json
    timeout 400 $B --workload 1.3b-cof --steps 4 --no-cpu-baseline 2>$out/bench_13.err | json > $out/bench_1p3b_cof.json
    timeout 900 python tools/bench_ingest.py --layers 40 2>$out/ingest.err | json > $out/ingest_14b.json; cat $out/ingest_14b.json
    timeout 900 python -m pytest tests/test_gpu_dit.py -x -q > $out/pytest_dit.log 2>&1; tail -3 $out/pytest_dit.log
    python - $out <<'PY'
import json, sys
for n in ("bench_14b_run1", "bench_14b_run2", "bench_1p3b_cof"):
    try:
        d = json.load(open(f"{sys.argv[1]}/{n}.json"))
        print(n, "telemetry", d["box"]["telemetry_during_timed_region"])
    except Exception as e:
        print(n, "no telemetry", e)
PY
    ;;
  final)  # the pass behind profiles/r06/ on the round's final tree: suite, smoke, headline line (default + driver form), rocprofv3 passes, yardstick
    timeout 1800 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
    timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -2 $out/smoke.log
    timeout 900 $B 2>$out/bench_default.err | json > $out/bench_14b_final.json
    timeout 900 $B --steps 20 --warmup 5 2>$out/bench_driver.err | json > $out/bench_14b_driver_like_20_steps.json
    timeout 600 $B --workload 14b-cof-33f 2>$out/bench_33f.err | json > $out/bench_14b_cof_33f_final.json
    timeout 600 $B --force-sp --steps 2 --no-cpu-baseline 2>$out/bench_forcesp.err | json > $out/bench_14b_force_sp_rccl_one_rank.json
    bash tools/profile_bench.sh r06_final > $out/prof.log 2>&1
    bash tools/profile_bench_sq.sh r06_final > $out/prof_sq.log 2>&1
    BENCH_ARGS="" PASSES="trace" bash tools/profile_bench.sh r06_fp8 --fp8 --fp8-layers qkv,ffn,o,cross,attn,attn_pv > $out/prof_fp8.log 2>&1
    timeout 600 $B --fp8 --fp8-layers qkv,ffn,o,cross,attn,attn_pv --no-cpu-baseline --no-e2e 2>$out/bench_fp8.err | json > $out/bench_14b_fp8_everything.json
    timeout 900 python tools/bench_gemm_yardstick.py > $out/gemm_yardstick_final_tree.log 2>&1; tail -25 $out/gemm_yardstick_final_tree.log
    ;;
  fp)  # one more box for the fingerprint table (0.6 s probe window): the headline line twice, 4 steps and the driver's 20
    timeout 600 $B --steps 4 --no-cpu-baseline --no-e2e 2>/dev/null | json > $out/bench_14b_run1.json
    timeout 600 $B --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | json > $out/bench_14b_run2_20steps.json
    timeout 600 $B --steps 4 --no-cpu-baseline --no-e2e 2>/dev/null | json > $out/bench_14b_run3.json
    [ "${2:-}" = tests ] && { timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_dit.py -q -x > $out/pytest_fullsize_dit.log 2>&1; tail -3 $out/pytest_fullsize_dit.log; }
    ;;
  e)  # padded heads under Ulysses (num_heads % P != 0); the 1.3B model sequence-parallel over 8 ranks on a shared GPU (gloo); whole suite
    timeout 1200 python -m pytest tests/test_gpu_sp.py -x -q > $out/pytest_sp.log 2>&1; tail -4 $out/pytest_sp.log
    timeout 1200 $B --gpus 8 --backend gloo --share-gpu --workload 1.3b-cof --layers 4 --steps 1 --warmup 1 --no-box-probe 2>$out/bench_sp8_13.err | json > $out/bench_sp8_gloo_shared_gpu_1p3b_padded_heads.json
    timeout 1800 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
    ;;
  g)  # every other bench mode once on the final tree (no exception, sane line): graphs, replicas, stress weights, the stand-alone tools
    timeout 300 $B --workload 1.3b-small --graph-loop --no-cpu-baseline 2>$out/b1.err | json > $out/bench_1p3b_small_graph_loop.json
    timeout 300 $B --workload 1.3b-small --graph --no-cpu-baseline 2>$out/b2.err | json > $out/bench_1p3b_small_graph.json
    timeout 300 $B --workload 1.3b-small --no-cpu-baseline 2>$out/b3.err | json > $out/bench_1p3b_small_eager.json
    timeout 600 $B --gpus 2 --mode dp --backend gloo --share-gpu --workload 1.3b-small --no-cpu-baseline 2>$out/b4.err | json > $out/bench_dp2_gloo_shared_gpu_1p3b_small.json
    timeout 600 $B --attn-stress --steps 2 --no-cpu-baseline --no-e2e 2>$out/b5.err | json > $out/bench_14b_attn_stress.json
    timeout 600 $B --workload 14b-720p --steps 2 --no-cpu-baseline 2>$out/b6.err | json > $out/bench_14b_720p.json
    timeout 600 python tools/bench_vae.py > $out/bench_vae.log 2>&1; tail -4 $out/bench_vae.log
    timeout 300 python tools/bench_t5.py > $out/bench_t5.log 2>&1; tail -3 $out/bench_t5.log
    ;;
  confirm)  # the last tree once more: suite, smoke, the default headline line
    timeout 1800 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
    timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -2 $out/smoke.log
    timeout 900 $B 2>$out/bench_default.err | json > $out/bench_14b_final.json
    ;;
  h)  # the e4m3 Linear on the persistent stream-K kernel ("schedule P"): numerics, A/B against the 8-wave per-tile kernel, the lossy bench line
    if [ "${2:-}" != "bench" ]; then
      timeout 600 python -m pytest tests/test_gpu_fp8.py -x -q -k "gemm" > $out/pytest_fp8_gemm.log 2>&1; tail -5 $out/pytest_fp8_gemm.log
      timeout 600 python tools/bench_gemm_fp8.py > $out/gemm_fp8_persistent_ab.log 2>&1; cat $out/gemm_fp8_persistent_ab.log
    fi
    if [ "${2:-}" = "bench" ]; then
      ALL="--fp8 --fp8-layers qkv,ffn,o,cross,attn,attn_pv"
      timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q > $out/pytest_fp8.log 2>&1; tail -3 $out/pytest_fp8.log
      timeout 600 $B $ALL --steps 4 --no-cpu-baseline --no-e2e 2>$out/bench_fp8.err | json > $out/bench_14b_fp8_everything.json
      timeout 600 $B --steps 4 --no-cpu-baseline --no-e2e 2>$out/bench_bf16.err | json > $out/bench_14b_bf16_same_box.json
      timeout 600 $B --fp8 --fp8-layers qkv,ffn,o,cross --steps 4 --no-cpu-baseline --no-e2e 2>$out/bench_fp8_lin.err | json > $out/bench_14b_fp8_linears_only.json
      PASSES="trace" PASS_TIMEOUT=600 bash tools/profile_bench.sh r06_fp8_pk $ALL > $out/prof_fp8.log 2>&1
      cp gpurun_out/prof_r06_fp8_pk/kernel_stats.csv $out/bench14b_fp8_everything_kernel_stats.csv
      BENCH_ARGS="$ALL" bash tools/profile_bench_sq.sh r06_fp8_pk > $out/profsq_fp8.log 2>&1
      cp gpurun_out/profsq_r06_fp8_pk/sq_summary.json $out/bench14b_fp8_everything_sq_insitu.json
    fi
    ;;
  emu)  # one rank of a P-way Ulysses group emulated on one GPU (bench.py --emulate-sp P): the compute-side projection of the N = 2 / 4 / 8 lines
    timeout 600 python -m pytest tests/test_gpu_sp.py -x -q -k "emulated" > $out/pytest_emu.log 2>&1; tail -3 $out/pytest_emu.log
    timeout 500 $B --steps 4 --no-cpu-baseline --no-e2e 2>$out/bench_n1.err | json > $out/bench_14b_n1_same_box.json
    for p in 8 4 2; do
      timeout 500 $B --emulate-sp $p --steps 4 --warmup 2 2>$out/bench_emu$p.err | json > $out/bench_14b_emulated_rank_of_sp$p.json
    done
    timeout 500 $B --steps 4 --no-cpu-baseline --no-e2e --no-verify 2>$out/bench_n1b.err | json > $out/bench_14b_n1_same_box_after.json
    timeout 500 $B --workload 1.3b-cof --steps 4 --no-cpu-baseline --no-e2e 2>$out/bench_1p3b_n1.err | json > $out/bench_1p3b_n1_same_box.json
    timeout 500 $B --workload 1.3b-cof --emulate-sp 8 --steps 4 --warmup 2 2>$out/bench_1p3b_emu8.err | json > $out/bench_1p3b_emulated_rank_of_sp8.json
    timeout 500 $B --workload 1.3b-cof --emulate-sp 4 --steps 4 --warmup 2 2>$out/bench_1p3b_emu4.err | json > $out/bench_1p3b_emulated_rank_of_sp4.json
    PASSES="trace" PASS_TIMEOUT=400 bash tools/profile_bench.sh r06_emu8 --emulate-sp 8 > $out/prof_emu8.log 2>&1
    cp gpurun_out/prof_r06_emu8/kernel_stats.csv $out/bench14b_emulated_rank_of_sp8_kernel_stats.csv 2>/dev/null
    ;;
  emu2)  # after the dispatch change (max-free attempt on few-round launches of long key streams): attention + SP tests, the emulated ranks again
    timeout 1200 python -m pytest tests/test_gpu_sp.py tests/test_gpu_kernels.py -x -q -k "sp_ or emulated or attention" > $out/pytest_attn_sp.log 2>&1; tail -3 $out/pytest_attn_sp.log
    timeout 500 $B --steps 4 --no-cpu-baseline --no-e2e 2>$out/bench_n1.err | json > $out/bench_14b_n1_same_box.json
    for p in 8 4; do
      timeout 500 $B --emulate-sp $p --steps 4 --warmup 2 2>$out/bench_emu$p.err | json > $out/bench_14b_emulated_rank_of_sp$p.json
    done
    WAN_ATTN_FAST=0 timeout 500 $B --emulate-sp 8 --steps 4 --warmup 2 2>$out/bench_emu8_lazy.err | json > $out/bench_14b_emulated_rank_of_sp8_lazy_form.json
    timeout 500 $B --workload 1.3b-cof --emulate-sp 8 --steps 4 --warmup 2 2>$out/bench_1p3b_emu8.err | json > $out/bench_1p3b_emulated_rank_of_sp8.json
    PASSES="trace" PASS_TIMEOUT=400 bash tools/profile_bench.sh r06_emu8 --emulate-sp 8 > $out/prof_emu8.log 2>&1
    cp gpurun_out/prof_r06_emu8/kernel_stats.csv $out/bench14b_emulated_rank_of_sp8_kernel_stats.csv 2>/dev/null
    ;;
  emu3)  # same-box A/B of the dispatch change at the 8-way shard: max-free attempt (new default) vs the lazy form (WAN_ATTN_FAST=0), alternating, 20 steps each
    for i in 1 2 3; do
      timeout 300 $B --emulate-sp 8 --steps 20 --warmup 5 --no-box-probe 2>/dev/null | json > $out/emu8_maxfree_$i.json
      WAN_ATTN_FAST=0 timeout 300 $B --emulate-sp 8 --steps 20 --warmup 5 --no-box-probe 2>/dev/null | json > $out/emu8_lazy_$i.json
    done
    python - $out <<'PY'
import json, sys
out = sys.argv[1]
with open(out + "/emulated_sp8_maxfree_vs_lazy_ab.log", "w") as fo:
    for i in (1, 2, 3):
        for tag in ("maxfree", "lazy"):
            try:
                d = json.load(open(f"{out}/emu8_{tag}_{i}.json"))
                line = f"{tag:8s} run {i}: {d['ms_per_step']:.2f} ms/step  attention {d['roofline']['avg_ms']:.3f} ms/layer (variant {d['roofline']['variant_code']})  exposed copies {d['exposed_comm_ms']['per_step']:.2f} ms/step"
            except Exception as e:
                line = f"{tag} run {i}: no line ({e})"
            print(line); fo.write(line + "\n")
PY
    ;;
  emu4)  # head groups on / off at the 8- and 4-way shards (compute side only: what the second attention launch per layer costs before it hides anything)
    for i in 1 2; do
      for p in 8 4; do
        for g in 2 1; do
          timeout 300 $B --emulate-sp $p --sp-head-groups $g --steps 20 --warmup 5 --no-box-probe 2>/dev/null | json > $out/emu${p}_groups${g}_$i.json
        done
      done
    done
    python - $out <<'PY'
import json, sys
out = sys.argv[1]
with open(out + "/emulated_sp_head_groups_ab.log", "w") as fo:
    for i in (1, 2):
        for p in (8, 4):
            for g in (2, 1):
                try:
                    d = json.load(open(f"{out}/emu{p}_groups{g}_{i}.json"))
                    line = f"rank of {p}, head groups {g}, run {i}: {d['ms_per_step']:.2f} ms/step  attention {d['roofline']['avg_ms']:.3f} ms/layer  exposed copies {d['exposed_comm_ms']['per_step']:.2f} ms/step"
                except Exception as e:
                    line = f"rank of {p}, head groups {g}, run {i}: no line ({e})"
                print(line); fo.write(line + "\n")
PY
    ;;
  last)  # the round's LAST tree: whole suite, smoke, headline line (default + driver form), rocprofv3 passes (trace, FETCH / WRITE, SQ), the emulated-rank table on the same box
    timeout 1800 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
    timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -2 $out/smoke.log
    timeout 900 $B 2>$out/bench_default.err | json > $out/bench_14b_last_tree.json
    for p in 2 4 8; do
      timeout 500 $B --emulate-sp $p --steps 8 --warmup 2 2>$out/bench_emu$p.err | json > $out/bench_14b_emulated_rank_of_sp${p}_last_tree.json
    done
    timeout 900 $B --steps 20 --warmup 5 2>$out/bench_driver.err | json > $out/bench_14b_last_tree_driver_like_20_steps.json
    bash tools/profile_bench.sh r06_last > $out/prof.log 2>&1
    bash tools/profile_bench_sq.sh r06_last > $out/prof_sq.log 2>&1
    cp gpurun_out/prof_r06_last/kernel_stats.csv $out/bench14b_kernel_stats.csv 2>/dev/null
    cp gpurun_out/prof_r06_last/pmc_summary.json $out/bench14b_pmc_summary.json 2>/dev/null
    cp gpurun_out/profsq_r06_last/sq_summary.json $out/bench14b_sq_insitu.json 2>/dev/null
    ;;
  sc1)  # stream-K fix-up without the agent-scope acquire fence (buffer_inv sc1 = the XCD's L2 dropped at every split tile): agent-coherent loads instead
    ALT=$PWD/tools/exp_lib/libwan_hip_sc1.so
    WAN_HIP_LIB=$ALT timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or stream_k or persistent" > $out/pytest_gemm_sc1.log 2>&1; tail -3 $out/pytest_gemm_sc1.log
    for i in 1 2; do
      timeout 600 python tools/bench_gemm_knobs.py --split > $out/gemm_split_fence_$i.log 2>&1
      WAN_HIP_LIB=$ALT timeout 600 python tools/bench_gemm_knobs.py --split > $out/gemm_split_sc1_$i.log 2>&1
    done
    for f in $out/gemm_split_fence_1.log $out/gemm_split_sc1_1.log $out/gemm_split_fence_2.log $out/gemm_split_sc1_2.log; do echo "== $f"; grep -v amdgpu.ids $f; done
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
for f in $out/*.err; do [ -f "$f" ] && { echo "== $f"; tail -5 "$f"; }; done; true
