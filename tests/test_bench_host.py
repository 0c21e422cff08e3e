"""CPU: the host arithmetic behind bench.py's JSON line -- workload table, FLOP formula, PMC traffic reader,
timing helper.  (The measured numbers themselves come from the GPU box; these pin what they are divided by.)"""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tokens(wl):
    return (wl["fs"] + wl["g"] + wl["ft"]) * (wl["h"] // 2) * (wl["w"] // 2)


def test_workloads_are_the_baseline_configs():
    w = bench.WORKLOADS
    assert tokens(w["14b-cof"]) == 67080 and tokens(w["1.3b-cof"]) == 67080          # [src 21 | ground 1 | tgt 21] x 30 x 52
    assert tokens(w["1.3b-small"]) == 2304                                            # configs[0]: 9 x 32 x 32 latent
    assert tokens(w["14b-720p"]) == 75600                                             # configs[3]: 21 x 45 x 80
    assert tokens(w["14b-cof-321f-720p"]) == 586800                                   # configs[4]
    assert tokens(w["14b-cof-720p"]) == 154800                                        # configs[3] in the CoF layout, per sample
    # the reference's demo configuration (scripts/obj_rem.sh:13: --num_frames 33 --source_frames 33 --reasoning_frames 4):
    # condition_count = (33 - 1) // 4 + 1 = 9 source latents, G = (4 - 1) // 4 + 1 = 1, 9 target latents (pipeline_wan.py:411-417)
    w33 = w["14b-cof-33f"]
    assert (w33["fs"], w33["g"], w33["ft"]) == ((33 - 1) // 4 + 1, (4 - 1) // 4 + 1, (33 - 1) // 4 + 1) and tokens(w33) == 19 * 30 * 52
    for name in ("14b-cof", "14b-t2v", "14b-720p", "14b-cof-321f-720p"):
        assert (w[name]["dim"], w[name]["ffn_dim"], w[name]["num_heads"], w[name]["num_layers"]) == (5120, 13824, 40, 40)
    assert (w["1.3b-cof"]["dim"], w["1.3b-cof"]["ffn_dim"], w["1.3b-cof"]["num_heads"], w["1.3b-cof"]["num_layers"]) == (1536, 8960, 12, 30)
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert len(json.load(f)["configs"]) == 5


def test_flop_formula_at_the_north_star_shape():
    total, attn = bench.dit_flops(67080, 5120, 13824, 40)
    # one forward = 5.32 PFLOP, 70 % of it self-attention; one self-attention launch = 4 L^2 C = 9.215e13
    assert abs(total / 5.32e15 - 1) < 5e-3
    assert abs(attn / 40 / 9.215e13 - 1) < 1e-3
    assert 0.68 < attn / total < 0.72


def test_pmc_traffic_reads_the_newest_committed_profile():
    total, d = bench.pmc_traffic("14b-cof", 1)
    assert d is not None and d["source"].startswith("profiles/r") and d["source"].endswith("bench14b_pmc_summary.json")
    with open(os.path.join(ROOT, d["source"])) as f:
        raw = max((v for k, v in json.load(f).items() if "attn_fwd" in k and "<0" in k), key=lambda v: v["fetch"]["avg_ms"])
    assert d["fetch_bytes_x2_corrected"] == raw["fetch"]["avg_counter"] * 1024 * 2        # the guide's gfx950 FETCH_SIZE correction
    assert d["write_bytes"] == raw["write"]["avg_counter"] * 1024
    assert total == d["fetch_bytes_x2_corrected"] + d["write_bytes"]
    assert d["algorithmic_bytes"] == 4 * 67080 * 5120 * 2                                   # q, k, v read + o written, bf16
    assert 1.0 < total / d["algorithmic_bytes"] < 10.0
    assert bench.pmc_traffic("14b-cof", 8) == (None, None) and bench.pmc_traffic("1.3b-small", 1) == (None, None)
    # the profile is attributed only to the kernel it was taken of: variant 2 (max-free attempt) is what the committed pass profiled;
    # a run that launched the lazy reference (variant 1) or an fp8 family gets null + the reason, never a stale number
    tot2, d2 = bench.pmc_traffic("14b-cof", 1, 2 | 16 | 32)
    assert tot2 == total and d2["profiled_kernel"].startswith(bench.PMC_KERNEL_OF_VARIANT[2])
    tot1, d1 = bench.pmc_traffic("14b-cof", 1, 1)
    assert tot1 is None and "stale profile" in d1["note"]
    tot5, d5 = bench.pmc_traffic("14b-cof", 1, 5)
    assert tot5 is None and "no committed PMC pass" in d5["note"]


def test_exposed_comm_split_and_box_object():
    """The N > 1 line's per-exchange split (tags set by the model's `_comm_pair`) and the box fingerprint object."""
    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t
    comm = []
    for _ in range(2 * 3):                                   # 2 steps x 3 layers
        comm += [("q_g0", Ev(0.0), Ev(1.5)), ("o_g1", Ev(0.0), Ev(0.5))]
    comm += [("all_gather", Ev(0.0), Ev(0.25)), ("all_gather", Ev(1.0), Ev(1.25))]
    d = bench.exposed_comm(comm, steps=2, layers=3)
    assert d["per_step_by_exchange"] == {"all_gather": 0.25, "o_g1": 1.5, "q_g0": 4.5}
    assert d["per_step"] == 6.25 and abs(d["per_layer"] - 6.25 / 3) < 1e-3
    assert bench.exposed_comm([], 2, 3) is None and bench.exposed_comm(None, 2, 3) is None
    assert bench.box_object(None) is None
    probe = lambda tf, tb: {"mfma_mix_tflops": tf, "copy_tbps": tb, "mfma_ms": 200.0, "copy_ms": 9.0}
    b = bench.box_object({"before": probe(1520.0, 5.0), "after": probe(1480.0, 5.2)})
    assert b["mfma_mix_tflops"] == 1500.0 and b["copy_tbps"] == 5.1
    assert b["rel_to_reference"] == round(1500.0 / bench.BOX_REFERENCE_MFMA_MIX_TFLOPS, 4)


def test_timing_helper_and_host_threads():
    calls = []
    best, mean, ts = bench._time_reps(lambda: calls.append(1), 3, 1e-3)
    assert 3 <= len(ts) <= 8 and len(calls) == len(ts) and 0 <= best <= max(ts)   # every call is timed, >= min_reps of them
    import time
    best, mean, ts = bench._time_reps(lambda: time.sleep(0.02), 3, 0.01)
    assert len(ts) == 1 and best == mean >= 0.02                      # a first call over budget is the single sample
    assert 1 <= bench.host_threads() <= (os.cpu_count() or 1)


def test_self_spawn_builds_a_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run on 127.0.0.1 (the
    driver's first multi-GPU run must not die on a missing launcher)."""
    import subprocess
    import sys
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 0
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    assert bench.self_spawn(4) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_profiled_kernel_names_are_the_dispatchers_choice():
    """The committed rocprofv3 summary of the headline bench must show the kernels the dispatcher picks TODAY for the 14B
    shapes (wan_gemm_plan / wan_attention_plan: host arithmetic of the shipped library): a profile taken before a dispatch
    change no longer documents the tree."""
    import csv
    import glob
    from videocof_amd import _lib
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench14b_kernel_stats.csv")))
    assert paths
    with open(paths[-1]) as f:
        names = [row["Name"] for row in csv.DictReader(f)]

    def seen(sub):
        return any(sub in n for n in names)
    lib = _lib.load()
    L, C, F, H = 67080, 5120, 13824, 40
    for (M, N, K) in ((L, 2 * C, C), (L, C, C), (L, F, C), (L, C, F)):
        # the model's Linears bring a workspace (ops.gemm / wan_dit_block_forward -> wan_gemm_bf16_ws): the persistent stream-K kernel;
        # without one the same shapes stay on the 4-wave one-workgroup-per-tile kernel
        kernel = _lib.GEMM_VARIANT_KERNELS[lib.wan_gemm_ws_plan(M, N, K)]
        assert kernel == "gemm_pk_kernel" and seen(kernel), (M, N, K, kernel, paths[-1])
        assert _lib.GEMM_VARIANT_KERNELS[lib.wan_gemm_plan(M, N, K)] == "gemm_w4_kernel" and lib.wan_gemm_workspace_bytes(M, N, K) > 0
    ws = lib.wan_attention_workspace_bytes(1, L, L, H, 128)
    v = lib.wan_attention_plan(1, L, L, H, 128, _lib.ATTN_Q_PRESCALED, ws)
    assert v & 15 == 2 and v & _lib.ATTN_VARIANT_XCD_PINNED and v & _lib.ATTN_VARIANT_SPLIT_TAIL       # max-free attempt + fix-up
    # (template arguments: VARIANT, SPLIT, REF, FIX, QK8, PERSIST)
    for sub in ("attn_fwd_w4_kernel<0, false, 0, false, false, false>", "attn_fwd_w4_kernel<0, false, 1, true, false, false>",
                "attn_fwd_w4_kernel<0, true, 1, false, false, false>", "attn_combine_kernel"):
        assert seen(sub), (sub, paths[-1])
    vc = lib.wan_attention_plan(1, L, 512, H, 128, _lib.ATTN_Q_PRESCALED, lib.wan_attention_workspace_bytes(1, L, 512, H, 128))
    # cross-attention: one lazy-reference launch, on the PERSISTENT form (10 520 query blocks > 256 CUs; tuning key attn_persist = 1)
    assert vc & 15 == 1 and lib.wan_get_tuning(b"attn_persist") == 1 and seen("attn_fwd_w4_kernel<1, false, 1, false, false, true>")
    # the lossy mode's profile (bench.py --fp8 --fp8-layers qkv,ffn,o,cross,attn,attn_pv): the all-fp8 attention kernel and its fix-up launch
    # (the fp8-QK^T lazy form), the split tail, the V^T MX quantiser, the e4m3 instantiation of the persistent stream-K GEMM and the K-smoothing kernels
    v8 = lib.wan_attention_plan(1, L, L, H, 128, _lib.ATTN_Q_PRESCALED | 2, ws)
    assert v8 & 15 == 4 and v8 & _lib.ATTN_VARIANT_SPLIT_TAIL
    vf8 = lib.wan_attention_plan(1, L, L, H, 128, _lib.ATTN_Q_PRESCALED | 2 | 4, ws)        # WAN_ATTN_QK_FP8 | WAN_ATTN_PV_FP8
    assert vf8 & 15 == 5 and vf8 & _lib.ATTN_VARIANT_SPLIT_TAIL and vf8 & _lib.ATTN_VARIANT_XCD_PINNED
    p8 = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench14b_fp8_everything_kernel_stats.csv")))
    assert p8
    with open(p8[-1]) as f:
        names8 = [row["Name"] for row in csv.DictReader(f)]
    for sub in ("attn_fwd_f8_kernel", "attn_fwd_w4_kernel<0, false, 1, true, true, false>", "attn_fwd_w4_kernel<0, true, 1, false, true, false>",
                "vt_quantize_mx_kernel", "gemm_pk_kernel<3, 1, true>", "gemm_pk_kernel<1, 1, true>", "col_partial_sums_kernel", "qk_quantize_fp8_kernel"):
        assert any(sub in n for n in names8), (sub, p8[-1])
    # an 8-way Ulysses rank's launches (5 local heads as head groups of 2 and 3: 2.05 / 3.08 rounds of 1 049 KV tiles) and the whole
    # 5-head shard take the max-free attempt too -- few rounds, but milliseconds of launch (bench.py --emulate-sp 8 found them on the lazy
    # form); short key streams on few rounds stay on the one-launch lazy form
    for hl in (2, 3, 5):
        wsl = lib.wan_attention_workspace_bytes(1, L, L, hl, 128)
        assert lib.wan_attention_plan(1, L, L, hl, 128, _lib.ATTN_Q_PRESCALED, wsl) & 15 == 2, hl
    ws_s = lib.wan_attention_workspace_bytes(1, 2304, 2304, 12, 128)
    assert lib.wan_attention_plan(1, 2304, 2304, 12, 128, _lib.ATTN_Q_PRESCALED, ws_s) & 15 == 1          # configs[0]: 108 workgroups x 36 tiles
    # without scratch, or with plain q: one lazy launch (the packed-shift form for plain q)
    assert lib.wan_attention_plan(1, L, L, H, 128, _lib.ATTN_Q_PRESCALED, 0) & 15 == 1
    assert lib.wan_attention_plan(1, L, L, H, 128, 0, ws) & 15 == 1
    assert lib.wan_attention_plan(1, L, L, H, 64, 0, ws) == 0


def test_committed_ingest_measurement_is_quoted_with_its_source():
    d = bench.committed_ingest()
    assert d is not None and d["source"].startswith("profiles/r") and d["measured_in_this_run"] is False
    assert d["check_ok"] is True and 25 < d["checkpoint_GB"] < 32 and d["load_s"] > 0 and d["merge_s"] > 0


def test_split_normalisation_tightens_the_committed_lines():
    """`value_normalised_split` (attention time x probe, the rest x probe^0.3) over the single-GPU headline lines committed under
    profiles/r06: a tighter spread than `value_normalised`, which is tighter than the raw values; equal to the raw value at the reference probe."""
    import glob
    vals = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r06", "bench_14b*.json"))):
        with open(f) as fh:
            d = json.load(fh)
        cfg = d.get("config", {})
        if cfg.get("parallelism") != "single" or "VideoCoF layout, 4-step, 81f@480p" not in cfg.get("workload", "") or d.get("dtype") != "bf16":
            continue
        if d.get("attn_stress") or d.get("graph") or not d.get("box") or not d.get("roofline") or d["steps"] < 4:
            continue
        wall = d["ms_per_step"] * 1e-3 * d["steps"]
        v = bench.normalised_split(67080 * d["steps"], wall, d["steps"], 40, d["roofline"], d["box"])
        vals.append((v, d["value_normalised"], d["value"]))
    assert len(vals) >= 10
    spread = lambda i: max(v[i] for v in vals) / min(v[i] for v in vals) - 1
    assert spread(0) < 0.025 < spread(1) < spread(2)
    roof, box = {"avg_ms": 60.0}, {"before": {"mfma_mix_tflops": 1500.0}, "after": {"mfma_mix_tflops": 1500.0}}
    assert bench.normalised_split(67080 * 4, 15.2, 4, 40, roof, box) == round(67080 * 4 / 15.2, 1)
    assert bench.normalised_split(1, 1.0, 4, 40, None, box) is None and bench.normalised_split(1, 1.0, 4, 40, roof, None) is None
