#!/usr/bin/env python3
"""Benchmark of the hot path: VideoCoF 4-step denoising on the HIP-backed Wan2.1 DiT.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 14b-cof]

A *step* is one denoise step of the reference loop (pipeline_wan.py:694-740): one DiT forward
over the whole CoF token slab + `noise_pred[:, :, :cc] = 0` + one UniPC update.  Inputs are
synthetic (random-init weights of the named architecture, N(0,1) latents and text embeddings,
SURVEY.md section 8d) and are resident in HBM before the timed region.  `value` is whole-job
latent tokens/s = B * L * K / wall with L = ALL DiT tokens per sample (the quantity the FLOP
formula is written in); `denoised_only_tokens_per_s` scales it by (G+Ft)/(Fs+G+Ft).

N > 1: one process per GPU (fast_infer.py:218-222 runs under torchrun).  Either launched as `python -m
torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, or as plain `python bench.py --gpus N`, which
re-executes itself under torch.distributed.run (free port on 127.0.0.1).  The SAME video is sequence-sharded over
the N GPUs with Ulysses head all-to-all on RCCL (strong scaling), or `--mode dp` runs N independent replicas (weak
scaling; what the reference CLI does, fast_infer.py:272).  The N > 1 line adds `ranks_seen`, `exposed_comm_ms`
(HIP events around the exchanges the projections do not cover: the waits before attention and the inverse exchange)
and the roofline of the local head shard.

Rank 0 prints ONE JSON line; it also carries
  roofline      -- the dominant kernel (self-attention flash kernel): algorithmic FLOP per launch
                   / mean launch duration measured with HIP events on the launch stream inside
                   the timed region, vs the bf16 dense MFMA peak (2.5 PFLOP/s);
  cpu_baseline  -- the CPU oracle (a port of the reference's math, oracle/wan_oracle.py) timed on
                   the host cores at the three points BASELINE.md section 4 names and extrapolated to
                   the workload with the FLOP formula (N=1, rank 0 only);
  parity        -- the last block + head of one more (untimed) forward of the same model and inputs,
                   compared on sampled token rows with the fp32 oracle evaluated by torch on the GPU; under
                   sequence parallelism every rank's probe is gathered first, so the N > 1 line checks its
                   own sharded forward (exchanges, rank-offset RoPE, padded keys, all-gather);
  box           -- the box fingerprint: `wan_box_probe` (a fixed calibration workload inside libwan_hip.so)
                   before and after the timed region, and the chip's clock / power / temperature sampled
                   during it; `value_normalised` = value x (reference probe rate / this box's): the figure to
                   compare across boxes and rounds (the chips are power-limited here and differ by ~4 %);
  rank_wall_s   -- (N > 1) every rank's own wall clock of the timed region and max / min.
`--force-sp` runs the whole sequence-parallel line over a 1-rank RCCL group (code path, not scaling).
`--cfg S` runs every step with classifier-free guidance as WanPipeline does (one forward over [uncond, cond], B = 2; BASELINE configs[3]).
`--emulate-sp P` (one GPU) runs what ONE rank of a P-way Ulysses group computes -- the shard shapes of every kernel, device-local
copies where the exchanges would be (videocof_amd.dist.EmulatedRank) -- and prints a PROJECTION line: the compute-side bound of a
P-GPU run, not a measurement of one (`projection` object; `parity` is skipped, the output is not a latent).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # dense, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (dim, ffn, heads, layers, latent frames (Fs, G, Ft) or F, h, w, text_len_used, description)
    "14b-cof": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, fs=21, g=1, ft=21, h=60, w=104,
                    desc="Wan2.1-T2V-14B + VideoCoF layout, 4-step, 81f@480p (BASELINE configs[2])"),
    "1.3b-cof": dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30, fs=21, g=1, ft=21, h=60, w=104,
                     desc="Wan2.1-T2V-1.3B + VideoCoF layout, 4-step, 81f@480p (BASELINE configs[1])"),
    "14b-t2v": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, fs=0, g=0, ft=21, h=60, w=104,
                    desc="Wan2.1-T2V-14B plain T2V layout, 81f@480p"),
    "14b-720p": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, fs=0, g=0, ft=21, h=90, w=160,
                     desc="Wan2.1-T2V-14B plain T2V layout, 81f@720p (BASELINE configs[3] shape; 50-step in the reference)"),
    "14b-cof-33f": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, fs=9, g=1, ft=9, h=60, w=104,
                        desc="Wan2.1-T2V-14B + VideoCoF layout, 4-step, 33 source frames @ 480x832 -- the demo configuration of the reference "
                             "(scripts/obj_rem.sh:13 --num_frames 33 --source_frames 33 --reasoning_frames 4), the one its README quotes "
                             "'~30s/video on H100' for (README.md:124; the demo runs at the clip's native resolution, unstated)"),
    "14b-cof-720p": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, fs=21, g=1, ft=21, h=90, w=160,
                         desc="Wan2.1-T2V-14B + VideoCoF layout, 81f@720p (BASELINE configs[3] shape in the CoF layout, one sample: "
                              "L = 154 800)"),
    "14b-cof-321f-720p": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, fs=81, g=1, ft=81, h=90, w=160,
                              desc="Wan2.1-T2V-14B + VideoCoF layout, 321f@720p length extrapolation (BASELINE configs[4] shape; "
                                   "586 800 tokens -- meant for --gpus 8)"),
    "1.3b-small": dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30, fs=4, g=1, ft=4, h=32, w=32,
                       desc="Wan2.1-T2V-1.3B dims on a 9x32x32 latent (BASELINE configs[0] shape, plumbing)"),
}


def dit_flops(L, C, ffn, layers, Lc=512):
    """Algorithmic FLOPs of one forward."""
    lin = layers * ((12 * C * C + 4 * C * ffn) * L + 4 * C * C * Lc)
    attn = layers * (4 * L * L * C + 4 * L * Lc * C)
    return lin + attn + 4 * L * 64 * C, layers * 4 * L * L * C


# committed PMC summaries by workload (tools/profile_bench.sh -> tools/pmc_summary.py), and the kernel instantiation each dispatcher
# variant code (wan_get_tuning("last_attn_variant") & 15) launches as its MAIN self-attention kernel
PMC_SUMMARY_OF_WORKLOAD = {"14b-cof": "bench14b_pmc_summary.json", "14b-cof-720p": "bench14b_cof_720p_pmc_summary.json",
                           "14b-cof-321f-720p": "bench14b_cof_321f_720p_pmc_summary.json", "14b-cof-33f": "bench14b_cof_33f_pmc_summary.json"}
PMC_KERNEL_OF_VARIANT = {1: "attn_fwd_w4_kernel<0, false, 1,", 2: "attn_fwd_w4_kernel<0, false, 0,"}


def pmc_traffic(workload, shards, variant_code=None):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (separate --pmc FETCH_SIZE / WRITE_SIZE runs, tools/profile_bench.sh -> tools/pmc_summary.py; FETCH_SIZE x2 per the
    gfx950 correction of the microarch guide).  Counters cannot be read live from inside the process, so this is null
    for shapes that were not profiled -- and null when the kernel the dispatcher launched in THIS run (`variant_code`) is not
    the kernel the committed pass profiled (a stale profile must not be attributed to a new kernel).
    ONE source of truth per workload: the newest profiles/r*/<PMC_SUMMARY_OF_WORKLOAD[workload]>."""
    if workload not in PMC_SUMMARY_OF_WORKLOAD or shards != 1:
        return None, None
    import glob
    wl = WORKLOADS[workload]
    L = (wl["fs"] + wl["g"] + wl["ft"]) * (wl["h"] // 2) * (wl["w"] // 2)
    alg = 4 * L * wl["dim"] * 2
    want = None if variant_code is None else PMC_KERNEL_OF_VARIANT.get(int(variant_code) & 15)
    if variant_code is not None and want is None:
        return None, {"note": f"no committed PMC pass for attention variant {int(variant_code) & 15}"}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", PMC_SUMMARY_OF_WORKLOAD[workload])), reverse=True):
        try:
            with open(path) as f:
                cands = {k: v for k, v in json.load(f).items() if "attn_fwd" in k and "<0" in k and "fetch" in v}
            name, d = max(cands.items(), key=lambda kv: kv[1]["fetch"]["avg_ms"])        # the self-attention MAIN launch (not tail / fix-up)
            if want is not None and not name.startswith(want):
                return None, {"note": f"stale profile: {os.path.relpath(path, ROOT)} profiled {name}, this run launched {want}...>"}
            fetch = d["fetch"]["avg_counter"] * 1024 * 2
            # (a WRITE_SIZE pass may be missing for the longest workloads: the output is L x C bf16 = a quarter of the algorithmic bytes)
            write = d["write"]["avg_counter"] * 1024 if "write" in d else L * wl["dim"] * 2
            return fetch + write, {"fetch_bytes_x2_corrected": fetch, "write_bytes": write, "write_measured": "write" in d, "algorithmic_bytes": alg,
                                   "note": "L2->fabric requests; includes Infinity-Cache hits", "profiled_kernel": name,
                                   "source": os.path.relpath(path, ROOT)}
        except Exception:
            continue
    return None, None


# The box fingerprint (wan_box_probe, include/wan_hip.h): a FIXED calibration workload run before and after the timed region.  The
# workload is MFMA-bound and the chips of this pool are power-limited in it, so the clock a box holds -- which differs by ~3.5 %
# between boxes, more than a round of kernel work moves the headline -- shows up 1:1 in `value`; `value_normalised` = value x
# (reference probe rate / this box's probe rate) is the figure to compare ACROSS boxes and rounds.  The reference rate is a constant
# (the mean over the round-6 boxes, profiles/r06/box_probe_*.json), not a peak.
BOX_REFERENCE_MFMA_MIX_TFLOPS = 1500.0


class BoxSampler:
    """Samples the GPU's own telemetry (sysfs hwmon of the amdgpu card: gfx clock, socket power, junction temperature) every 100 ms
    from a background thread while the timed region runs -- what the chip actually held during THIS measurement, next to what the
    calibration probe reached before and after it.  File reads only (no subprocess, nothing enqueued on the GPU); absent or unreadable
    files simply leave the fields out."""

    def __init__(self, device_index=0, period_s=0.1):
        import glob
        import threading
        self.period, self.samples, self._stop, self._thread = period_s, {"sclk_mhz": [], "power_w": [], "temp_c": []}, threading.Event(), None
        self.files = {}
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
        want = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            want = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        pick = None
        for h in cards:
            dev_dir = os.path.realpath(os.path.join(h, "..", ".."))
            if want and os.path.basename(dev_dir).lower() == want.lower():
                pick = h
        if pick is None and len(cards) >= 1:
            pick = cards[min(device_index, len(cards) - 1)]
        if pick:
            for key, names in (("sclk_mhz", ("freq1_input",)), ("power_w", ("power1_input", "power1_average")), ("temp_c", ("temp2_input", "temp1_input"))):
                for n in names:
                    f = os.path.join(pick, n)
                    if os.path.exists(f):
                        self.files[key] = f
                        break
        self.source = pick

    def _read(self):
        for key, f in self.files.items():
            try:
                with open(f) as fh:
                    v = float(fh.read().strip())
                self.samples[key].append(v / 1e6 if key in ("sclk_mhz", "power_w") else v / 1e3)
            except Exception:
                pass

    def start(self):
        import threading
        if not self.files:
            return self

        def loop():
            while not self._stop.wait(self.period):
                self._read()
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2)
        out = {"source": self.source, "samples": max((len(v) for v in self.samples.values()), default=0)}
        for key, v in self.samples.items():
            if v:
                out[key] = {"mean": round(sum(v) / len(v), 1), "min": round(min(v), 1), "max": round(max(v), 1)}
        return out


def box_object(box):
    if box is None:
        return None
    b, a = box["before"], box["after"]
    mean = 0.5 * (b["mfma_mix_tflops"] + a["mfma_mix_tflops"])
    return {"mfma_mix_tflops": round(mean, 1), "copy_tbps": round(0.5 * (b["copy_tbps"] + a["copy_tbps"]), 3),
            "before": b, "after": a, "telemetry_during_timed_region": box.get("telemetry"),
            "reference_mfma_mix_tflops": BOX_REFERENCE_MFMA_MIX_TFLOPS,
            "rel_to_reference": round(mean / BOX_REFERENCE_MFMA_MIX_TFLOPS, 4),
            "what": "wan_box_probe on rank 0: 32x32x16 bf16 MFMAs + LDS fragment reads + softmax VALU stream on random operands, one "
                    "4-wave workgroup per CU, ~0.6 s measured after ~0.3 s of the same (chip at its power limit), and a 256 MiB copy; "
                    "before = after the warm-up steps, after = right after the timed region; the same kernels every round"}


def normalised_split(tokens, wall, steps, layers, roof, box):
    """`value_normalised` scales the WHOLE step by the probe, but only the self-attention launches follow the probe one to one (they run
    its instruction mix: 0.97-0.98 of its rate on every box); the rest of a step -- Linears on another MFMA shape, HBM-bound rows --
    follows it weakly.  Fitted on the 18 single-GPU headline lines of round 6 (10 boxes, probe 1 364 ... 1 532; profiles/r06/bench_14b*.json):
    rest ~ probe^-0.3.  This figure normalises the two parts separately -- measured attention time x probe / reference, the remainder x
    (probe / reference)^0.3 -- and spreads 2.5 % over those lines where `value_normalised` spreads 3.4 % (raw: 10 %); it does not
    over-correct on boxes far from the reference.  None without kernel events or probe."""
    if box is None or not roof or not roof.get("avg_ms"):
        return None
    rel = 0.5 * (box["before"]["mfma_mix_tflops"] + box["after"]["mfma_mix_tflops"]) / BOX_REFERENCE_MFMA_MIX_TFLOPS
    attn_s = roof["avg_ms"] * 1e-3 * layers * steps
    if not 0.0 < attn_s < wall:
        return None
    return round(tokens / (attn_s * rel + (wall - attn_s) * rel ** 0.3), 1)


def committed_ingest():
    """What precedes the first edit of a process: checkpoint ingest + the three LoRA merges of fast_infer.py:366-386 at the real size
    (28.6 GB of bf16 safetensors shards, three rank-128 LoRA files), measured by tools/bench_ingest.py on a GPU box and committed -- it
    writes 32 GB of synthetic files first, too long for the default bench run, so the line quotes the newest committed measurement."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "ingest_14b.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            return {"load_s": d["load_s"], "merge_s": d["merge_s"], "checkpoint_GB": round(d["checkpoint_bytes"] / 1e9, 1),
                    "lora_GB": round(d["lora_bytes"] / 1e9, 2), "check_ok": d["check"]["ok"], "measured_in_this_run": False,
                    "source": os.path.relpath(path, ROOT), "how": "python tools/bench_ingest.py (page-cache reads; synthetic values, real layout)"}
        except Exception:
            continue
    return None


def host_threads():
    """Threads this process may really use: the affinity mask and the cgroup CPU quota, not os.cpu_count() (a
    container on a 256-thread host with a smaller quota oversubscribes badly when torch is given all 256)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def _time_reps(fn, min_reps, budget_s):
    """Wall times of repeated calls.  Every call is timed, the first (cold caches, first-touch allocations) included and
    labelled; if it alone exceeds the budget -- a slow host -- it is the single sample; otherwise min_reps - 1 further
    repetitions (more while the budget lasts, at most 8 in all).  Returns (best, mean of the warm ones, all times)."""
    ts = []
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        spent = time.perf_counter() - t_start
        if len(ts) == 1 and ts[0] > budget_s:
            break
        if len(ts) >= min_reps and (spent >= budget_s or len(ts) >= 8):
            break
    warm = ts[1:] or ts
    return min(ts), sum(warm) / len(warm), ts


def cpu_baseline(wl, L_target, budget_s=30.0):
    """the CPU oracle (port of the reference's fp32 math, oracle/wan_oracle.py) on the host
    cores, at MEASURED points -- (a) BASELINE configs[0] end to end (1.3B dims, 9x32x32 latent, L = 2 304), (b) one
    block of the workload's width at L = 2 304, 4 096 and 8 192, >= 3 timed calls each -- and an EXTRAPOLATION of (b) to the
    workload's token count with the section-8d FLOP formula: per-layer time = lin_flop(L) / R_lin + attn_flop(L) /
    R_attn, the two rates fitted (least squares) to the three block points (the share of attention grows with L, so one
    rate for both would flatter the CPU at L = 2 304 and penalise it at L = 67 080).  The spread of the extrapolation =
    the three values the three PAIRS of points give."""
    from oracle import wan_oracle as O
    cores = host_threads()
    torch.set_num_threads(cores)
    C, ffn, H, layers = wl["dim"], wl["ffn_dim"], wl["num_heads"], wl["num_layers"]
    g = torch.Generator().manual_seed(0)
    cfg = O.DiTConfig(dim=C, ffn_dim=ffn, num_heads=H, num_layers=1)
    sd, p = {}, "blocks.0."
    for attn in ("self_attn", "cross_attn"):
        for lin in "qkvo":
            sd[f"{p}{attn}.{lin}.weight"] = torch.randn(C, C, generator=g) * (1.0 / math.sqrt(C))
            sd[f"{p}{attn}.{lin}.bias"] = torch.zeros(C)
        sd[f"{p}{attn}.norm_q.weight"] = torch.ones(C)
        sd[f"{p}{attn}.norm_k.weight"] = torch.ones(C)
    sd[p + "norm3.weight"], sd[p + "norm3.bias"] = torch.ones(C), torch.zeros(C)
    sd[p + "ffn.0.weight"], sd[p + "ffn.0.bias"] = torch.randn(ffn, C, generator=g) / math.sqrt(C), torch.zeros(ffn)
    sd[p + "ffn.2.weight"], sd[p + "ffn.2.bias"] = torch.randn(C, ffn, generator=g) / math.sqrt(ffn), torch.zeros(C)
    sd[p + "modulation"] = torch.randn(1, 6, C, generator=g) / math.sqrt(C)
    e0 = torch.randn(6, C, generator=g) * 0.1
    ctx = torch.randn(512, C, generator=g)
    ang = O.rope_angles(128)

    def lin_flop(L):
        return (12 * C * C + 4 * C * ffn) * L + 4 * C * C * 512

    def attn_flop(L):
        return 4 * L * L * C + 4 * L * 512 * C

    def rnd(ts):
        return [round(t, 3) for t in ts]

    points = []
    with torch.no_grad():
        for grid, share in (((9, 16, 16), 0.12), ((8, 32, 16), 0.2), ((8, 32, 32), 0.4)):
            L = grid[0] * grid[1] * grid[2]
            x = torch.randn(L, C, generator=g)
            best, mean, ts = _time_reps(lambda: O.block_forward(x, e0, ctx, sd, 0, cfg, grid, ang, 4, (4, 5), L), 3,
                                        budget_s * share)
            points.append({"L": L, "s_per_block_min": round(best, 4), "s_per_block_mean_warm": round(mean, 4), "reps": len(ts),
                           "s_all": rnd(ts), "gflops": round((lin_flop(L) + attn_flop(L)) / best / 1e9, 1)})
        # (a) configs[0] end to end: the real 1.3B architecture on a 9x32x32 latent
        from videocof_amd.weights import dit_param_shapes
        cfg0 = O.DiTConfig(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30)
        sd0 = {k: torch.randn(v, generator=g) * (0.02 if len(v) > 1 else 0.1)
               for k, v in dit_param_shapes(dim=1536, ffn_dim=8960, num_layers=30).items()}
        lat0, ctx0 = torch.randn(1, 16, 9, 32, 32, generator=g), [torch.randn(37, 4096, generator=g)]
        b0, m0, ts0 = _time_reps(lambda: O.dit_forward(sd0, cfg0, lat0, torch.tensor([500]), ctx0, 2304, [4], [(4, 5)]), 3,
                                 budget_s * 1.2)          # ~11 s per call on 16 threads: three calls, never a single sample
    config0 = {"workload": "BASELINE configs[0]: Wan2.1-T2V-1.3B single forward, 9x32x32 latent, L=2304, fp32",
               "s_per_forward_min": round(b0, 3), "s_per_forward_mean_warm": round(m0, 3), "reps": len(ts0), "s_all": rnd(ts0),
               "tokens_per_s": round(2304 / b0, 1)}

    # two-rate model t = lin/R_lin + attn/R_attn: least squares over the block points, and each pair of points on its own
    def solve(pts):
        a = torch.tensor([[lin_flop(q["L"]), attn_flop(q["L"])] for q in pts], dtype=torch.float64)
        b = torch.tensor([q["s_per_block_min"] for q in pts], dtype=torch.float64)
        sc = a.abs().max(dim=0).values                      # column scaling: the two FLOP columns differ by orders of magnitude
        x = torch.linalg.lstsq(a / sc, b[:, None]).solution[:, 0] / sc
        return float(x[0]), float(x[1])

    def per_layer(il, ia):
        return lin_flop(L_target) * il + attn_flop(L_target) * ia

    inv_rlin, inv_rattn = solve(points)
    fit = "two-rate (linear | attention) least-squares fit of the three measured block points"
    if inv_rlin <= 0 or inv_rattn <= 0:        # noisy timing: fall back to one rate from the largest point
        q = points[-1]
        inv_rlin = inv_rattn = q["s_per_block_min"] / (lin_flop(q["L"]) + attn_flop(q["L"]))
        fit = "single rate from the L=8192 point (the two-rate fit was ill-conditioned)"
    t_layer = per_layer(inv_rlin, inv_rattn)
    value = L_target / (t_layer * layers)
    pair_values = []
    for i in range(len(points)):
        for j in range(i + 1, len(points)):
            il, ia = solve([points[i], points[j]])
            if il > 0 and ia > 0:
                pair_values.append(round(L_target / (per_layer(il, ia) * layers), 3))
    resid = [round((lin_flop(q["L"]) * inv_rlin + attn_flop(q["L"]) * inv_rattn) / q["s_per_block_min"] - 1.0, 4) for q in points]
    pts_txt = ", ".join(f"L={q['L']} ({q['reps']} calls, min {q['s_per_block_min']:.2f} s)" for q in points)
    return {"value": round(value, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "extrapolated": True,
            "sample": f"oracle/wan_oracle.py (fp32 torch-CPU port of the reference), {cores} host threads.  MEASURED: one "
                      f"{C}-wide block at {pts_txt}; configs[0] end to end ({len(ts0)} calls, min {b0:.2f} s).  `value` is "
                      f"EXTRAPOLATED to L={L_target} x {layers} layers with the SURVEY 8d FLOP formula, {fit}: "
                      f"{t_layer:.1f} s/layer; pairwise fits give {min(pair_values or [value]):.2f}..{max(pair_values or [value]):.2f} tokens/s.",
            "measured_points": {"block": points, "config0_end_to_end": config0},
            "fit": {"linear_gflops": round(1e-9 / inv_rlin, 1), "attention_gflops": round(1e-9 / inv_rattn, 1),
                    "s_per_layer_at_target": round(t_layer, 2), "s_per_step_at_target": round(t_layer * layers, 1),
                    "relative_residual_at_points": resid,
                    "value_spread_pairwise_fits": [min(pair_values or [value]), max(pair_values or [value])]}}


def verify_last_block(model, wl, lat, t, ctx, seq_len, fsi, gfi, out, L):
    """Parity of the bench's own forward (replaces a bare isfinite check): the residual stream entering the LAST block
    of that forward (kept by the model's probe) goes through the oracle's block + head + unpatchify -- evaluated in
    fp32 on the GPU with torch's kernels, which share nothing with libwan_hip.so -- and is compared with the forward's
    output on sampled token rows (first rows, src|ground|tgt boundaries, the 8-row last query block, a stride)."""
    from oracle import wan_oracle as O
    li = wl["num_layers"] - 1
    cfg = O.DiTConfig(dim=wl["dim"], ffn_dim=wl["ffn_dim"], num_heads=wl["num_heads"], num_layers=wl["num_layers"])
    keep = (f"blocks.{li}.", "head.", "time_", "text_embedding")
    sd = {k: v.detach().float() for k, v in model.state_dict().items() if k.startswith(keep)}
    x_in = model._probe[:L]
    e, e0 = O.time_embed(t.reshape(1).to(x_in.device), sd, cfg)
    cemb = O.text_embed([c.float() for c in ctx], sd, cfg)[0]
    grid = (lat.shape[2], lat.shape[3] // 2, lat.shape[4] // 2)
    fs = fsi[0] if fsi else None
    gr = gfi[0] if gfi else None
    y = O.block_forward(x_in, e0[0], cemb.bfloat16().float(), sd, li, cfg, grid, O.rope_angles(128), fs, gr, L)
    ref = O.unpatchify(O.head_forward(y, e[0], sd, cfg), grid, cfg)               # [16, F, H, W]
    hw = grid[1] * grid[2]
    rows = set(range(8)) | set(range(L - 8, L)) | set(range(0, L, 997))
    if fs:
        rows |= set(range(fs * hw - 4, fs * hw + 4)) | set(range((gr[1] if gr else fs) * hw - 4, (gr[1] if gr else fs) * hw + 4))
    rows = torch.tensor(sorted(r for r in rows if 0 <= r < L), device=x_in.device)
    f_, h_, w_ = rows // hw, (rows // grid[2]) % grid[1], rows % grid[2]

    def pick(v):                      # the 2x2 output pixels x 16 channels of each sampled token
        return torch.stack([v[:, f_, 2 * h_ + i, 2 * w_ + j] for i in (0, 1) for j in (0, 1)])
    got, want = pick(out[0].float()), pick(ref)
    rel = float((got - want).norm() / want.norm())
    cos = float(torch.nn.functional.cosine_similarity(got.flatten(), want.flatten(), dim=0))
    return {"what": "last block + head + unpatchify of the bench's own forward vs the fp32 oracle on the probed residual "
                    "stream, sampled token rows", "rows": int(rows.numel()), "rel_l2": round(rel, 5), "cosine": round(cos, 6),
            "tolerance": {"rel_l2": 1e-2, "cosine": 0.9999}, "ok": bool(rel < 1e-2 and cos > 0.9999)}


def exposed_comm(comm, steps, layers):
    """The N > 1 line's split of the communication time the compute stream actually waited for (rank 0's HIP events, tagged by the
    model: videocof_amd/wan_transformer3d.py `_comm_pair`), per step and per exchange."""
    if not comm:
        return None
    by = {}
    for tag, a, b in comm:
        by[tag] = by.get(tag, 0.0) + a.elapsed_time(b)
    total = sum(by.values())
    return {"per_step": round(total / steps, 3), "per_layer": round(total / steps / layers, 4),
            "per_step_by_exchange": {k: round(v / steps, 3) for k, v in sorted(by.items())},
            "what": "HIP events on rank 0's compute stream around: q_g0 = wait_k / wait_v / V^T unpack / wait for the FIRST head group "
                    "of q before attention (k and V^T travel under the V and q projections, the second head group of q under the first "
                    "group's attention); o_g1 = the inverse exchange of the LAST head group (the first group's leaves under the second "
                    "group's attention); all_gather = the one all-gather of the head output per forward"}


def self_spawn(n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU, rendezvous
    on a free port of 127.0.0.1 (the container hostname may not resolve).  Rank 0's JSON line goes to the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


class _ToyTokenizer:
    """Whitespace tokenizer with the umT5 call signature: the tokenizer is host-side string work, not part of the measured path."""

    def __init__(self, vocab):
        self.vocab = vocab

    def __call__(self, prompt, padding=None, max_length=512, truncation=True, add_special_tokens=True, return_tensors="pt"):
        from types import SimpleNamespace
        ids = torch.zeros(len(prompt), max_length, dtype=torch.long)
        mask = torch.zeros(len(prompt), max_length, dtype=torch.long)
        for b, p in enumerate(prompt):
            toks = [2 + (sum(map(ord, w)) % (self.vocab - 2)) for w in p.split()][: max_length - 1] + [1]
            ids[b, :len(toks)] = torch.tensor(toks)
            mask[b, :len(toks)] = 1
        return SimpleNamespace(input_ids=ids, attention_mask=mask)


def e2e_edit(model, wl, dev):
    """One whole VideoCoF edit the way fast_infer.py:366-420 runs it, on the bench's own DiT: prompt string + 81-frame 480 x 832
    source clip in, grounding + edited clips out -- umT5-XXL encode, WanVAE encode, the 4-step CoF denoise loop through
    ``WanPipeline`` with ITS defaults (cache_context and skip_source_prediction: parity-neutral hoists the headline loop above does
    not use), WanVAE decode of the grounding and edit segments.  Random-init weights of the real architectures, a synthetic clip.
    Synchronised wall seconds per stage from the pipeline's own stage clock, after one 1-step warm-up call; untimed above."""
    from videocof_amd import AutoencoderKLWan, FlowUniPCMultistepScheduler, WanPipeline, WanT5EncoderModel
    from videocof_amd.weights import random_t5_state_dict, random_vae_state_dict
    vae = AutoencoderKLWan()
    vae.load_state_dict(random_vae_state_dict(dev), device=dev)
    tcfg = dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32)
    t5 = WanT5EncoderModel(shared_pos=False, **tcfg)
    t5.load_state_dict(random_t5_state_dict(dev, **tcfg), device=dev)
    model._attn_events = None
    model._comm_events = None
    pipe = WanPipeline(tokenizer=_ToyTokenizer(tcfg["vocab"]), text_encoder=t5, vae=vae, transformer=model,
                       scheduler=FlowUniPCMultistepScheduler(shift=1))
    frames, height, width = (wl["fs"] - 1) * 4 + 1, wl["h"] * 8, wl["w"] * 8
    g = torch.Generator(device=dev).manual_seed(0)
    video = (torch.rand(1, 3, frames, height, width, device=dev, generator=g) * 2 - 1).bfloat16()
    prompt = "remove the red cup from the wooden table and keep everything else unchanged"
    kw = dict(video=video, prompt=prompt, height=height, width=width, source_frames=frames, reasoning_frames=4,
              num_inference_steps=4, guidance_scale=1.0, shift=3, repeat_rope=True, cot=True, generator=g,
              output_type="numpy", return_dict=True)
    pipe(**{**kw, "num_inference_steps": 1})            # warm-up: allocator, workspaces, kernel attributes of the VAE / T5 kernels
    pipe.stage_seconds = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe(**kw)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    st = {k: round(v, 4) for k, v in pipe.stage_seconds.items()}
    pipe.stage_seconds = None
    return {"sec_per_video": round(total, 3), "stages_s": st, "other_s": round(total - sum(st.values()), 4),
            "what": f"VideoCoF edit end to end on one GPU: umT5-XXL + WanVAE encode ({frames}f@{height}x{width}) + WanPipeline 4-step CoF loop "
                    "(guidance 1.0, pipeline defaults: cache_context, skip_source_prediction) + WanVAE decode (grounding + edit)",
            "edit_video_shape": list(out.edit_videos.shape), "ground_video_shape": list(out.ground_videos.shape),
            "finite": bool(np.isfinite(out.edit_videos).all()) if hasattr(out.edit_videos, "shape") else None,
            "data": "synthetic clip, random-init weights; toy whitespace tokenizer"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="14b-cof", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="sp", choices=["sp", "dp"], help="N>1: Ulysses sequence parallel or replicas")
    ap.add_argument("--layers", type=int, default=0,
                    help="PROFILING ONLY: run the first N blocks of the architecture (per-launch counters of a kernel do not depend on "
                         "the layer count; a PMC pass over 40 layers at L = 586 800 would take an hour).  The line says so in "
                         "config.layers_override and its value is never a headline number.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo (host-staged exchanges) only to exercise the N>1 code path on a box "
                         "whose ranks share one GPU -- its numbers are meaningless")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (with --backend gloo)")
    ap.add_argument("--force-sp", action="store_true",
                    help="ONE rank, the whole sequence-parallel line: a 1-rank RCCL process group + model.force_ulysses -- every exchange is a "
                         "real (identity) RCCL all-to-all issued async on the group's stream, the parity probe is gathered through the group, "
                         "the walls through all_gather: the N > 1 code path of this file on a box with one GPU.  Never a headline line.")
    ap.add_argument("--emulate-sp", type=int, default=0, metavar="P",
                    help="one GPU: run what ONE rank of a P-way Ulysses group computes (shard shapes, device-local copies in place of "
                         "the exchanges) and print a PROJECTION of the P-GPU line's compute side; not a measurement of P GPUs")
    ap.add_argument("--sp-head-groups", type=int, default=2, choices=[1, 2],
                    help="sequence parallel: 2 (default) = q and o travel as two head groups, the second under the first group's attention and "
                         "vice versa (two attention launches per layer); 1 = one exchange + one launch (an A/B knob for a real N-GPU box)")
    ap.add_argument("--cfg", type=float, default=0.0, metavar="S",
                    help="classifier-free guidance with scale S > 1 as WanPipeline runs it (pipeline_wan.py:700-731): every step ONE forward "
                         "over the batch [uncond, cond] (B = 2, two prompts), then uncond + S (cond - uncond) -- the per-step work of BASELINE "
                         "configs[3] (inference.py defaults: guidance 5.0); tokens counted = B x L per step")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-box-probe", action="store_true",
                    help="skip the ~0.4 s box fingerprint (wan_box_probe) before and after the timed region")
    ap.add_argument("--no-verify", action="store_true", help="skip the (untimed) oracle check of the last block")
    ap.add_argument("--fp8-layers", default="qkv,ffn",
                    help="with --fp8: comma-separated subset of qkv,ffn,o,cross (all four = every per-token Linear of a block), attn "
                         "(self-attention QK^T on the fp8 matrix pipe, P.V stays bf16) and attn_pv (with attn: P.V in fp8 as well)")
    ap.add_argument("--fp8-no-smooth-k", action="store_true",
                    help="with --fp8-layers ...,attn: quantise k as it is instead of k minus its token mean (saves one 0.5 ms pass per layer)")
    ap.add_argument("--fp8", action="store_true",
                    help="run the q|k, v, ffn.0 and ffn.2 projections in OCP e4m3 (WanTransformer3DModel.enable_fp8_linear): a "
                         "LOSSY option with its own error statement; the line says so in `dtype` and is never the headline")
    ap.add_argument("--attn-stress", action="store_true",
                    help="give the random model checkpoint-like q / k statistics (norm_q / norm_k gains log-normal up to 8, one "
                         "large-bias channel per head: log2-domain scores of several hundred) -- the input class on which a "
                         "max-free attention attempt fails its check; the line reports flagged workgroups, the sticky switch "
                         "and the lazy-reference repair events.  Never the headline line.")
    ap.add_argument("--graph", action="store_true",
                    help="replay the forward + CoF mask from a hipGraph (videocof_amd.GraphedForward; text K/V hoisted out of "
                         "the step as WanPipeline does).  For launch-bound small shapes; never the headline line.")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the (separately timed) end-to-end edit -- umT5 + WanVAE encode + WanPipeline + WanVAE decode -- behind `e2e`")
    ap.add_argument("--graph-loop", action="store_true",
                    help="replay the WHOLE K-step loop (forwards, CoF mask, UniPC updates) from ONE hipGraph "
                         "(videocof_amd.GraphedLoop = WanPipeline(capture_graph='loop')).  For launch-bound small shapes.")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(self_spawn(args.gpus))        # no launcher: become one (the ranks come back through main())
    args.gpus = world
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    from videocof_amd import FlowUniPCMultistepScheduler, WanTransformer3DModel
    from videocof_amd import dist as vdist
    from videocof_amd.weights import random_dit_state_dict

    emu = int(args.emulate_sp) if world == 1 else 0
    if emu and (emu < 2 or args.force_sp or args.graph or args.graph_loop or args.mode != "sp"):
        raise SystemExit("--emulate-sp P: P >= 2, on one GPU, without --force-sp / --graph / --graph-loop / --mode dp")
    force_sp = args.force_sp and world == 1
    if force_sp:
        import socket
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 or force_sp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the collective libraries print connection banners on fd 1 ("[Gloo] Rank 0 is connected to ..."): keep stdout for the
        # ONE JSON line and send whatever they print during set-up to stderr
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if force_sp:
                dist.init_process_group(args.backend, rank=0, world_size=1, **({"device_id": dev} if args.backend == "nccl" else {}))
            elif args.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group("gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    wl = dict(WORKLOADS[args.workload])
    if args.layers > 0:
        wl["num_layers"] = min(args.layers, wl["num_layers"])
    sp = (world > 1 or force_sp or emu > 0) and args.mode == "sp"
    sp_degree = emu if emu else world          # the Ulysses degree the shapes of this run belong to
    # (num_heads % world != 0 -- the 12-head 1.3B model on 8 GPUs -- runs with heads padded to a multiple of the degree:
    # WanTransformer3DModel._pad_heads_for_ulysses; the line says so in config.sp_padded_heads)

    # ---------------- model + synthetic inputs (resident in HBM before timing)
    torch.manual_seed(0)
    model = WanTransformer3DModel(dim=wl["dim"], ffn_dim=wl["ffn_dim"], num_heads=wl["num_heads"],
                                  num_layers=wl["num_layers"])
    shapes = dict(dim=wl["dim"], ffn_dim=wl["ffn_dim"], num_layers=wl["num_layers"])
    sd = random_dit_state_dict(dev, seed=0, **shapes)
    if args.attn_stress:
        gs = torch.Generator(device=dev).manual_seed(123)
        for name, t in sd.items():
            if name.endswith(("self_attn.norm_q.weight", "self_attn.norm_k.weight")):
                t.mul_(torch.exp(0.9 * torch.randn(t.shape, device=dev, generator=gs)).clamp(max=8.0))
            elif name.endswith(("self_attn.q.bias", "self_attn.k.bias")):
                t.view(-1, 128)[:, 5] = 12.0          # same sign for every token: q . k picks up a large common term
    model.load_state_dict(sd, device=dev)
    del sd
    if args.fp8:
        model.enable_fp8_linear(tuple(args.fp8_layers.split(",")), attn_smooth_k=not args.fp8_no_smooth_k)
    if sp:
        if emu:
            vdist.init_sequence_parallel(backend="emulated", rank=0, world_size=emu)
        else:
            vdist.init_sequence_parallel()
        model.enable_multi_gpus_inference()
        model.force_ulysses = force_sp
        model.sp_head_groups = args.sp_head_groups
    Fs, G, Ft = wl["fs"], wl["g"], wl["ft"]
    Ftot = Fs + G + Ft
    cof = Fs > 0
    g = torch.Generator(device=dev).manual_seed(0 if sp else rank)       # fast_infer.py:390 seeds per rank
    latents = torch.randn(1, 16, Ftot, wl["h"], wl["w"], device=dev, generator=g).bfloat16()
    ctx = [torch.randn(37, 4096, device=dev, generator=g).bfloat16()]
    ctx1 = ctx                                   # the one-sample prompt list the untimed parity forward runs with
    cfg_scale = float(args.cfg)
    if cfg_scale and cfg_scale <= 1.0:
        raise SystemExit("--cfg S: S > 1 (guidance 1.0 is the default single-sample step)")
    if cfg_scale and (args.graph or args.graph_loop):
        raise SystemExit("--cfg runs the eager loop")
    nb = 2 if cfg_scale else 1                   # samples per forward
    if cfg_scale:
        ctx = [torch.randn(12, 4096, device=dev, generator=g).bfloat16()] + ctx          # [negative prompt, prompt] (:605-608)
    L = Ftot * (wl["h"] // 2) * (wl["w"] // 2)
    seq_len = L
    fsi = [Fs] * nb if cof else None
    gfi = [(Fs, Fs + G)] * nb if cof else None
    sched = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, solver_order=2)

    prof = [] if not args.no_kernel_events else None
    model._attn_events = prof
    comm = [] if (sp and prof is not None) else None
    model._comm_events = comm

    fwd = model
    if args.graph:
        if sp:
            raise SystemExit("--graph covers the single-device forward")
        from videocof_amd import GraphedForward
        fwd = GraphedForward(model)
        model.mask_source_frames = Fs                # the CoF mask inside the captured unpatchify kernel
        prof = model._attn_events = None             # HIP events cannot be recorded inside a replayed graph

    gloop = None
    if args.graph_loop:
        if sp or args.graph:
            raise SystemExit("--graph-loop covers the single-device forward and excludes --graph")
        from videocof_amd import GraphedLoop
        gloop = GraphedLoop(model)
        model.cache_context = True                   # the text K/V are computed once per loop, inside the graph
        model.mask_source_frames = Fs
        prof = model._attn_events = None

    def run(n_steps):
        nonlocal latents
        sched.set_timesteps(max(n_steps, 1), device=dev, shift=3)

        def loop(lat, c):
            if gloop is not None:
                sched._reset()
                sched.set_begin_index(0)             # no device round trip inside a capture
            for t in sched.timesteps[:n_steps]:
                if cfg_scale:
                    v2 = fwd(torch.cat([lat] * 2), t.expand(2), c, seq_len, frame_split_indices=fsi, ground_frame_indices=gfi)
                    vu, vc = v2.chunk(2)
                    v = vu + cfg_scale * (vc - vu)                                                   # :729-731
                else:
                    v = fwd(lat, t.expand(1), c, seq_len, frame_split_indices=fsi, ground_frame_indices=gfi)
                if cof and not args.graph and gloop is None:
                    v[:, :, :Fs] = 0
                lat = sched.step(v, t, lat, return_dict=False)[0]
            return lat
        if gloop is not None:
            return gloop((n_steps,), latents, ctx, loop, keep=(sched.timesteps,))
        return loop(latents, ctx)

    grouped = world > 1 or force_sp             # a process group exists: barriers, gathered walls, a teardown

    def fence():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
            torch.cuda.synchronize()

    if args.graph_loop:
        run(args.steps)                              # eager call of this signature
        run(args.steps)                              # capture + first replay
    elif args.warmup > 0:
        run(max(args.warmup, 2) if args.graph else args.warmup)      # graph: one eager call + the capture
    if prof is not None:
        prof.clear()
    if comm is not None:
        comm.clear()
    if model._ws_self.buf is not None:
        model._ws_self.buf[8:12].zero_()          # scratch header word [2]: repair events of the lazy softmax reference
    box = None
    if not args.no_box_probe:
        from videocof_amd import ops as vops
        box = {"before": vops.box_probe(dev, target_ms=900)}     # (0.6 s measured after 0.3 s of the same) the warm-up steps above have brought the chip to temperature
    sampler = BoxSampler(local_rank).start() if box is not None else None
    fence()
    t0 = time.perf_counter()
    out = run(args.steps)
    fence()
    wall = time.perf_counter() - t0
    if box is not None:
        box["telemetry"] = sampler.stop()
        box["after"] = vops.box_probe(dev, target_ms=900)
    rank_walls = None
    if grouped:
        # every rank's own wall time of the timed region: the line's time is the MAX; the list and max / min show a straggler
        tws = [torch.zeros(1, device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64) for _ in range(world)]
        dist.all_gather(tws, torch.tensor([wall], device=tws[0].device, dtype=torch.float64))
        rank_walls = [float(v.item()) for v in tws]
        wall = max(rank_walls)
    assert torch.isfinite(out.float()).all(), "non-finite latents"

    # ---------------- parity of this run's own forward (untimed): probe the last block, compare with the oracle.
    # Under sequence parallelism EVERY rank runs the probed forward (it contains collectives); each rank's probe is its token shard
    # of the residual stream entering the last block, gathered over the SP group (the group's own all-gather), and rank 0 checks the
    # sharded forward's output -- exchanges, rank-offset RoPE, padded-key masking, final all-gather and all -- against the oracle's
    # single-device block + head + unpatchify on the gathered stream.
    parity = None
    if emu:
        parity = {"skipped": "--emulate-sp: the arrived operands are this rank's own slabs, the output is not a denoised latent"}
    elif not args.no_verify and (rank == 0 or sp) and not args.graph_loop:
        model._attn_events = None
        model._comm_events = None
        model.mask_source_frames = 0
        model._probe_layer = wl["num_layers"] - 1
        tv = sched.timesteps[:1]
        try:
            v = model(latents, tv.expand(1), ctx1, seq_len, frame_split_indices=fsi and fsi[:1], ground_frame_indices=gfi and gfi[:1])
            if sp:
                grp = vdist.get_sp_group()
                model._probe = grp.all_gather_tokens(model._probe.view(1, -1, wl["dim"]))[0].clone()
        finally:
            model._probe_layer = None
        if rank == 0:
            try:
                parity = verify_last_block(model, wl, latents, tv, ctx1, seq_len, fsi and fsi[:1], gfi and gfi[:1], v, L)
                if sp:
                    parity["what"] = ("sequence-parallel forward over %d ranks: " % world) + parity["what"] + \
                                     " (the probes are the ranks' token shards, gathered)"
                if args.fp8:        # the lossy mode is reported against its own stated bound (tests/test_gpu_fp8.py), not the bf16 one
                    parity["tolerance"] = {"rel_l2": 8e-2, "cosine": 0.995}
                    parity["ok"] = bool(parity["rel_l2"] < 8e-2 and parity["cosine"] > 0.995)
            except Exception as e:      # the check must never take the measured number down with it, but it must be visible
                parity = {"error": repr(e), "ok": False}
        model._probe = None
        model._attn_events = prof
        model._comm_events = comm

    # ---------------- dominant kernel: self-attention launches inside the timed region
    roof = None
    if prof:
        ms = [a.elapsed_time(b) for a, b in prof]
        avg_ms = sum(ms) / len(ms)
        heads_local = (model._sp_pad.H if getattr(model, "_sp_pad", None) is not None else wl["num_heads"]) // (sp_degree if sp else 1)
        Lk = L
        Lq = model._last_attn_rows
        flop = 4.0 * Lq * Lk * heads_local * 128
        ach = flop / (avg_ms * 1e-3) / 1e12
        from videocof_amd import _lib
        vcode = int(model._last_attn_variant)
        traffic, traffic_detail = pmc_traffic(args.workload, sp_degree if sp else 1, vcode)
        hdr = model._ws_self.buf[:16].view(torch.int32).tolist() if model._ws_self.buf is not None else [None] * 4
        repairs = hdr[2]
        roof = {"kernel": "self-attention wan_attention_fwd: " + _lib.attn_variant_name(vcode), "variant_code": vcode,
                "kernel_reported_by": "wan_get_tuning('last_attn_variant') right after the launch",
                "attn_flagged_wgs": hdr[1], "attn_flagged_wgs_of": (model._last_attn_rows + 255) // 256 * heads_local,
                "attn_max_free_attempt_switched_off": None if hdr[0] is None else bool(hdr[0]),
                "attn_repair_events": repairs, "heads_local": heads_local,
                "attn_stats_note": "flagged = workgroups of the LAST self-attention call whose max-free attempt failed its check and "
                                   "were redone by the lazy-reference launch; repair events = wave-level reference raises of the "
                                   "lazy form over the whole timed region",
                "bound": "mfma", "achieved": round(ach, 1),
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                "traffic": traffic, "traffic_detail": traffic_detail, "launches": len(ms),
                "avg_ms": round(avg_ms, 3), "flop_per_launch": flop, "dtype_peak": "bf16 dense MFMA"}
        if box is not None and (vcode & 15) in (1, 2):
            # the same instruction mix (32x32x16 bf16 MFMAs + LDS fragment reads + softmax VALU stream) with nothing else to do, on this box
            # in this run: what the chip SUSTAINS at its power limit -- the ceiling the kernel can be held against next to the 2.5 PF spec
            probe = 0.5 * (box["before"]["mfma_mix_tflops"] + box["after"]["mfma_mix_tflops"])
            roof["sustained_mix"] = {"probe_tflops": round(probe, 1), "frac_of_probe": round(ach / probe, 4),
                                     "what": "wan_box_probe of this run (box.mfma_mix_tflops): the kernel's own instruction mix, no loads from HBM, "
                                             "no barriers, at the chip's power limit; `frac` above stays against the dense bf16 spec peak"}
        if (vcode & 15) == 5:       # both products at the fp8 rate (2 x bf16)
            peak = 2 * PEAK_BF16_TFLOPS
            roof.update(peak=round(peak, 1), frac=round(ach / peak, 4), traffic=None,
                        traffic_detail={"note": "the committed PMC passes are of the bf16 kernel; none was taken for this lossy variant"},
                        dtype_peak="dense fp8 MFMA (2 x bf16): QK^T and P.V both on the fp8 matrix pipe")
        if (vcode & 15) == 4:       # fp8 QK^T: half of the flops run at the fp8 rate (2 x bf16), half (P.V) at the bf16 rate
            peak = 1.0 / (0.5 / (2 * PEAK_BF16_TFLOPS) + 0.5 / PEAK_BF16_TFLOPS)
            roof.update(peak=round(peak, 1), frac=round(ach / peak, 4), traffic=None,
                        traffic_detail={"note": "the committed PMC passes are of the bf16 kernel; none was taken for this lossy variant "
                                                "(its K tiles are half the bytes)"},
                        dtype_peak="harmonic mix: QK^T at the dense fp8 MFMA rate (2 x bf16), P.V at the dense bf16 rate")

    units = world if (world > 1 and not sp) else 1            # dp: every rank denoises its own video
    units *= nb                                                # --cfg: two samples per forward (tokens/s = B x L x steps / wall)
    tokens = units * L * args.steps
    value = tokens / wall
    tot_flop, attn_flop = dit_flops(L, wl["dim"], wl["ffn_dim"], wl["num_layers"])
    res = {
        "metric": "denoised latent tokens/s (4-step 81f@480p Wan2.1 DiT denoise loop, all DiT tokens counted)",
        "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 2), "higher_is_better": True,
        "scaling": "strong" if sp or world == 1 else "weak", "vs_baseline": None,
        "dtype": (f"fp8-e4m3 ({args.fp8_layers}; 'attn' = the QK^T product of self-attention, 'attn_pv' = its P.V product) + bf16 (softmax, the rest); "
                  "LOSSY option, not the headline") if args.fp8 else "bf16",
        "data": "synthetic (random-init weights, N(0,1) latents + text embeddings)",
        "config": {"workload": wl["desc"], "layout": "VideoCoF (src|ground|tgt)" if cof else "T2V",
                   "latent": [1, 16, Ftot, wl["h"], wl["w"]], "grid": [Ftot, wl["h"] // 2, wl["w"] // 2],
                   "tokens_per_sample": L, "global_batch": units, "guidance_scale": cfg_scale or 1.0,
                   "layers_override": (wl["num_layers"] if args.layers > 0 else None),
                   "sp_padded_heads": (model._sp_pad.pad_heads if getattr(model, "_sp_pad", None) is not None else None),
                   "sp_head_groups": args.sp_head_groups if sp else None,
                   "parallelism": ("EMULATED rank 0 of ulysses-sp%d on one GPU (projection)" % emu) if emu else
                                  ("ulysses-sp%d" % world) if sp else ("replicas-dp%d" % world if world > 1 else "single")},
        "tokens_per_s_per_gpu": round(value / world, 1),
        "sec_per_video_4step": round(wall / args.steps * 4, 3),
        "denoised_only_tokens_per_s": round(value * (G + Ft) / Ftot, 1),
        "model_tflops_per_s": round(units * tot_flop * args.steps / wall / 1e12, 1),
        "mfma_frac_whole_step": round(units * tot_flop * args.steps / wall / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
        "roofline": roof,
        "box": box_object(box),
        "value_normalised": None if box is None else round(value * BOX_REFERENCE_MFMA_MIX_TFLOPS / (0.5 * (box["before"]["mfma_mix_tflops"] + box["after"]["mfma_mix_tflops"])), 1),
        "value_normalised_split": normalised_split(tokens, wall, args.steps, wl["num_layers"], roof, box),
        "ranks_seen": dist.get_world_size() if grouped else 1,
        "backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else " (host-staged, numbers meaningless)")
                    + (" -- ONE rank, force_ulysses: identity exchanges, the N > 1 code path only" if force_sp else "")) if grouped else None,
        "exposed_comm_ms": exposed_comm(comm, args.steps, wl["num_layers"]),
        "rank_wall_s": None if rank_walls is None else {
            "per_rank": [round(v, 4) for v in rank_walls], "max_over_min": round(max(rank_walls) / min(rank_walls), 4),
            "what": "each rank's own wall clock of the timed region (between the two barriers); the line's time is the max"},
        "parity": parity,
        "graph": "loop" if args.graph_loop else bool(args.graph),
        "attn_stress": bool(args.attn_stress),
        "fp8_attn_smooth_k": (not args.fp8_no_smooth_k) if (args.fp8 and "attn" in args.fp8_layers.split(",")) else None,
    }
    if emu:
        res["metric"] = "PROJECTION (one rank of a %d-way Ulysses group emulated on one GPU): " % emu + res["metric"]
        res["tokens_per_s_per_gpu"] = round(value / emu, 1)
        res["mfma_frac_whole_step"] = round(units * tot_flop * args.steps / wall / 1e12 / (PEAK_BF16_TFLOPS * emu), 4)
        res["projection"] = {
            "of_n_gpus": emu, "emulated_rank": 0, "ms_per_step_compute_side": round(wall / args.steps * 1e3, 2),
            "what": "ONE GPU ran what rank 0 of %d computes per step -- token-local kernels on L/%d rows, self-attention on H/%d heads over all "
                    "L keys, wire layouts, pack / unpack passes, head-group pipelining -- with device-local copies of the send buffers where "
                    "the RCCL exchanges would be.  `value` = L * steps / wall: the whole-job rate a %d-GPU run reaches if every rank takes "
                    "this long, i.e. its compute-side bound; exposed xGMI time comes on top.  No byte crossed a link; the output is not a "
                    "latent (parity skipped)." % (emu, emu, emu, emu)}
    if rank == 0 and world == 1 and not force_sp and not emu and not cfg_scale and not args.no_e2e and args.workload in ("14b-cof", "14b-cof-33f") and not (args.fp8 or args.attn_stress or args.graph or args.graph_loop):
        try:        # the metric's second half (sec / video of a whole edit); separate from the timed region above, never takes it down
            res["e2e"] = e2e_edit(model, wl, dev)
        except Exception as e:
            res["e2e"] = {"error": repr(e)}
    if rank == 0 and isinstance(res.get("e2e"), dict) and "error" not in res["e2e"]:
        res["e2e"]["ingest"] = committed_ingest()
    if rank == 0:
        if world == 1 and not emu and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(wl, L)
                res["gpu_over_cpu"] = round(value / res["cpu_baseline"]["value"], 1)
            except Exception as e:     # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"error": repr(e)}
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
