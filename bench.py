#!/usr/bin/env python3
"""Benchmark of the hot path: VideoCoF 4-step denoising on the HIP-backed Wan2.1 DiT.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 14b-cof]

A *step* is one denoise step of the reference loop (pipeline_wan.py:694-740): one DiT forward
over the whole CoF token slab + `noise_pred[:, :, :cc] = 0` + one UniPC update.  Inputs are
synthetic (random-init weights of the named architecture, N(0,1) latents and text embeddings,
SURVEY.md section 8d) and are resident in HBM before the timed region.  `value` is whole-job
latent tokens/s = B * L * K / wall with L = ALL DiT tokens per sample (the quantity the FLOP
formula is written in); `denoised_only_tokens_per_s` scales it by (G+Ft)/(Fs+G+Ft).

N > 1 (launched by `python -m torch.distributed.run --nproc-per-node N ...`): the SAME video is
sequence-sharded over the N GPUs with Ulysses head all-to-all on RCCL (strong scaling), or
`--mode dp` runs N independent replicas (weak scaling; what the reference CLI does,
fast_infer.py:272).

Rank 0 prints ONE JSON line; it also carries
  roofline      -- the dominant kernel (self-attention flash kernel): algorithmic FLOP per launch
                   / mean launch duration measured with HIP events on the launch stream inside
                   the timed region, vs the bf16 dense MFMA peak (2.5 PFLOP/s);
  cpu_baseline  -- the CPU oracle (a port of the reference's math, oracle/wan_oracle.py) timed on
                   the host cores on a bounded sample of the same workload (N=1, rank 0 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

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


def pmc_traffic(workload, shards):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, profiles/r01/
    bench14b_1step_pmc_summary.json; FETCH_SIZE x2 per the gfx950 correction of the microarch guide).
    Counters cannot be read live from inside the process, so this is null for shapes that were not profiled."""
    if workload != "14b-cof" or shards != 1:
        return None, None
    note = "L2->fabric requests; includes Infinity-Cache hits (K/V re-streamed per query block)"
    alg = 4 * 67080 * 5120 * 2
    try:    # passes of the current kernel (tools/profile_bench.sh -> tools/pmc_summary.py)
        path = os.path.join(ROOT, "profiles", "r01", "bench14b_prescaled_pmc_summary.json")
        with open(path) as f:
            d = next(v for k, v in json.load(f).items() if k.startswith("attn_fwd_v2_kernel<0"))
        fetch, write = d["fetch"]["avg_counter"] * 1024 * 2, d["write"]["avg_counter"] * 1024
        return fetch + write, {"fetch_bytes_x2_corrected": fetch, "write_bytes": write, "algorithmic_bytes": alg,
                               "note": note, "source": "profiles/r01/bench14b_prescaled_pmc_summary.json"}
    except Exception:
        pass
    try:    # earlier passes (plain-q form of the same kernel)
        path = os.path.join(ROOT, "profiles", "r01", "bench14b_1step_pmc_summary.json")
        with open(path) as f:
            d = json.load(f)["attn_fwd_kernel"]
        fetch = d["fetch"]["avg_counter_KB"] * 1024 * 2
        write = d["write"]["avg_counter_KB"] * 1024
        return fetch + write, {"fetch_bytes_x2_corrected": fetch, "write_bytes": write, "algorithmic_bytes": alg,
                               "note": note, "source": "profiles/r01/bench14b_1step_pmc_summary.json"}
    except Exception:
        return None, None


def cpu_baseline(wl, budget_s=25.0):
    """Time the CPU oracle on a bounded sample: ONE transformer block of the workload's width on a
    (9,16,16) CoF grid = 2304 tokens, fp32, all host cores; scale by the layer count."""
    from oracle import wan_oracle as O
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    C, ffn, H = wl["dim"], wl["ffn_dim"], wl["num_heads"]
    cfg = O.DiTConfig(dim=C, ffn_dim=ffn, num_heads=H, num_layers=1)
    g = torch.Generator().manual_seed(0)
    sd = {}
    p = "blocks.0."
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
    grid = (9, 16, 16)
    L = 9 * 16 * 16
    x = torch.randn(L, C, generator=g)
    e0 = torch.randn(6, C, generator=g) * 0.1
    ctx = torch.randn(512, C, generator=g)
    ang = O.rope_angles(128)
    with torch.no_grad():
        O.block_forward(x, e0, ctx, sd, 0, cfg, grid, ang, 4, (4, 5), L)          # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            O.block_forward(x, e0, ctx, sd, 0, cfg, grid, ang, 4, (4, 5), L)
            n += 1
            if time.perf_counter() - t0 > budget_s / 2 or n >= 8:
                break
        dt = (time.perf_counter() - t0) / n
    return {"value": L / (dt * wl["num_layers"]), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle/wan_oracle.block_forward, 1 of {wl['num_layers']} blocks (C={C}, {H} heads), "
                      f"L=2304 tokens (9x16x16 CoF grid), fp32, {n} reps, {dt:.2f} s/block; tokens/s = "
                      f"L / (s_per_block * layers).  Attention share at L=2304 is far below the "
                      f"L=67080 workload's, so this flatters the CPU."}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="14b-cof", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="sp", choices=["sp", "dp"], help="N>1: Ulysses sequence parallel or replicas")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo (host-staged exchanges) only to exercise the N>1 code path on a box "
                         "whose ranks share one GPU -- its numbers are meaningless")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (with --backend gloo)")
    ap.add_argument("--no-kernel-events", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
        args.gpus = world
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    from videocof_amd import FlowUniPCMultistepScheduler, WanTransformer3DModel
    from videocof_amd import dist as vdist
    from videocof_amd.weights import random_dit_state_dict

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    wl = WORKLOADS[args.workload]
    sp = world > 1 and args.mode == "sp"
    if sp and wl["num_heads"] % world:
        raise SystemExit(f"{wl['num_heads']} heads cannot be split over {world} GPUs (Ulysses)")

    # ---------------- model + synthetic inputs (resident in HBM before timing)
    torch.manual_seed(0)
    model = WanTransformer3DModel(dim=wl["dim"], ffn_dim=wl["ffn_dim"], num_heads=wl["num_heads"],
                                  num_layers=wl["num_layers"])
    shapes = dict(dim=wl["dim"], ffn_dim=wl["ffn_dim"], num_layers=wl["num_layers"])
    model.load_state_dict(random_dit_state_dict(dev, seed=0, **shapes), device=dev)
    if sp:
        vdist.init_sequence_parallel()
        model.enable_multi_gpus_inference()
    Fs, G, Ft = wl["fs"], wl["g"], wl["ft"]
    Ftot = Fs + G + Ft
    cof = Fs > 0
    g = torch.Generator(device=dev).manual_seed(0 if sp else rank)       # fast_infer.py:390 seeds per rank
    latents = torch.randn(1, 16, Ftot, wl["h"], wl["w"], device=dev, generator=g).bfloat16()
    ctx = [torch.randn(37, 4096, device=dev, generator=g).bfloat16()]
    L = Ftot * (wl["h"] // 2) * (wl["w"] // 2)
    seq_len = L
    fsi = [Fs] if cof else None
    gfi = [(Fs, Fs + G)] if cof else None
    sched = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, solver_order=2)

    prof = [] if not args.no_kernel_events else None
    model._attn_events = prof

    def run(n_steps):
        nonlocal latents
        sched.set_timesteps(max(n_steps, 1), device=dev, shift=3)
        lat = latents
        for t in sched.timesteps[:n_steps]:
            v = model(lat, t.expand(1), ctx, seq_len, frame_split_indices=fsi, ground_frame_indices=gfi)
            if cof:
                v[:, :, :Fs] = 0
            lat = sched.step(v, t, lat, return_dict=False)[0]
        return lat

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup > 0:
        run(args.warmup)
    if prof is not None:
        prof.clear()
    fence()
    t0 = time.perf_counter()
    out = run(args.steps)
    fence()
    wall = time.perf_counter() - t0
    if world > 1:
        tw = torch.tensor([wall], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    assert torch.isfinite(out.float()).all(), "non-finite latents"

    # ---------------- dominant kernel: self-attention launches inside the timed region
    roof = None
    if prof:
        ms = [a.elapsed_time(b) for a, b in prof]
        avg_ms = sum(ms) / len(ms)
        heads_local = wl["num_heads"] // (world if sp else 1)
        Lk = L
        Lq = model._last_attn_rows
        flop = 4.0 * Lq * Lk * heads_local * 128
        ach = flop / (avg_ms * 1e-3) / 1e12
        traffic, traffic_detail = pmc_traffic(args.workload, world if sp else 1)
        roof = {"kernel": "attn_fwd_v2_kernel<0, true> (self-attention; per wan_attention_fwd call = main launch + split-KV tail round + merge)", "bound": "mfma", "achieved": round(ach, 1),
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                "traffic": traffic, "traffic_detail": traffic_detail, "launches": len(ms),
                "avg_ms": round(avg_ms, 3), "flop_per_launch": flop, "dtype_peak": "bf16 dense MFMA"}

    units = world if (world > 1 and not sp) else 1            # dp: every rank denoises its own video
    tokens = units * L * args.steps
    value = tokens / wall
    tot_flop, attn_flop = dit_flops(L, wl["dim"], wl["ffn_dim"], wl["num_layers"])
    res = {
        "metric": "denoised latent tokens/s (4-step 81f@480p Wan2.1 DiT denoise loop, all DiT tokens counted)",
        "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 2), "higher_is_better": True,
        "scaling": "strong" if sp or world == 1 else "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic (random-init weights, N(0,1) latents + text embeddings)",
        "config": {"workload": wl["desc"], "layout": "VideoCoF (src|ground|tgt)" if cof else "T2V",
                   "latent": [1, 16, Ftot, wl["h"], wl["w"]], "grid": [Ftot, wl["h"] // 2, wl["w"] // 2],
                   "tokens_per_sample": L, "global_batch": units, "guidance_scale": 1.0,
                   "parallelism": ("ulysses-sp%d" % world) if sp else ("replicas-dp%d" % world if world > 1 else "single")},
        "tokens_per_s_per_gpu": round(value / world, 1),
        "sec_per_video_4step": round(wall / args.steps * 4, 3),
        "denoised_only_tokens_per_s": round(value * (G + Ft) / Ftot, 1),
        "model_tflops_per_s": round(units * tot_flop * args.steps / wall / 1e12, 1),
        "mfma_frac_whole_step": round(units * tot_flop * args.steps / wall / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
        "roofline": roof,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(wl)
                res["gpu_over_cpu"] = round(value / res["cpu_baseline"]["value"], 1)
            except Exception as e:     # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"error": repr(e)}
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
