#!/usr/bin/env python
"""TOOLS ONLY -- never imported by the product path.

Head-room yardstick for `wan_gemm_bf16` on the per-layer Linear shapes of the 14B DiT (M = 67 080 tokens and the
8-way Ulysses shard M = 8 392): the same products through `torch.mm` (hipBLASLt / rocBLAS, whatever torch picks) in the
SAME process on the same random operands, alternating arms, HIP events around `reps` back-to-back launches.
Also times our kernel at K, 2K (same M, N) to separate the per-output-tile fixed cost from the K-loop rate:
t(K) = tiles_rounds * (fixed + K/64 * per_ktile).

    python tools/bench_gemm_yardstick.py [--reps 5] [--rounds 3] [--json out.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from videocof_amd import ops  # noqa: E402

SHAPES = [  # (name, M, N, K, epilogue)
    ("q|k proj", 67080, 10240, 5120, "bf16"),
    ("v proj (T)", 67080, 5120, 5120, "bf16_t"),
    ("o proj + gate + resid", 67080, 5120, 5120, "resid"),
    ("cross q", 67080, 5120, 5120, "bf16"),
    ("ffn.0 + gelu", 67080, 13824, 5120, "gelu"),
    ("ffn.2 + gate + resid", 67080, 5120, 13824, "resid"),
    ("SP8 o/q proj", 8392, 5120, 5120, "bf16"),
    ("SP8 q|k proj", 8392, 10240, 5120, "bf16"),
    ("SP8 ffn.0 + gelu", 8392, 13824, 5120, "gelu"),
    ("SP8 ffn.2 + resid", 8392, 5120, 13824, "resid"),
    ("1.3B small ffn.2", 2304, 1536, 8960, "resid"),
    ("1.3B small ffn.0", 2304, 8960, 1536, "gelu"),
    ("1.3B small q|k", 2304, 3072, 1536, "bf16"),
    ("1.3B small o + resid", 2304, 1536, 1536, "resid"),
    # BASELINE configs[1]: the 1.3B-width Linears at the VideoCoF token count (K = 1536 except ffn.2)
    ("1.3B cof q|k", 67080, 3072, 1536, "bf16"),
    ("1.3B cof v (T)", 67080, 1536, 1536, "bf16_t"),
    ("1.3B cof o + gate + resid", 67080, 1536, 1536, "resid"),
    ("1.3B cof cross q", 67080, 1536, 1536, "bf16"),
    ("1.3B cof ffn.0 + gelu", 67080, 8960, 1536, "gelu"),
    ("1.3B cof ffn.2 + gate + resid", 67080, 1536, 8960, "resid"),
]
EPI = {"bf16": ops.EPI_BF16, "gelu": ops.EPI_GELU_BF16, "resid": ops.EPI_RESID_F32, "bf16_t": ops.EPI_BF16_T}


def time_ms(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default="", help="substring filter on the shape names (comma list); skips the K-scaling rows too")
    ap.add_argument("--tuning", default="", help="comma list k=v applied through wan_set_tuning before the run")
    args = ap.parse_args()
    for kv in filter(None, args.tuning.split(",")):
        k, v = kv.split("=")
        ops.set_tuning(k, int(v))
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    rows = []
    only = [t for t in args.only.split(",") if t]
    for name, M, N, K, epi in SHAPES:
        if only and not any(t in name for t in only):
            continue
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = (torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02)
        bias = torch.randn(N, device=dev, dtype=torch.float32)
        gate = torch.randn(1, N, device=dev, dtype=torch.float32)
        x = torch.randn(M, N, device=dev, dtype=torch.float32) if epi == "resid" else None
        out_t = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        wt = w.t()

        def ours():
            if epi == "resid":
                ops.gemm(a, w, bias, EPI[epi], out=x, gate=gate, rows_per_batch=M)
            elif epi == "bf16_t":
                ops.gemm(a, w, bias, EPI[epi], out=vt)
            else:
                ops.gemm(a, w, bias, EPI[epi], out=out_t)

        vt = torch.zeros(N, ops.round_up(M, 64), device=dev, dtype=torch.bfloat16) if epi == "bf16_t" else None

        def lib():  # the bare product (no epilogue): an upper bound on what the library does for this Linear
            torch.mm(a, wt, out=out_t)

        t_ours, t_lib = [], []
        for _ in range(args.rounds):
            t_ours.append(time_ms(ours, args.reps))
            t_lib.append(time_ms(lib, args.reps))
        fl = 2.0 * M * N * K
        row = {"shape": name, "M": M, "N": N, "K": K, "epilogue": epi,
               "wan_ms": min(t_ours), "wan_tflops": fl / min(t_ours) / 1e9, "wan_ms_all": t_ours,
               "torch_mm_ms": min(t_lib), "torch_mm_tflops": fl / min(t_lib) / 1e9, "torch_mm_ms_all": t_lib}
        rows.append(row)
        print(f"{name:24s} M={M:6d} N={N:6d} K={K:6d}  wan {row['wan_ms']:.3f} ms {row['wan_tflops']:.0f} TF"
              f"   torch.mm {row['torch_mm_ms']:.3f} ms {row['torch_mm_tflops']:.0f} TF   ratio {row['torch_mm_ms'] / row['wan_ms']:.3f}",
              flush=True)
        del a, w, x, out_t, vt
        torch.cuda.empty_cache()
    # fixed cost per output tile: same M, N at K and 2K (plain bf16 epilogue)
    for M, N, K in ([] if only else [(67080, 5120, 5120), (67080, 5120, 2560), (8392, 5120, 5120)]):
        ts = []
        for kk in (K, 2 * K):
            a = torch.randn(M, kk, device=dev, dtype=torch.bfloat16)
            w = torch.randn(N, kk, device=dev, dtype=torch.bfloat16) * 0.02
            o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            ts.append(min(time_ms(lambda: ops.gemm(a, w, None, ops.EPI_BF16, out=o), args.reps) for _ in range(args.rounds)))
            del a, w, o
        per_k = (ts[1] - ts[0]) / (K / 64)          # ms per K tile (all rounds of workgroups together)
        fixed = ts[0] - per_k * (K / 64)
        print(f"K-scaling M={M} N={N}: t(K={K}) {ts[0]:.3f} ms, t(2K) {ts[1]:.3f} ms -> fixed {fixed:.3f} ms "
              f"({100 * fixed / ts[0]:.1f} % of t(K)), K-loop rate {2.0 * M * N * 64 / per_k / 1e9:.0f} TF", flush=True)
        rows.append({"kscale": [M, N, K], "t_k": ts[0], "t_2k": ts[1], "fixed_ms": fixed, "loop_tflops": 2.0 * M * N * 64 / per_k / 1e9})
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
