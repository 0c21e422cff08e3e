#!/usr/bin/env python3
"""Developer aid: the persistent GEMM's plan knobs (rasterisation group, smallest stream-K range, ticket order) swept at the shapes
where it trails the library (the 8-way Ulysses shard's M = 8 392; the 1.3B model's K = 1 536), one process, arms interleaved.
usage: python tools/bench_gemm_knobs.py"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from videocof_amd import ops  # noqa: E402

dev = "cuda:0"
shapes = [("SP8 o/q", 8392, 5120, 5120, ops.EPI_BF16), ("SP8 o+resid", 8392, 5120, 5120, ops.EPI_RESID_F32), ("SP8 ffn.2+resid", 8392, 5120, 13824, ops.EPI_RESID_F32),
          ("SP8 q|k", 8392, 10240, 5120, ops.EPI_BF16),
          ("1.3B q|k", 67080, 3072, 1536, ops.EPI_BF16), ("1.3B o+resid", 67080, 1536, 1536, ops.EPI_RESID_F32), ("1.3B ffn.0+gelu", 67080, 8960, 1536, ops.EPI_GELU_BF16),
          ("1.3B ffn.2+resid", 67080, 1536, 8960, ops.EPI_RESID_F32),
          ("14B q|k", 67080, 10240, 5120, ops.EPI_BF16), ("14B o+resid", 67080, 5120, 5120, ops.EPI_RESID_F32), ("14B cross-q", 67080, 5120, 5120, ops.EPI_BF16),
          ("14B ffn.0+gelu", 67080, 13824, 5120, ops.EPI_GELU_BF16), ("14B ffn.2+resid", 67080, 5120, 13824, ops.EPI_RESID_F32),
          ("SP8 ffn.0+gelu", 8392, 13824, 5120, ops.EPI_GELU_BF16), ("SP4 o/q", 16770, 5120, 5120, ops.EPI_BF16), ("SP2 o/q", 33540, 5120, 5120, ops.EPI_BF16)]
if "--split" in sys.argv:
    ARMS_SPLIT = True
arms = [("default", {})] + [(f"gm={v}", {"gemm_gm": v}) for v in (1, 2, 4, 6, 8)] + [(f"min_units={v}", {"gemm_pk_min_units": v}) for v in (2, 5, 20, 40)] + \
       [("lockstep order", {"gemm_pk_order": 1})]
g = torch.Generator(device=dev).manual_seed(0)
base_arms = arms
for name, M, N, K, epi in shapes:
    if "--split" in sys.argv:      # whole leftover tiles (min_units = units per tile: no stream-K cut) and half tiles against the default quarter
        arms = [("default", {}), ("no split", {"gemm_pk_min_units": K // 128}), ("half tiles", {"gemm_pk_min_units": max(1, K // 256)}),
                ("quarter tiles", {"gemm_pk_min_units": max(1, (K // 128 + 3) // 4)})]
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    bias = torch.zeros(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == ops.EPI_RESID_F32 else torch.bfloat16)
    kw = dict(gate=torch.zeros(1, N, device=dev), rows_per_batch=M) if epi == ops.EPI_RESID_F32 else {}
    ref = torch.mm(a, w.t())
    best = {}
    for rnd in range(3):
        for label, knobs in arms + [("torch.mm (hipBLASLt)", None)]:
            if knobs is None:
                fn = lambda: torch.mm(a, w.t(), out=ref)
            else:
                for k, v in knobs.items():
                    ops.set_tuning(k, v)
                fn = lambda: ops.gemm(a, w, bias, epi, out=out, **kw)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            best[label] = min(best.get(label, 1e9), (time.perf_counter() - t0) / 10 * 1e3)
            if knobs:
                for k in knobs:
                    ops.set_tuning(k, 0)
    base = best["default"]
    print(f"{name:16s} M={M} N={N} K={K}: " + "  ".join(f"{l} {ms:.3f}" + (f" ({base / ms:.3f}x)" if l != "default" else f" ms {2.0 * M * N * K / ms / 1e9:.0f} TF") for l, ms in best.items()), flush=True)
