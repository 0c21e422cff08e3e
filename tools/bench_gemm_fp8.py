#!/usr/bin/env python3
"""Time the e4m3 Linear on the 14B per-layer shapes (developer aid): the persistent stream-K kernel's e4m3 instantiation
(wan_gemm_fp8_ws, gemm_pk = 1, the default) against the 8-wave per-tile kernel (wan_gemm_fp8, gemm_pk = 0), arms alternating in one
process, and the bf16 Linear of the same shape beside them.  usage: python tools/bench_gemm_fp8.py"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from videocof_amd import ops  # noqa: E402

dev = "cuda:0"
L = 8392 if "--sp8" in sys.argv else 67080          # --sp8: the token count of an 8-way Ulysses shard (the plan may keep those per-tile: --force-pk)
PK_ON = 2 if "--force-pk" in sys.argv else 1
shapes = [("q|k", L, 10240, 5120, ops.EPI_BF16), ("o / cross-q", L, 5120, 5120, ops.EPI_BF16), ("ffn.0+gelu", L, 13824, 5120, ops.EPI_GELU_BF16),
          ("ffn.2+resid", L, 5120, 13824, ops.EPI_RESID_F32), ("v (T)", L, 5120, 5120, ops.EPI_BF16_T)]
g = torch.Generator(device=dev).manual_seed(0)
for name, M, N, K, epi in shapes:
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    aq, asc = ops.quantize_rows_fp8(a)
    wq, wsc = ops.quantize_weight_fp8(w)
    bias = torch.zeros(N, device=dev)
    if epi == ops.EPI_BF16_T:
        out = torch.zeros(N, ops.round_up(M, 64), device=dev, dtype=torch.bfloat16)
    elif epi == ops.EPI_RESID_F32:
        out = torch.zeros(M, N, device=dev, dtype=torch.float32)
    else:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    gate = torch.zeros(1, N, device=dev) if epi == ops.EPI_RESID_F32 else None
    kw = dict(gate=gate, rows_per_batch=M) if gate is not None else {}
    obf = out if out.dtype == torch.float32 or epi == ops.EPI_BF16_T else torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    arms = [("e4m3 per-tile (8 waves)", 0, lambda: ops.gemm_fp8(aq, asc, wq, wsc, bias, epi, out=out, **kw)),
            ("e4m3 persistent       ", PK_ON, lambda: ops.gemm_fp8(aq, asc, wq, wsc, bias, epi, out=out, **kw)),
            ("bf16 persistent       ", 1, lambda: ops.gemm(a, w, bias, epi, out=obf, **kw))]
    best = {}
    for rnd in range(3):
        for label, pk, fn in arms:
            ops.set_tuning("gemm_pk", pk)
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 6 * 1e3
            best[label] = min(best.get(label, 1e9), ms)
        ops.set_tuning("gemm_pk", 1)
    for label, _, _ in arms:
        ms = best[label]
        print(f"{name:12s} M={M} N={N} K={K}  {label}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s", flush=True)
    print(f"{name:12s} persistent / per-tile = {best[arms[0][0]] / best[arms[1][0]]:.3f}x, e4m3 persistent / bf16 = {best[arms[2][0]] / best[arms[1][0]]:.3f}x", flush=True)
