#!/usr/bin/env python3
"""Time wan_gemm_fp8 on the 14B per-layer shapes (developer aid).  usage: python tools/bench_gemm_fp8.py [gemm_phases ...]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from videocof_amd import ops  # noqa: E402

dev = "cuda:0"
L = 67080
shapes = [("q|k", L, 10240, 5120, ops.EPI_BF16), ("o / cross-q", L, 5120, 5120, ops.EPI_BF16), ("ffn.0+gelu", L, 13824, 5120, ops.EPI_GELU_BF16),
          ("ffn.2+resid", L, 5120, 13824, ops.EPI_RESID_F32), ("v (T)", L, 5120, 5120, ops.EPI_BF16_T)]
phases = [int(a) for a in sys.argv[1:]] or [0]
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
    for rnd in range(2):
        for ph in phases:
            if ph:
                ops.set_tuning("gemm_phases", ph)
            kw = dict(gate=gate, rows_per_batch=M) if gate is not None else {}
            for _ in range(2):
                ops.gemm_fp8(aq, asc, wq, wsc, bias, epi, out=out, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                ops.gemm_fp8(aq, asc, wq, wsc, bias, epi, out=out, **kw)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            print(f"gemm_fp8 {name:12s} M={M} N={N} K={K} phases={ph or 'default'}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s", flush=True)
