#!/usr/bin/env python3
"""Developer aid: why is ffn.0 + GELU 5 % slower inside a step (7.06 ms) than alone (6.6-6.7 ms)?  The same launch timed with HIP events
(a) back to back, (b) behind a 2 GB fill (caches flushed, chip cool: an HBM-bound kernel), (c) behind the LN-modulate launch that produces its
input in the model, (d) behind a self-attention launch of the headline shape (the chip at its power-limited clock), (e) behind both.
usage: python tools/bench_gemm_insitu.py"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from videocof_amd import ops  # noqa: E402

dev = "cuda:0"
L, C, F, H = 67080, 5120, 13824, 40
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(L, C, device=dev, generator=g)
sc, sh = torch.randn(1, C, device=dev, generator=g) * 0.1, torch.randn(1, C, device=dev, generator=g) * 0.1
a = ops.ln_modulate(x, sc, sh, True, L, 1e-6)
w = (torch.randn(F, C, device=dev, generator=g) * 0.02).bfloat16()
bias = torch.zeros(F, device=dev)
out = torch.empty(L, F, device=dev, dtype=torch.bfloat16)
big = torch.empty(2 << 30, device=dev, dtype=torch.uint8)
q = torch.randn(1, L, C, device=dev, generator=g).bfloat16()
k = torch.randn(1, L, C, device=dev, generator=g).bfloat16()
vt = torch.randn(1, C, ops.round_up(L, 64), device=dev, generator=g).bfloat16()
o_attn = torch.empty_like(q)


def gemm():
    ops.gemm(a, w, bias, ops.EPI_GELU_BF16, out=out)


def attn():
    ops.attention_fwd(q, k, vt, H, k_len=L, out=o_attn)


arms = [("back to back", lambda: None), ("behind a 2 GB fill", lambda: big.fill_(1)),
        ("behind its LN-modulate", lambda: ops.ln_modulate(x, sc, sh, True, L, 1e-6, out=a)),
        ("behind a self-attention launch", attn), ("behind attention + 2 GB fill", lambda: (attn(), big.fill_(1)))]
for _ in range(3):
    gemm()
torch.cuda.synchronize()
for rnd in range(2):
    for label, pre in arms:
        ts = []
        for _ in range(6):
            pre()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gemm()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        print(f"ffn.0 + GELU {label:34s}: median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f}  max {ts[-1]:.3f}", flush=True)
