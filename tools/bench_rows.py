#!/usr/bin/env python3
"""HBM-bound row kernels: achieved GB/s of wan_rmsnorm_silu_cl on the WanVAE activation shapes (bf16 in + bf16 out)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocof_amd import ops  # noqa: E402

DEV = "cuda:0"
for C, T, H, W in [(96, 4, 480, 832), (192, 4, 240, 416), (384, 2, 120, 208), (384, 1, 60, 104)]:
    x = torch.randn(T, H, W, C, device=DEV).bfloat16()
    g = torch.ones(C, device=DEV)
    ops.rmsnorm_silu_cl(x, g, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ops.rmsnorm_silu_cl(x, g, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"rmsnorm_silu_cl C={C:3d} pixels={T*H*W:8d}: {dt*1e3:7.3f} ms  {x.numel()*4/dt/1e9:7.0f} GB/s")
