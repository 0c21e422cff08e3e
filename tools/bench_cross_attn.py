#!/usr/bin/env python3
"""Cross-attention launch (WanT2VCrossAttention, wan_transformer3d.py:308-336: 512 text rows) at the bench shapes, in one process:
the persistent form of the 4-wave kernel (tuning key attn_persist = 1, the product default) against one workgroup per query block
(attn_persist = 0), alternating, best of 5 rounds of 20 launches each.  Prints ms per launch and TFLOP/s; checks the two agree bitwise."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocof_amd import ops

dev = torch.device("cuda", 0)
for (name, L, H) in (("14B CoF 480p", 67080, 40), ("14B CoF 33f", 29640, 40), ("1.3B CoF 480p", 67080, 12), ("14B SP8 shard", 8392, 40), ("14B CoF 720p", 154800, 40)):
    C = H * 128
    g = torch.Generator(device=dev).manual_seed(0)
    q = (torch.randn(1, L, C, device=dev, generator=g) * 0.3).bfloat16()
    k = torch.randn(1, 512, C, device=dev, generator=g).bfloat16()
    vt = torch.randn(1, C, 512, device=dev, generator=g).bfloat16()
    out = torch.empty_like(q)
    ws = ops.AttentionWorkspace()
    res, outs = {}, {}
    for rnd in range(5):
        for arm in (1, 0):
            ops.set_tuning("attn_persist", arm)
            for _ in range(3):
                ops.attention_fwd(q, k, vt, H, out=out, q_prescaled=True, workspace=ws)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                ops.attention_fwd(q, k, vt, H, out=out, q_prescaled=True, workspace=ws)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 20
            if rnd > 0:
                res[arm] = min(res.get(arm, 1e9), ms)
            outs[arm] = out.clone()
    ops.set_tuning("attn_persist", 1)
    fl = 4.0 * L * 512 * C
    print(f"{name:15s} L={L:6d} H={H:2d}: one workgroup per block {res[0]:.3f} ms ({fl / res[0] / 1e9:.0f} TF/s) | persistent {res[1]:.3f} ms "
          f"({fl / res[1] / 1e9:.0f} TF/s) | ratio {res[0] / res[1]:.3f} | bitwise equal {bool(torch.equal(outs[0], outs[1]))}", flush=True)
