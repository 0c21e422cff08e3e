#!/usr/bin/env python3
"""Per-shape timing of wan_conv_cl on the causal 3x3x3 convolutions of the WanVAE decoder / encoder at 480x832
(SURVEY.md a18): TFLOP/s of each (Cin, Cout, T, H, W) against the 2.5 PFLOP/s bf16 MFMA peak."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocof_amd import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [  # Cin, Cout, T, H, W, what
    (384, 384, 1, 60, 104, "dec mid / stage0, 1 latent frame"),
    (384, 384, 2, 120, 208, "dec stage1 after upsample3d"),
    (192, 192, 4, 240, 416, "dec stage2"),
    (96, 96, 4, 480, 832, "dec stage3 (full res)"),
    (96, 96, 4, 480, 832, "enc stage0 (full res)"),
    (192, 192, 4, 240, 416, "enc stage1"),
    (384, 384, 2, 120, 208, "enc stage2"),
    (96, 3, 4, 480, 832, "dec head conv (Cout 3 -> padded)"),
]


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    print("tuning:", {k: ops.get_tuning(k) for k in ("conv_xcd", "conv_fast", "conv_patch", "conv_mfma")})
    ab = "--ab-mfma" in sys.argv          # alternate the LDS-patch kernel's two matrix instructions (conv_mfma = 32 | 16) per shape
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    for cin, cout, T, H, W, what in SHAPES:
        if only and not any(o in what for o in only):
            continue
        co = max(cout, 8)
        K = 27 * cin
        Kp = ops.round_up(K, 64)
        x = torch.randn(T, H, W, cin, device=DEV, generator=g).bfloat16()
        hist = torch.randn(2, H, W, cin, device=DEV, generator=g).bfloat16()
        w = torch.zeros(co, Kp, device=DEV, dtype=torch.bfloat16)
        w[:, :K] = (torch.randn(co, K, device=DEV, generator=g) * 0.02).bfloat16()
        b = torch.zeros(co, device=DEV)
        f = lambda: ops.conv_cl(x, w, b, co, (3, 3, 3), pad=(2, 1, 1), out_thw=(T, H, W), hist=hist)
        flop = 2.0 * T * H * W * cout * K

        def timed(n=20):
            f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                f()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n
        if ab and cout % 96 == 0:
            best = {32: 1e9, 16: 1e9}
            for rnd in range(4):
                for mi in (32, 16):
                    ops.set_tuning("conv_mfma", mi)
                    dt = timed()
                    if rnd:
                        best[mi] = min(best[mi], dt)
            ops.set_tuning("conv_mfma", 0)
            print(f"{what:38s} Cin={cin:3d} Cout={cout:3d} pixels={T*H*W:8d}: 32x32x16 {best[32]*1e3:7.3f} ms {flop/best[32]/1e12:7.1f} TF/s | "
                  f"16x16x32 {best[16]*1e3:7.3f} ms {flop/best[16]/1e12:7.1f} TF/s | ratio {best[32]/best[16]:.3f}", flush=True)
            continue
        dt = timed()
        print(f"{what:38s} Cin={cin:3d} Cout={cout:3d} pixels={T*H*W:8d}: {dt*1e3:7.3f} ms  {flop/dt/1e12:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
