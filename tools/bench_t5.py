#!/usr/bin/env python3
"""umT5-XXL text-encoder timing on MI355X (once per video; reported separately from tokens/s).

    python tools/bench_t5.py [--batch 2] [--length 512] [--iters 5]

Random-init weights of the real architecture (config/wan2.1/wan_civitai.yaml:14-26 of the reference:
vocab 256384, dim 4096, 64 heads x 64, ffn 10240, 24 layers = 5.68 B parameters), synthetic token ids.
batch 2 = prompt + negative prompt, the classifier-free-guidance case of inference.py."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--length", type=int, default=512)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    from videocof_amd import WanT5EncoderModel
    from videocof_amd.weights import random_t5_state_dict
    cfg = dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32)
    dev = torch.device("cuda:0")
    m = WanT5EncoderModel(shared_pos=False, **cfg)
    m.load_state_dict(random_t5_state_dict(dev, **cfg), device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    B, L = args.batch, args.length
    ids = torch.randint(1, cfg["vocab"], (B, L), device=dev, generator=g)
    mask = torch.zeros(B, L, dtype=torch.long, device=dev)
    lens = [max(1, L // 4), max(1, L // 40)][:B] + [L] * max(0, B - 2)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    out = m(ids, mask)[0]
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        out = m(ids, mask)[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    C, A, Fd, H = cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"], cfg["num_heads"]
    flop = cfg["num_layers"] * B * (2 * L * (4 * C * A + 3 * C * Fd) + 4 * L * L * A)
    print(json.dumps({"what": "umT5-XXL encoder forward", "batch": B, "length": L, "valid_tokens": lens,
                      "ms": round(dt * 1e3, 2), "tflops_per_s": round(flop / dt / 1e12, 1),
                      "flop": flop, "params_b": 5.68, "dtype": "bf16 weights/activations, fp32 residual + softmax"}))


if __name__ == "__main__":
    main()
