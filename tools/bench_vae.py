#!/usr/bin/env python3
"""WanVAE encode/decode timing on MI355X (reported separately from tokens/s).

    python tools/bench_vae.py [--frames 81] [--height 480] [--width 832] [--iters 2]

Random-init weights of the real architecture (126.9 M parameters), synthetic video.  FLOP counts are
the conv + attention FLOPs of the reference VAE (encode 81f@480p 163.7 TF,
decode 21 latent frames 275.4 TF, decode 1 latent frame 4.35 TF)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=81)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--iters", type=int, default=2)
    args = ap.parse_args()
    from videocof_amd import AutoencoderKLWan
    from videocof_amd.weights import random_vae_state_dict
    dev = torch.device("cuda:0")
    vae = AutoencoderKLWan()
    vae.load_state_dict(random_vae_state_dict(dev), device=dev)
    if os.environ.get("WAN_VAE_DECODE_CHUNK"):            # A/B of the decoder's temporal chunking (1 = the reference's per-frame loop)
        vae.decode_chunk = int(os.environ["WAN_VAE_DECODE_CHUNK"])
    g = torch.Generator(device=dev).manual_seed(0)
    video = (torch.rand(1, 3, args.frames, args.height, args.width, device=dev, generator=g) * 2 - 1).bfloat16()
    tl = (args.frames - 1) // 4 + 1
    z = torch.randn(1, 16, tl, args.height // 8, args.width // 8, device=dev, generator=g).bfloat16()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.iters, out

    scale = (args.height * args.width) / (480 * 832)
    t_enc, lat = timed(lambda: vae.encode(video)[0].mode())
    t_dec, vid = timed(lambda: vae.decode(z).sample)
    t_dec1, _ = timed(lambda: vae.decode(z[:, :, :1]).sample)
    assert torch.isfinite(lat.float()).all() and torch.isfinite(vid.float()).all()
    enc_tf = 163.7 * scale * (args.frames / 81)
    dec_tf = 275.4 * scale * (tl / 21)
    print(json.dumps({
        "workload": f"WanVAE {args.frames}f@{args.height}x{args.width} bf16, random-init weights",
        "sec_per_encode": round(t_enc, 3), "encode_tflops_per_s": round(enc_tf / t_enc, 1),
        "sec_per_decode": round(t_dec, 3), "decode_tflops_per_s": round(dec_tf / t_dec, 1),
        "sec_per_decode_1_latent_frame": round(t_dec1, 4),
        "decode_chunk": vae.decode_chunk, "latent_shape": list(lat.shape), "video_shape": list(vid.shape)}))


if __name__ == "__main__":
    main()
