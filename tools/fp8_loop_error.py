#!/usr/bin/env python3
"""The lossy e4m3 modes at the end of a whole denoise loop (round-5 review item 6): the 4-step UniPC loop of the headline configuration
(14B dims, VideoCoF layout 81f@480p, L = 67 080, synthetic weights / inputs) run in bf16 and in each fp8 mode FROM THE SAME NOISE, and the
final latents compared -- rel-L2 and cosine over the frames the loop moves (the source frames never move).  With random-init weights the
velocity field means nothing, but a step's error is fed back through the sampler three times exactly as in a real edit.
usage: python tools/fp8_loop_error.py [--layers N] [--steps 4]      prints one JSON line"""
import argparse
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from videocof_amd import FlowUniPCMultistepScheduler, WanTransformer3DModel  # noqa: E402
from videocof_amd.weights import random_dit_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=40)
ap.add_argument("--steps", type=int, default=4)
args = ap.parse_args()
dev = torch.device("cuda", 0)
dim, ffn, heads = 5120, 13824, 40
Fs, G, Ft, h, w = 21, 1, 21, 60, 104
L = (Fs + G + Ft) * (h // 2) * (w // 2)
modes = [("bf16", None), ("bf16 again", None), ("e4m3 Linears (qkv,ffn,o,cross)", ("qkv", "ffn", "o", "cross")),
         ("e4m3 attention products (attn,attn_pv)", ("attn", "attn_pv")), ("everything", ("qkv", "ffn", "o", "cross", "attn", "attn_pv"))]
results, ref = {}, None
for label, layers in modes:
    torch.manual_seed(0)
    model = WanTransformer3DModel(dim=dim, ffn_dim=ffn, num_heads=heads, num_layers=args.layers)
    sd = random_dit_state_dict(dev, seed=0, dim=dim, ffn_dim=ffn, num_layers=args.layers)
    model.load_state_dict(sd, device=dev)
    del sd
    if layers is not None:
        model.enable_fp8_linear(layers)
    g = torch.Generator(device=dev).manual_seed(0)
    lat = torch.randn(1, 16, Fs + G + Ft, h, w, device=dev, generator=g).bfloat16()
    ctx = [torch.randn(37, 4096, device=dev, generator=g).bfloat16()]
    sched = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, solver_order=2)
    sched.set_timesteps(args.steps, device=dev, shift=3)
    per_step = []
    for t in sched.timesteps[:args.steps]:
        v = model(lat, t.expand(1), ctx, L, frame_split_indices=[Fs], ground_frame_indices=[(Fs, Fs + G)])
        v[:, :, :Fs] = 0
        lat = sched.step(v, t, lat, return_dict=False)[0]
        per_step.append(lat[:, :, Fs:].float().clone())
    torch.cuda.synchronize()
    if ref is None:
        ref = per_step
    else:
        rel = [float((a - b).norm() / b.norm()) for a, b in zip(per_step, ref)]
        cos = float(torch.nn.functional.cosine_similarity(per_step[-1].flatten(), ref[-1].flatten(), dim=0))
        results[label] = {"rel_l2_after_each_step": [round(x, 5) for x in rel], "rel_l2_final": round(rel[-1], 5), "cosine_final": round(cos, 6)}
        print(label, results[label], file=sys.stderr, flush=True)
    del model
    torch.cuda.empty_cache()
print(json.dumps({"what": "final latents of the 4-step UniPC CoF loop vs the bf16 loop from the same noise (frames the loop moves)", "workload": f"14B dims, {args.layers} layers, L = {L}",
                  "steps": args.steps, "modes": results}))
