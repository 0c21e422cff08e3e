#!/usr/bin/env python3
"""End-to-end VideoCoF edit on one MI355X: prompt strings + source video in, edited video out.

    python tools/bench_e2e.py [--model 14b|1.3b] [--steps 4] [--frames 81] [--height 480] [--width 832]

Everything the reference's fast_infer.py runs per video, through this package's WanPipeline: umT5-XXL encode of
the prompt, WanVAE encode of the source clip, the CoF denoise loop (source | grounding | target latents,
guidance 1.0 as in fast_infer.py:163), WanVAE decode of the grounding and edit segments.  Random-init weights of
the real architectures, synthetic video, a toy whitespace tokenizer (the tokenizer is host-side and not timed
meaningfully).  Reported: wall seconds per stage (HIP events would hide host gaps; these are synchronised
wall-clock stages) and the total."""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

DIMS = {"14b": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40),
        "1.3b": dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30)}


class ToyTokenizer:
    def __init__(self, vocab):
        self.vocab = vocab

    def __call__(self, prompt, padding=None, max_length=512, truncation=True, add_special_tokens=True, return_tensors="pt"):
        ids = torch.zeros(len(prompt), max_length, dtype=torch.long)
        mask = torch.zeros(len(prompt), max_length, dtype=torch.long)
        for b, p in enumerate(prompt):
            toks = [2 + (sum(map(ord, w)) % (self.vocab - 2)) for w in p.split()][: max_length - 1] + [1]
            ids[b, :len(toks)] = torch.tensor(toks)
            mask[b, :len(toks)] = 1
        return SimpleNamespace(input_ids=ids, attention_mask=mask)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="14b", choices=sorted(DIMS))
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--frames", type=int, default=81)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=832)
    args = ap.parse_args()
    from videocof_amd import (AutoencoderKLWan, FlowUniPCMultistepScheduler, WanPipeline, WanT5EncoderModel,
                              WanTransformer3DModel)
    from videocof_amd.weights import random_dit_state_dict, random_t5_state_dict, random_vae_state_dict
    dev = torch.device("cuda:0")
    d = DIMS[args.model]
    dit = WanTransformer3DModel(**d)
    dit.load_state_dict(random_dit_state_dict(dev, seed=0, dim=d["dim"], ffn_dim=d["ffn_dim"], num_layers=d["num_layers"]), device=dev)
    vae = AutoencoderKLWan()
    vae.load_state_dict(random_vae_state_dict(dev), device=dev)
    tcfg = dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32)
    t5 = WanT5EncoderModel(shared_pos=False, **tcfg)
    t5.load_state_dict(random_t5_state_dict(dev, **tcfg), device=dev)
    pipe = WanPipeline(tokenizer=ToyTokenizer(tcfg["vocab"]), text_encoder=t5, vae=vae, transformer=dit,
                       scheduler=FlowUniPCMultistepScheduler(shift=1))
    g = torch.Generator(device=dev).manual_seed(0)
    video = (torch.rand(1, 3, args.frames, args.height, args.width, device=dev, generator=g) * 2 - 1).bfloat16()
    prompt = "remove the red cup from the wooden table and keep everything else unchanged"
    kw = dict(video=video, prompt=prompt, height=args.height, width=args.width, source_frames=args.frames,
              reasoning_frames=4, num_inference_steps=args.steps, guidance_scale=1.0, shift=3, repeat_rope=True, cot=True,
              generator=g, output_type="numpy", return_dict=True)

    stages = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        stages[name] = round(time.perf_counter() - t0, 4)
        return out

    pipe(**{**kw, "num_inference_steps": 1})            # warm-up: allocator, workspaces, LDS attributes
    # stage timings with the same calls the pipeline makes
    timed("text_encoder", lambda: pipe.encode_prompt(prompt, None, False, device=dev))
    timed("vae_encode", lambda: vae.encode(video)[0].mode())
    out = timed("pipeline_total", lambda: pipe(**kw))
    lat = out.latents if getattr(out, "latents", None) is not None else None
    tl = (args.frames - 1) // 4 + 1
    z = torch.randn(1, 16, tl + 1, args.height // 8, args.width // 8, device=dev, generator=g).bfloat16()
    timed("vae_decode_ground_plus_edit", lambda: (vae.decode(z[:, :, :1]).sample, vae.decode(z[:, :, 1:]).sample))
    stages["dit_denoise_loop"] = round(stages["pipeline_total"] - stages["text_encoder"] - stages["vae_encode"]
                                       - stages["vae_decode_ground_plus_edit"], 4)
    print(json.dumps({"what": f"VideoCoF edit end to end, Wan2.1-{args.model} dims, {args.frames}f@{args.height}x{args.width}, "
                              f"{args.steps} steps, guidance 1.0", "seconds": stages,
                      "edit_video_shape": list(out.edit_videos.shape), "ground_video_shape": list(out.ground_videos.shape),
                      "note": "dit_denoise_loop = pipeline_total - the separately timed stages; pipeline uses cache_context "
                              "and skip_source_prediction (its defaults)"}))


if __name__ == "__main__":
    main()
