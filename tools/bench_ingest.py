#!/usr/bin/env python3
"""Weight ingest at real scale: a SYNTHETIC checkpoint directory shaped like Wan2.1-T2V-14B -- sharded
safetensors (bf16, ~28.6 GB), config.json -- plus three rank-128 LoRA files with ComfyUI key names (the three merges of
fast_infer.py:366-386: VideoCoF, FusionX, and an optional third), loaded through ``WanTransformer3DModel.from_pretrained``
(wan_transformer3d.py:1157-1299 rules) and ``lora_utils.merge_lora`` (lora_utils.py:371-500), timed, and CHECKED: sampled rows of
sampled Linears against  W0 + sum_i m_i * alpha_i / r * up_i @ down_i  evaluated in fp64 from the files.

    python tools/bench_ingest.py [--layers 40] [--dir /tmp/...] [--keep] [--rank 128] [--shards 7]

Prints ONE JSON line: write_s (untimed preparation), load_s, merge_s (3 merges), bytes, GB/s, the check.  There is no network on the
box, hence synthetic values; the file layout, key names, dtypes and sizes are the real ones."""
from __future__ import annotations

import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG_14B = dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, in_dim=16, out_dim=16, text_dim=4096, freq_dim=256,
               text_len=512, eps=1e-6, model_type="t2v", patch_size=[1, 2, 2])
LORA_TARGETS = [f"{a}.{l}" for a in ("self_attn", "cross_attn") for l in "qkvo"] + ["ffn.0", "ffn.2"]


def write_checkpoint(path, cfg, shards, dev, seed=0):
    """config.json + `shards` safetensors files holding every parameter of the architecture (bf16; values N(0, 0.02) for matrices,
    small non-zero vectors).  Returns (bytes written, seconds)."""
    from safetensors.torch import save_file
    from videocof_amd.weights import dit_param_shapes
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    shapes = dit_param_shapes(dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], num_layers=cfg["num_layers"], in_dim=cfg["in_dim"],
                              out_dim=cfg["out_dim"], text_dim=cfg["text_dim"], freq_dim=cfg["freq_dim"])
    names = list(shapes)
    per = (len(names) + shards - 1) // shards
    g = torch.Generator(device=dev).manual_seed(seed)
    total, t0 = 0, time.perf_counter()
    for s in range(shards):
        part = {}
        for k in names[s * per:(s + 1) * per]:
            shp = shapes[k]
            if len(shp) >= 2 and "modulation" not in k:
                t = torch.randn(shp, device=dev, generator=g) * 0.02
            elif k.endswith(("norm_q.weight", "norm_k.weight", "norm3.weight")):
                t = 1.0 + 0.1 * torch.randn(shp, device=dev, generator=g)
            else:
                t = 0.02 * torch.randn(shp, device=dev, generator=g)
            part[k] = t.to(torch.bfloat16).cpu()
            total += part[k].numel() * 2
        save_file(part, os.path.join(path, f"diffusion_pytorch_model-{s + 1:05d}-of-{shards:05d}.safetensors"))
        del part
    return total, time.perf_counter() - t0


def write_lora(fpath, cfg, rank, dev, seed, alpha):
    """One LoRA file over every attention / FFN Linear of every block, ComfyUI names (lora_utils.py:379-394 renaming rules)."""
    from safetensors.torch import save_file
    g = torch.Generator(device=dev).manual_seed(seed)
    C, F = cfg["dim"], cfg["ffn_dim"]
    sd, total = {}, 0
    for i in range(cfg["num_layers"]):
        for tgt in LORA_TARGETS:
            out_f, in_f = (F, C) if tgt == "ffn.0" else (C, F) if tgt == "ffn.2" else (C, C)
            base = f"diffusion_model.blocks.{i}.{tgt}"
            sd[base + ".lora_down.weight"] = (torch.randn(rank, in_f, device=dev, generator=g) * 0.02).to(torch.bfloat16).cpu()
            sd[base + ".lora_up.weight"] = (torch.randn(out_f, rank, device=dev, generator=g) * 0.02).to(torch.bfloat16).cpu()
            sd[base + ".alpha"] = torch.tensor(float(alpha))
            total += (rank * in_f + out_f * rank) * 2
    save_file(sd, fpath)
    return total


def check(model, path, lora_files, mults, samples, dev):
    """Sampled rows of sampled Linears of the LOADED + MERGED model against fp64 arithmetic on the file contents."""
    from safetensors import safe_open
    import glob
    weights = model.linear_weights()
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    where = {}
    for f in files:
        with safe_open(f, "pt") as h:
            for k in h.keys():
                where[k] = f
    worst, n = 0.0, 0
    for (blk, tgt) in samples:
        mod = f"blocks.{blk}.{tgt}"
        with safe_open(where[mod + ".weight"], "pt") as h:
            w0 = h.get_tensor(mod + ".weight")
        rows = torch.tensor([0, 1, w0.shape[0] // 2, w0.shape[0] - 1])
        want = w0[rows].to(dev, torch.float64)
        cur = want.clone()
        for lf, m in zip(lora_files, mults):
            with safe_open(lf, "pt") as h:
                base = f"diffusion_model.{mod}"
                up = h.get_tensor(base + ".lora_up.weight").to(dev, torch.float64)
                down = h.get_tensor(base + ".lora_down.weight").to(dev, torch.float64)
                alpha = float(h.get_tensor(base + ".alpha"))
            # the merge keeps the weight in bf16 between merges (one rounding per merge call, as the reference's `weight.data +=`
            # on a bf16 parameter does): model that rounding, in fp64 otherwise
            cur = (cur + m * alpha / up.shape[1] * (up[rows.to(dev)] @ down)).to(torch.bfloat16).to(torch.float64)
        got = weights[mod][rows.to(dev)].to(torch.float64)
        # one bf16 ulp of slack per merge for the fp32-vs-fp64 product landing on the other side of a rounding boundary
        tol = 3 * 2.0 ** -8 * cur.abs().clamp(min=1e-3)
        worst = max(worst, float(((got - cur).abs() / tol).max()))
        n += got.numel()
    return {"sampled_linears": len(samples), "sampled_elements": n, "max_err_over_tolerance": round(worst, 4), "ok": bool(worst <= 1.0),
            "tolerance": "3 bf16 ulps (2^-8 relative) per element: one rounding boundary per merge call"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=40)
    ap.add_argument("--rank", type=int, default=128)
    ap.add_argument("--shards", type=int, default=7)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = dict(CFG_14B, num_layers=args.layers)
    work = args.dir or tempfile.mkdtemp(prefix="wan_ingest_", dir="/tmp")
    free = shutil.disk_usage(os.path.dirname(work.rstrip("/")) or "/tmp").free
    need = int(1.15 * (args.layers / 40.0) * (28.6e9 + 3 * 1.05e9 * args.rank / 128))
    if free < need:
        raise SystemExit(f"{work}: {free / 1e9:.0f} GB free, need {need / 1e9:.0f} GB")
    try:
        from videocof_amd import WanTransformer3DModel
        from videocof_amd.lora_utils import merge_lora
        from types import SimpleNamespace
        ckpt = os.path.join(work, "transformer")
        nbytes, write_s = write_checkpoint(ckpt, cfg, args.shards, dev)
        loras, lbytes = [], 0
        for i, (name, alpha) in enumerate((("videocof.safetensors", args.rank), ("fusionx.safetensors", args.rank / 2), ("third.safetensors", args.rank))):
            f = os.path.join(work, name)
            lbytes += write_lora(f, cfg, args.rank, dev, 100 + i, alpha)
            loras.append(f)
        mults = (1.0, 1.0, 0.5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model = WanTransformer3DModel.from_pretrained(ckpt)
        torch.cuda.synchronize()
        load_s = time.perf_counter() - t0
        pipe = SimpleNamespace(transformer=model)
        t0 = time.perf_counter()
        merged = []
        for f, m in zip(loras, mults):
            merge_lora(pipe, f, m, device=dev)
            merged.append(model.lora_layers_merged)
        torch.cuda.synchronize()
        merge_s = time.perf_counter() - t0
        L = args.layers
        samples = [(0, "self_attn.q"), (0, "self_attn.k"), (L // 2, "self_attn.v"), (L - 1, "cross_attn.o"), (L // 3, "ffn.0"), (L - 1, "ffn.2")]
        chk = check(model, ckpt, loras, mults, samples, dev)
        res = {"what": f"synthetic Wan2.1-T2V-14B-shaped checkpoint ({args.layers} layers, {args.shards} bf16 safetensors shards) through "
                       f"from_pretrained + three rank-{args.rank} LoRA files through merge_lora",
               "checkpoint_bytes": nbytes, "lora_bytes": lbytes, "write_s_untimed_preparation": round(write_s, 2),
               "load_s": round(load_s, 2), "load_GBps": round(nbytes / load_s / 1e9, 2),
               "merge_s": round(merge_s, 2), "lora_layers_merged": merged, "check": chk,
               "page_cache": "files were written by this process just before: reads come from the page cache, not from disk"}
        print(json.dumps(res), flush=True)
        return 0 if chk["ok"] else 1
    finally:
        if not args.keep and args.dir is None:
            shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
