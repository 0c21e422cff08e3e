"""Synthetic weights in the reference's checkpoint format.

State dicts use the reference's ``nn.Module`` parameter names
(``blocks.0.self_attn.q.weight`` ...; videox_fun/models/wan_transformer3d.py:662-685),
so the same dict loads into the reference model (golden generation), the CPU
oracle and the HIP-backed ``WanTransformer3DModel`` of this package.

Two fills:
* ``deterministic_dit_state_dict`` -- platform-independent integer-hash fill
  (no torch/numpy RNG involved) used by fixtures and parity tests.  Unlike the
  reference's ``init_weights`` (:1133-1155) it gives **non-zero biases, non-unit
  norm weights and a non-zero head** so every epilogue is exercised
  (parity checklist item 1).
* ``random_dit_state_dict`` -- on-device torch RNG following ``init_weights``
  (xavier-uniform Linears, N(0,.02) embeddings) but with a N(0,.02) head, for
  benchmarks at 1.3B/14B size.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

_MASK = (1 << 64) - 1


def det_uniform(name: str, shape: Iterable[int], scale: float = 1.0, center: float = 0.0) -> torch.Tensor:
    """Uniform(center-scale, center+scale) from a splitmix64 hash of (crc32(name), index)."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    seed = np.uint64(zlib.crc32(name.encode()) * 0x9E3779B97F4A7C15 & _MASK)
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + seed
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))      # [0,1)
    v = (2.0 * u - 1.0) * scale + center
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def dit_param_shapes(dim: int, ffn_dim: int, num_layers: int, in_dim: int = 16, out_dim: int = 16,
                     text_dim: int = 4096, freq_dim: int = 256,
                     patch_size: Tuple[int, int, int] = (1, 2, 2)) -> Dict[str, Tuple[int, ...]]:
    """Every parameter of the T2V WanTransformer3DModel, reference names -> shapes."""
    C = dim
    pp = patch_size[0] * patch_size[1] * patch_size[2]
    s: Dict[str, Tuple[int, ...]] = {
        "patch_embedding.weight": (C, in_dim) + tuple(patch_size), "patch_embedding.bias": (C,),
        "text_embedding.0.weight": (C, text_dim), "text_embedding.0.bias": (C,),
        "text_embedding.2.weight": (C, C), "text_embedding.2.bias": (C,),
        "time_embedding.0.weight": (C, freq_dim), "time_embedding.0.bias": (C,),
        "time_embedding.2.weight": (C, C), "time_embedding.2.bias": (C,),
        "time_projection.1.weight": (6 * C, C), "time_projection.1.bias": (6 * C,),
        "head.head.weight": (pp * out_dim, C), "head.head.bias": (pp * out_dim,),
        "head.modulation": (1, 2, C),
    }
    for i in range(num_layers):
        p = f"blocks.{i}"
        s[p + ".modulation"] = (1, 6, C)
        for attn in ("self_attn", "cross_attn"):
            for lin in ("q", "k", "v", "o"):
                s[f"{p}.{attn}.{lin}.weight"] = (C, C)
                s[f"{p}.{attn}.{lin}.bias"] = (C,)
            s[f"{p}.{attn}.norm_q.weight"] = (C,)
            s[f"{p}.{attn}.norm_k.weight"] = (C,)
        s[p + ".norm3.weight"] = (C,)
        s[p + ".norm3.bias"] = (C,)
        s[p + ".ffn.0.weight"] = (ffn_dim, C)
        s[p + ".ffn.0.bias"] = (ffn_dim,)
        s[p + ".ffn.2.weight"] = (C, ffn_dim)
        s[p + ".ffn.2.bias"] = (C,)
    return s


def deterministic_dit_state_dict(**cfg) -> Dict[str, torch.Tensor]:
    sd = {}
    for name, shape in dit_param_shapes(**cfg).items():
        if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm3.weight"):
            sd[name] = det_uniform(name, shape, 0.25, 1.0)            # non-unit gains
        elif name.endswith(".bias"):
            sd[name] = det_uniform(name, shape, 0.1)                  # non-zero biases
        elif name.endswith("modulation"):
            sd[name] = det_uniform(name, shape, 1.7 / shape[-1] ** 0.5)
        elif name.endswith(".weight"):
            fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
            sd[name] = det_uniform(name, shape, (6.0 / (fan_in + fan_out)) ** 0.5)   # xavier-uniform bound
        else:
            raise KeyError(name)
    return sd


@torch.no_grad()
def random_dit_state_dict(device, dtype=torch.bfloat16, seed: int = 0, exercise_epilogues: bool = False,
                          **cfg) -> Dict[str, torch.Tensor]:
    """``exercise_epilogues``: non-zero biases and non-unit norm gains (the reference's init zeroes / ones them), for
    parity tests at sizes where the integer-hash fill is too slow."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sd = {}
    for name, shape in dit_param_shapes(**cfg).items():
        if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm3.weight"):
            t = torch.ones(shape, device=device, dtype=torch.float32)
            if exercise_epilogues:
                t = t + (torch.rand(shape, device=device, generator=g, dtype=torch.float32) - 0.5) * 0.5
        elif name.endswith(".bias"):
            t = torch.zeros(shape, device=device, dtype=torch.float32)
            if exercise_epilogues:
                t = (torch.rand(shape, device=device, generator=g, dtype=torch.float32) - 0.5) * 0.2
        elif name.endswith("modulation"):
            t = torch.randn(shape, device=device, generator=g, dtype=torch.float32) / shape[-1] ** 0.5
        elif name.startswith(("text_embedding", "time_embedding")) or name == "head.head.weight":
            t = torch.randn(shape, device=device, generator=g, dtype=torch.float32) * 0.02
        else:
            fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
            bound = (6.0 / (fan_in + fan_out)) ** 0.5
            t = (torch.rand(shape, device=device, generator=g, dtype=torch.float32) * 2 - 1) * bound
        sd[name] = t.to(dtype) if t.dim() > 1 and "modulation" not in name else t
    return sd


# ----------------------------------------------------------------------------------------------
# WanVAE (videox_fun/models/wan_vae.py:269-476, 599-617): parameter names exactly as saved by
# AutoencoderKLWan.state_dict() ("model." prefix, :699-702)
# ----------------------------------------------------------------------------------------------
def vae_param_shapes(dim: int = 96, z_dim: int = 16, dim_mult=(1, 2, 4, 4), num_res_blocks: int = 2,
                     temporal_downsample=(False, True, True), prefix: str = "model.") -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}

    def conv3(name, cin, cout, k=(3, 3, 3)):
        s[name + ".weight"] = (cout, cin) + tuple(k)
        s[name + ".bias"] = (cout,)

    def res(name, cin, cout):
        s[name + ".residual.0.gamma"] = (cin, 1, 1, 1)
        conv3(name + ".residual.2", cin, cout)
        s[name + ".residual.3.gamma"] = (cout, 1, 1, 1)
        conv3(name + ".residual.6", cout, cout)
        if cin != cout:
            conv3(name + ".shortcut", cin, cout, (1, 1, 1))

    def attn(name, c):
        s[name + ".norm.gamma"] = (c, 1, 1)
        s[name + ".to_qkv.weight"], s[name + ".to_qkv.bias"] = (3 * c, c, 1, 1), (3 * c,)
        s[name + ".proj.weight"], s[name + ".proj.bias"] = (c, c, 1, 1), (c,)

    mult = list(dim_mult)
    # encoder (:288-320)
    dims = [dim * u for u in [1] + mult]
    conv3("encoder.conv1", 3, dims[0])
    idx, cur = 0, dims[0]
    for i, out_dim in enumerate(dims[1:]):
        for _ in range(num_res_blocks):
            res(f"encoder.downsamples.{idx}", cur, out_dim)
            cur = out_dim
            idx += 1
        if i != len(mult) - 1:
            s[f"encoder.downsamples.{idx}.resample.1.weight"] = (cur, cur, 3, 3)
            s[f"encoder.downsamples.{idx}.resample.1.bias"] = (cur,)
            if temporal_downsample[i]:
                conv3(f"encoder.downsamples.{idx}.time_conv", cur, cur, (3, 1, 1))
            idx += 1
    res("encoder.middle.0", cur, cur)
    attn("encoder.middle.1", cur)
    res("encoder.middle.2", cur, cur)
    s["encoder.head.0.gamma"] = (cur, 1, 1, 1)
    conv3("encoder.head.2", cur, 2 * z_dim)
    conv3("conv1", 2 * z_dim, 2 * z_dim, (1, 1, 1))
    conv3("conv2", z_dim, z_dim, (1, 1, 1))
    # decoder (:392-425)
    tup = list(temporal_downsample)[::-1]
    dims = [dim * u for u in [mult[-1]] + mult[::-1]]
    conv3("decoder.conv1", z_dim, dims[0])
    res("decoder.middle.0", dims[0], dims[0])
    attn("decoder.middle.1", dims[0])
    res("decoder.middle.2", dims[0], dims[0])
    idx = 0
    for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            in_dim = in_dim // 2
        for _ in range(num_res_blocks + 1):
            res(f"decoder.upsamples.{idx}", in_dim, out_dim)
            in_dim = out_dim
            idx += 1
        if i != len(mult) - 1:
            s[f"decoder.upsamples.{idx}.resample.1.weight"] = (out_dim // 2, out_dim, 3, 3)
            s[f"decoder.upsamples.{idx}.resample.1.bias"] = (out_dim // 2,)
            if tup[i]:
                conv3(f"decoder.upsamples.{idx}.time_conv", out_dim, 2 * out_dim, (3, 1, 1))
            idx += 1
    s["decoder.head.0.gamma"] = (dims[-1], 1, 1, 1)
    conv3("decoder.head.2", dims[-1], 3)
    return {prefix + k: v for k, v in s.items()}


def deterministic_vae_state_dict(**cfg) -> Dict[str, torch.Tensor]:
    sd = {}
    for name, shape in vae_param_shapes(**cfg).items():
        if name.endswith("gamma"):
            sd[name] = det_uniform(name, shape, 0.2, 1.0)
        elif name.endswith(".bias"):
            sd[name] = det_uniform(name, shape, 0.05)
        else:
            fan_in = int(np.prod(shape[1:]))
            sd[name] = det_uniform(name, shape, (3.0 / fan_in) ** 0.5)      # unit-gain uniform
    return sd


@torch.no_grad()
def random_vae_state_dict(device, seed: int = 0, **cfg) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sd = {}
    for name, shape in vae_param_shapes(**cfg).items():
        if name.endswith("gamma"):
            sd[name] = torch.ones(shape, device=device)
        elif name.endswith(".bias"):
            sd[name] = torch.zeros(shape, device=device)
        else:
            fan_in = int(np.prod(shape[1:]))
            sd[name] = (torch.rand(shape, device=device, generator=g) * 2 - 1) * (3.0 / fan_in) ** 0.5
    return sd


# ----------------------------------------------------------------------------------------------
# umT5 encoder (videox_fun/models/wan_text_encoder.py:266-279), shared_pos=False: parameter names as in
# WanT5EncoderModel.state_dict()
# ----------------------------------------------------------------------------------------------
def t5_param_shapes(vocab: int, dim: int, dim_attn: int, dim_ffn: int, num_heads: int, num_layers: int,
                    num_buckets: int) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {"token_embedding.weight": (vocab, dim), "norm.weight": (dim,)}
    for i in range(num_layers):
        p = f"blocks.{i}."
        s[p + "norm1.weight"] = (dim,)
        s[p + "attn.q.weight"] = (dim_attn, dim)
        s[p + "attn.k.weight"] = (dim_attn, dim)
        s[p + "attn.v.weight"] = (dim_attn, dim)
        s[p + "attn.o.weight"] = (dim, dim_attn)
        s[p + "norm2.weight"] = (dim,)
        s[p + "ffn.gate.0.weight"] = (dim_ffn, dim)
        s[p + "ffn.fc1.weight"] = (dim_ffn, dim)
        s[p + "ffn.fc2.weight"] = (dim, dim_ffn)
        s[p + "pos_embedding.embedding.weight"] = (num_buckets, num_heads)
    return s


def _t5_scale(name: str, shape, dim: int, dim_attn: int, dim_ffn: int) -> float:
    """Uniform half-width giving the std of init_weights (wan_text_encoder.py:21-35), but with a q scale
    that keeps un-scaled T5 scores O(1) and a position table large enough to matter in the tests."""
    if name.endswith("attn.q.weight"):
        return (3.0 / dim) ** 0.5 * (dim_attn // 1) ** -0.25
    if name.endswith(("attn.k.weight", "attn.v.weight", "ffn.gate.0.weight", "ffn.fc1.weight")):
        return (3.0 / dim) ** 0.5
    if name.endswith("attn.o.weight"):
        return (3.0 / dim_attn) ** 0.5
    if name.endswith("ffn.fc2.weight"):
        return (3.0 / dim_ffn) ** 0.5
    if name.endswith("pos_embedding.embedding.weight"):
        return 1.5
    if name == "token_embedding.weight":
        return 1.7
    raise KeyError(name)


def deterministic_t5_state_dict(**cfg) -> Dict[str, torch.Tensor]:
    sd = {}
    for name, shape in t5_param_shapes(**cfg).items():
        if name.endswith(("norm1.weight", "norm2.weight", "norm.weight")):
            sd[name] = det_uniform("t5." + name, shape, 0.25, 1.0)
        else:
            sd[name] = det_uniform("t5." + name, shape, _t5_scale(name, shape, cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"]))
    return sd


@torch.no_grad()
def random_t5_state_dict(device, dtype=torch.bfloat16, seed: int = 0, **cfg) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sd = {}
    for name, shape in t5_param_shapes(**cfg).items():
        if name.endswith(("norm1.weight", "norm2.weight", "norm.weight")):
            sd[name] = torch.ones(shape, device=device, dtype=dtype)
        else:
            hw = _t5_scale(name, shape, cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"])
            sd[name] = ((torch.rand(shape, device=device, generator=g, dtype=torch.float32) * 2 - 1) * hw).to(dtype)
    return sd
