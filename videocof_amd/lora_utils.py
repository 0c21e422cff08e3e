"""LoRA merge for reference-format checkpoints (the step before the hot path).

The reference merges LoRAs into the dense ``nn.Linear`` weights before inference
(``merge_lora``, ``videox_fun/utils/lora_utils.py:371-500``; three merges in
``fast_infer.py:366-386``): ``W += multiplier * (alpha / r) * up @ down``.  The kernels of this
package only ever see dense weights, so the merge happens on the *state dict* (reference key names)
before ``WanTransformer3DModel.load_state_dict`` packs it:

    sd = load_file("diffusion_pytorch_model.safetensors")
    merge_lora_state_dict(sd, load_file("videocof.safetensors"), 1.0, device="cuda")
    model.load_state_dict(sd)

Key conventions (the renaming rules of lora_utils.py:379-394, restated as they stand in ``_groups`` -- including which entries they DROP):
  * ComfyUI/Wan:  ``diffusion_model.blocks.0.self_attn.q.lora_down.weight`` (+ ``lora_up``, ``alpha``)
  * PEFT:         ``blocks.0.self_attn.q.lora_A.default.weight`` / ``lora_B.default.weight``
  * kohya:        ``lora_unet__blocks_0_self_attn_q.lora_down.weight``
Text-encoder entries (``lora_te``) and entries without both matrices are skipped, as the reference does
(:408-411, :472-476).  The matmul runs in fp32 on ``device`` (a one-off weight preparation, not a
per-token op) and the result is stored back in the weight's dtype.
"""
from __future__ import annotations

import warnings
from collections import defaultdict
from typing import Dict, Optional

import torch

__all__ = ["merge_lora_state_dict", "unmerge_lora_state_dict", "merge_lora", "unmerge_lora"]


def _module_index(sd: Dict[str, torch.Tensor]) -> Dict[str, str]:
    """underscore-flattened module path -> dotted module path, for every 2-D+ '.weight' entry."""
    idx = {}
    for k, v in sd.items():
        if k.endswith(".weight") and v.dim() >= 2:
            mod = k[:-len(".weight")]
            idx[mod.replace(".", "_")] = mod
    return idx


def _groups(lora_sd: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
    """LoRA tensors grouped the way the reference groups them (lora_utils.py:378-395): flattened module name -> {element: tensor}.

    The renaming rules are restated as they stand, because WHICH entries of a file take part follows from them:
      * a name containing ``diffusion_model``: ``diffusion_model.`` -> ``lora_unet__``, ``blocks.`` -> ``blocks_``, and the dots around
        ``self_attn`` / ``cross_attn`` / ``ffn`` become underscores -- nothing else, so ``diffusion_model.head.head.lora_up.weight`` or
        ``diffusion_model.text_embedding.0.lora_down.weight`` keep a dot inside the module path;
      * a name containing ``lora_A`` / ``lora_B`` (PEFT): the same, plus ``.lora_A.default.`` -> ``.lora_down.`` (only with ``.default.``);
      * the name is then cut AT ITS FIRST DOT into (layer, element).
    A pair is merged only if its layer holds elements called exactly ``lora_up.weight`` and ``lora_down.weight`` (:472-476) -- so the
    ``head`` / ``text_embedding`` / ``time_embedding`` / ``time_projection`` entries of a ``diffusion_model.``-style file (layer = the part
    before the remaining dot, element = ``head.lora_up.weight`` ...) are DROPPED by the reference, as are ``lora_A`` names without
    ``.default.``; the kohya spelling of the same modules (``lora_unet__head_head.lora_up.weight``) is merged.  Mirrored, not repaired: a
    checkpoint must give the weights here that it gives there.  ``lora_te`` layers (text encoder) are not this model's."""
    groups: Dict[str, Dict[str, torch.Tensor]] = defaultdict(dict)
    for key, val in lora_sd.items():
        k = key
        if "diffusion_model" in k:
            k = (k.replace("diffusion_model.", "lora_unet__").replace("blocks.", "blocks_").replace(".self_attn.", "_self_attn_")
                 .replace(".cross_attn.", "_cross_attn_").replace(".ffn.", "_ffn_"))
        if "lora_A" in k or "lora_B" in k:
            k = ("lora_unet__" + k).replace("blocks.", "blocks_").replace(".self_attn.", "_self_attn_") \
                .replace(".cross_attn.", "_cross_attn_").replace(".ffn.", "_ffn_") \
                .replace(".lora_A.default.", ".lora_down.").replace(".lora_B.default.", ".lora_up.")
        if "." not in k:
            raise ValueError(f"LoRA tensor name without an element part: {key!r}")          # (:394 fails to unpack the split)
        layer, elem = k.split(".", 1)
        if "lora_te" in layer:
            continue
        # the module the reference's attribute walk ends at (:417-468) is the one whose dotted path flattens to this name
        groups[layer.split("lora_unet_")[-1].lstrip("_")][elem] = val
    return groups


@torch.no_grad()
def merge_lora_state_dict(sd: Dict[str, torch.Tensor], lora_sd: Dict[str, torch.Tensor], multiplier: float = 1.0,
                          device: Optional[str] = None) -> int:
    """In-place ``sd[w] += multiplier * alpha/r * up @ down`` for every LoRA pair that resolves to a
    weight of ``sd``.  Returns the number of merged layers."""
    index = _module_index(sd)
    groups = _groups(lora_sd)
    merged = 0
    for flat, elems in groups.items():
        mod = index.get(flat)
        if mod is None or "lora_up.weight" not in elems or "lora_down.weight" not in elems:
            continue
        w = sd[mod + ".weight"]
        dev = torch.device(device) if device is not None else w.device
        up = elems["lora_up.weight"].to(dev, torch.float32)
        down = elems["lora_down.weight"].to(dev, torch.float32)
        scale = float(elems["alpha"]) / up.shape[1] if "alpha" in elems else 1.0            # :479-482
        if up.dim() == 4:
            delta = torch.mm(up.flatten(1), down.flatten(1)).reshape(w.shape)
        else:
            delta = torch.mm(up, down)      # (rocBLAS / hipBLASLt through torch: checkpoint ingest, once per load, outside every timed path)
        if delta.shape != w.shape:
            raise ValueError(f"LoRA for {mod}: delta {tuple(delta.shape)} does not match weight {tuple(w.shape)}")
        sd[mod + ".weight"] = (w.to(dev, torch.float32) + multiplier * scale * delta).to(w.dtype).to(w.device)
        merged += 1
    return merged


def unmerge_lora_state_dict(sd, lora_sd, multiplier: float = 1.0, device: Optional[str] = None) -> int:
    """Inverse of merge_lora_state_dict (lora_utils.py:503-620)."""
    return merge_lora_state_dict(sd, lora_sd, -multiplier, device)


@torch.no_grad()
def merge_lora(pipeline, lora_path, multiplier, device=None, dtype=torch.float32, state_dict=None,
               transformer_only=False, sub_transformer_name="transformer"):
    """Same call as the reference's ``merge_lora`` (lora_utils.py:371-500; three calls in fast_infer.py:366-386),
    applied IN PLACE to the packed device weights of a loaded ``videocof_amd.WanTransformer3DModel``:
    ``W = bf16(fp32(W) + multiplier * alpha/r * up @ down)``.  Text-encoder entries are ignored (the reference
    skips them with ``transformer_only`` and VideoCoF's LoRAs carry none); returns the pipeline."""
    model = getattr(pipeline, sub_transformer_name)
    weights = model.linear_weights()
    if state_dict is None:
        # the file goes straight to the device that holds the weights: ONE transfer (a rank-128 LoRA over every attention / FFN Linear of
        # the 14B model is 1.2 GB), instead of 800 small pageable ones issued module by module below
        from safetensors.torch import load_file
        wdev = next(iter(weights.values())).device
        state_dict = load_file(lora_path, device=str(wdev))
    index = {name.replace(".", "_"): name for name in weights}
    groups = _groups(state_dict)
    merged, unresolved = 0, []
    for flat, elems in groups.items():
        if "lora_up.weight" not in elems or "lora_down.weight" not in elems:
            continue                                   # norm-only entries etc., skipped like :476-479
        mod = index.get(flat)
        if mod is None:
            unresolved.append(flat)
            continue
        w = weights[mod]
        up = elems["lora_up.weight"].to(w.device, torch.float32).flatten(1)
        down = elems["lora_down.weight"].to(w.device, torch.float32).flatten(1)
        scale = float(elems["alpha"]) / up.shape[1] if "alpha" in elems else 1.0
        delta = torch.mm(up, down)      # (rocBLAS / hipBLASLt through torch: checkpoint ingest, once per load, outside every timed path)
        if delta.shape != w.shape:
            raise ValueError(f"LoRA for {mod}: delta {tuple(delta.shape)} does not match weight {tuple(w.shape)}")
        # One bf16 rounding of the weight per merge call, as in the reference: its `weight.data += ...` runs on the bf16
        # parameter (`dtype=weight_dtype`, fast_infer.py:366-386; lora_utils.py:466-496), where up, down and their
        # product are bf16-rounded too -- here the delta stays fp32 until the single final rounding.  The time-MLP
        # weights are stored fp32 (bf16-rounded values, see load_state_dict) and are re-rounded the same way.
        w.copy_((w.float() + multiplier * scale * delta).to(torch.bfloat16).to(w.dtype))
        merged += 1
    if unresolved:
        warnings.warn(f"merge_lora: {len(unresolved)} LoRA module(s) match no weight of the transformer and were "
                      f"skipped (first: {unresolved[:3]}); {merged} merged", stacklevel=2)
    model.lora_layers_merged = merged                 # the reference returns the pipeline, so the count rides here
    if hasattr(model, "_ctx_cache"):
        model._ctx_cache = None          # hoisted text K/V were built from the old cross-attention weights
    if hasattr(model, "_reset_attention_scratch"):
        model._reset_attention_scratch() # the sticky "max-free attempt off" word described the pre-merge q / k statistics
    if getattr(model, "_fp8", ()):
        # the reference quantises first and merges afterwards (fast_infer.py:352-359, 371-385): the e4m3 copies made by
        # enable_fp8_linear must follow the merged bf16 weights, or q|k / v / ffn keep running the pre-LoRA values
        model.enable_fp8_linear(model._fp8, attn_smooth_k=model.fp8_attn_smooth_k)
    return pipeline


def unmerge_lora(pipeline, lora_path, multiplier=1, device=None, dtype=torch.float32, sub_transformer_name="transformer",
                 state_dict=None):
    """Inverse of ``merge_lora`` (lora_utils.py:503-620); exact up to the bf16 rounding of the merged weights."""
    return merge_lora(pipeline, lora_path, -multiplier, device, dtype, state_dict, sub_transformer_name=sub_transformer_name)
