"""MI355X-native Wan2.1 DiT behind the reference's ``WanTransformer3DModel`` surface.

Drop-in for ``videox_fun/models/wan_transformer3d.py:567-1105`` on the T2V /
VideoCoF path: same constructor argument names, same reference-format state-dict
keys, same ``forward(x, t, context, seq_len, ..., frame_split_indices,
ground_frame_indices)`` call and the attributes ``WanPipeline`` touches
(``.config.in_channels``, ``.config.patch_size``, ``.num_inference_steps``,
``.current_steps``, ``.freqs``, ``.dtype``, ``.device``).

Host orchestration is PyTorch-ROCm; all per-token arithmetic runs in the HIP
kernels of ``libwan_hip.so`` (see ``include/wan_hip.h``).  Data layout in HBM for one
forward of B samples with Ll local tokens each (Ll = L when not sequence-parallel):

    x    fp32 [B*Ll, C]     residual stream (fp32 from the first block on)
    h    bf16 [B*Ll, C]     LN/modulate output feeding the GEMMs
    qk   bf16 [B*Ll, 2C]    fused q|k projection, RMSNorm+RoPE applied in place
    vt   bf16 [B, C, ldvt]  V^T written by the V-projection epilogue (attention operand)
    att  bf16 [B*Ll, C]     attention output
    ff   bf16 [B*Ll, ffn]   GELU(ffn.0)

There is no eager fallback: CPU tensors raise.
"""
from __future__ import annotations

import glob
import json
import math
import os
from collections import namedtuple
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import RopeParams
from .dist import get_sp_group

__all__ = ["WanTransformer3DModel", "sinusoidal_embedding_1d", "rope_params"]


def sinusoidal_embedding_1d(dim: int, position: torch.Tensor) -> torch.Tensor:
    """cat(cos, sin) of position * 10000^(-i/half) in fp64 (wan_transformer3d.py:31-41)."""
    if dim % 2:
        raise AssertionError("dim must be even")
    half = dim // 2
    pos = position.to(torch.float64)
    inv = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float64, device=pos.device) / half)
    ang = torch.outer(pos, inv)
    return torch.cat([ang.cos(), ang.sin()], dim=1)


def rope_params(max_seq_len: int, dim: int, theta: float = 10000.0) -> torch.Tensor:
    """complex128 [max_seq_len, dim/2] unit phasors (wan_transformer3d.py:44-52)."""
    inv = 1.0 / torch.pow(torch.tensor(theta, dtype=torch.float64),
                          torch.arange(0, dim, 2, dtype=torch.float64) / dim)
    ang = torch.outer(torch.arange(max_seq_len, dtype=torch.float64), inv)
    return torch.polar(torch.ones_like(ang), ang)


# what nn.Module.load_state_dict returns: unpackable as `m, u = model.load_state_dict(...)` (fast_infer.py:294)
IncompatibleKeys = namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys"])


class StateDictShapeError(ValueError, RuntimeError):
    """A checkpoint tensor whose shape does not fit the configured model.  ValueError for callers that validate inputs,
    RuntimeError because that is what ``nn.Module.load_state_dict`` raises for "size mismatch for <key>"."""


def ulysses_head_padding(num_heads: int, degree: int, head_dim: int):
    """The channel map of Ulysses with num_heads % degree != 0: (H_pad, C_pad, src, valid) with H_pad the next multiple of the
    degree, and for every padded channel cp in WIRE order -- rank r = cp // (Hl * d) holds local slots s = 0 .. Hl - 1, slot s of rank r
    is head s * degree + r (round-robin: the ranks that are one head short are the LAST ones, and every rank's real heads come first)
    -- the real channel it copies (``src``) or ``valid`` = False for a dummy head's channel."""
    Hl = (num_heads + degree - 1) // degree
    Hp, Cp = Hl * degree, Hl * degree * head_dim
    cp = torch.arange(Cp)
    r, sl, j = cp // (Hl * head_dim), (cp % (Hl * head_dim)) // head_dim, cp % head_dim
    hreal = sl * degree + r
    valid = hreal < num_heads
    src = torch.where(valid, hreal * head_dim + j, torch.zeros_like(cp))
    return Hp, Cp, src, valid


class _Block:
    """Packed weights of one WanAttentionBlock (device tensors; bf16 matrices, fp32 vectors)."""
    __slots__ = ("w_qk", "b_qk", "w_v", "b_v", "w_o", "b_o", "nq", "nk",
                 "w_cq", "b_cq", "w_ck", "b_ck", "w_cv", "b_cv", "w_co", "b_co", "ncq", "nck",
                 "n3w", "n3b", "w1", "b1", "w2", "b2", "modulation", "f8", "_cw", "spw")


class WanTransformer3DModel(nn.Module):
    def __init__(self, model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048,
                 ffn_dim=8192, freq_dim=256, text_dim=4096, out_dim=16, num_heads=16, num_layers=32,
                 window_size=(-1, -1), qk_norm=True, cross_attn_norm=True, eps=1e-6, in_channels=16,
                 hidden_size=2048, add_control_adapter=False, in_dim_control_adapter=24,
                 downscale_factor_control_adapter=8, add_ref_conv=False, in_dim_ref_conv=16,
                 cross_attn_type=None):
        super().__init__()
        if model_type != "t2v" or cross_attn_type not in (None, "t2v_cross_attn"):
            raise NotImplementedError("only the t2v cross-attention variant is on the VideoCoF path")
        if add_control_adapter or add_ref_conv:
            raise NotImplementedError("control adapter / ref conv belong to other model families")
        if not (qk_norm and cross_attn_norm) or tuple(window_size) != (-1, -1):
            raise NotImplementedError("Wan2.1 T2V uses qk_norm, cross_attn_norm and global attention")
        if dim % num_heads or dim // num_heads != 128:
            raise NotImplementedError(f"head_dim {dim / num_heads} (the gfx950 attention kernel is built for 128)")
        self.config = SimpleNamespace(
            model_type=model_type, patch_size=tuple(patch_size), text_len=text_len, in_dim=in_dim, dim=dim,
            ffn_dim=ffn_dim, freq_dim=freq_dim, text_dim=text_dim, out_dim=out_dim, num_heads=num_heads,
            num_layers=num_layers, window_size=tuple(window_size), qk_norm=qk_norm,
            cross_attn_norm=cross_attn_norm, eps=eps,
            # as @register_to_config stores them: the constructor's own `in_channels` / `hidden_size` arguments (defaults 16 / 2048,
            # independent of in_dim / dim: wan_transformer3d.py:579-604) and the switches of the other model families
            in_channels=in_channels, hidden_size=hidden_size, add_control_adapter=add_control_adapter,
            in_dim_control_adapter=in_dim_control_adapter, downscale_factor_control_adapter=downscale_factor_control_adapter,
            add_ref_conv=add_ref_conv, in_dim_ref_conv=in_dim_ref_conv, cross_attn_type=cross_attn_type)
        self.model_type, self.patch_size, self.text_len = model_type, tuple(patch_size), text_len
        self.in_dim, self.dim, self.ffn_dim, self.freq_dim = in_dim, dim, ffn_dim, freq_dim
        self.text_dim, self.out_dim, self.num_heads, self.num_layers = text_dim, out_dim, num_heads, num_layers
        self.eps = eps
        self.d = dim // num_heads
        # q leaves the RMSNorm(+RoPE) kernel already multiplied by softmax_scale*log2(e) (fp32, before its
        # bf16 rounding) so the attention kernel needs no per-score multiply-add (wan_hip.h, WAN_ATTN_Q_PRESCALED)
        self._qs = ops.q_prescale(self.d)
        d = self.d
        # complex table kept for API parity (:692-699); kernels use fp32 cos/sin of the same fp64 angles
        self.freqs = torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)),
                                rope_params(1024, 2 * (d // 6))], dim=1)
        self._rope_dev: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self.blocks: List[_Block] = []
        self._w: Dict[str, torch.Tensor] = {}
        self._dtype = torch.bfloat16
        self._device = torch.device("cpu")
        self.teacache = None
        self.should_calc = True
        self.cfg_skip_ratio = None
        self.current_steps = 0
        self.num_inference_steps = None
        self.sp_world_size = 1
        self.sp_world_rank = 0
        self._sp = None
        # Take the Ulysses branch even when the group has ONE rank (tests: the async RCCL exchanges, their waits and the
        # persistent wire buffers become live code on a single GPU; results equal the plain path).  Off in production.
        self.force_ulysses = False
        # Ulysses head-group pipelining: q and o travel as TWO head groups, each in its own all-to-all, and the attention runs once per
        # group -- the q exchange of group 1 runs under the attention of group 0, the o exchange of group 0 under the attention of
        # group 1.  1 = one exchange / one attention launch per layer.  Needs >= 2 local heads.
        self.sp_head_groups = 2
        self._sp_pad = None                 # heads padded to a multiple of the Ulysses degree (see _pad_heads_for_ulysses)
        self._usp = False                   # this forward runs the Ulysses branch (set by forward)
        self._comm_events = None            # bench.py: list collecting (start, end) HIP events around the EXPOSED exchanges
        self.cache_context = False          # hoist step-invariant text K/V (parity neutral)
        self._ctx_cache = None
        self._fp8 = ()                      # enable_fp8_linear: which projections run in e4m3 (lossy, opt-in)
        self.fp8_attn_exponents = (5, 2)    # "attn": q8 = e4m3(q * scale * log2e * 2^5), k8 = e4m3(k * 2^2) (include/wan_hip.h a9')
        # "attn" with K smoothing: replace the two static exponents by per-LAYER ones measured on the first forward after
        # enable_fp8_linear (the largest |q| and |k - mean| of that layer's operands, one bit of head-room): a checkpoint whose q / k
        # gains put the static choice into the e4m3 clamp (+-448) no longer saturates.  What calibration cannot buy is precision:
        # 3 mantissa bits move a log2-domain score by ~4 % of |q||k|/sqrt(d) -- DESIGN.md section 13 states where the mode stops.
        self.fp8_attn_calibrate = True
        self.fp8_attn_smooth_k = True       # "attn": quantise k - mean_tokens(k) (sageattn's smooth_k; softmax-invariant)
        self.use_block_composite = True     # single-device blocks through wan_dit_block_forward (one C call per block)
        self.use_forward_composite = True   # ... and, when nothing hooks into the block loop, the whole token path through wan_dit_forward
        self._cdw = None                    # ctypes wan_dit_weights of the loaded blocks (built on first use)
        # cached activation workspaces by call shape (_workspaces): the eager forward keeps ONE (the last shape; another shape
        # evicts it), a captured hipGraph PINS the set it was recorded with (videocof_amd/graph.py) so that it is never freed
        # under the graph
        self._bufs = {}
        self._bufs_last = None
        # bumped whenever device memory a captured graph may have baked in is replaced (weights reloaded, fp8 copies rebuilt,
        # workspaces released): GraphedForward entries of an older epoch are discarded instead of replayed
        self._graph_epoch = 0
        # Number of leading latent frames whose prediction the caller discards (WanPipeline zeroes
        # noise_pred[:, :, :condition_count], pipeline_wan.py:736).  When set (B = 1, no SP) the LAST block and
        # the head run only on the remaining tokens' query rows -- their keys/values still cover every token --
        # and the discarded frames come back as zeros.  Parity neutral for the pipeline.
        self.skip_source_frames = 0
        # The CoF mask itself on the device: output frames [0, mask_source_frames) are written as zeros by the
        # unpatchify kernel (wan_unpatchify's zero_frames) instead of a separate slice-assign on the result
        # (`noise_pred[:, :, :condition_count] = 0`, pipeline_wan.py:736).  0 = plain forward.
        self.mask_source_frames = 0
        # one attention scratch per call site: the sticky "max-free attempt off" word of one site never reaches another
        self._ws_self, self._ws_cross = ops.AttentionWorkspace(), ops.AttentionWorkspace()
        self._ws_self_sfx, self._ws_cross_sfx = ops.AttentionWorkspace(), ops.AttentionWorkspace()      # _last_block_suffix's launches
        self._probe_layer = None            # bench.py / tests: keep a copy of the residual stream entering this block
        self._probe = None
        self._attn_events = None            # bench.py: list collecting (start, end) HIP events per self-attn launch
        self._last_attn_rows = 0
        self._last_attn_variant = 0

    # ------------------------------------------------------------------ properties
    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device

    # ------------------------------------------------------------------ weights
    def expected_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """Reference state-dict key -> shape for this configuration (the parameters of the modules built in
        wan_transformer3d.py:633-690 / WanAttentionBlock :425-470 / Head :530-545): what ``load_state_dict`` checks every
        tensor against and what ``from_pretrained``'s "Size don't match, skip" rule (:1279-1286) compares with."""
        C, F, T = self.dim, self.ffn_dim, self.text_dim
        pt, ph, pw = self.patch_size
        sh = {"patch_embedding.weight": (C, self.in_dim, pt, ph, pw), "patch_embedding.bias": (C,),
              "text_embedding.0.weight": (C, T), "text_embedding.0.bias": (C,),
              "text_embedding.2.weight": (C, C), "text_embedding.2.bias": (C,),
              "time_embedding.0.weight": (C, self.freq_dim), "time_embedding.0.bias": (C,),
              "time_embedding.2.weight": (C, C), "time_embedding.2.bias": (C,),
              "time_projection.1.weight": (6 * C, C), "time_projection.1.bias": (6 * C,),
              "head.head.weight": (self.out_dim * pt * ph * pw, C), "head.head.bias": (self.out_dim * pt * ph * pw,),
              "head.modulation": (1, 2, C)}
        for i in range(self.num_layers):
            p = f"blocks.{i}."
            sh[p + "modulation"] = (1, 6, C)
            for a in ("self_attn", "cross_attn"):
                for l in ("q", "k", "v", "o"):
                    sh[f"{p}{a}.{l}.weight"], sh[f"{p}{a}.{l}.bias"] = (C, C), (C,)
                sh[f"{p}{a}.norm_q.weight"] = sh[f"{p}{a}.norm_k.weight"] = (C,)
            sh[p + "norm3.weight"] = sh[p + "norm3.bias"] = (C,)
            sh[p + "ffn.0.weight"], sh[p + "ffn.0.bias"] = (F, C), (F,)
            sh[p + "ffn.2.weight"], sh[p + "ffn.2.bias"] = (C, F), (C,)
        return sh

    def _fresh_value(self, key: str, shape: Tuple[int, ...], gen: torch.Generator) -> torch.Tensor:
        """The value a parameter the checkpoint lacks has on a FRESH reference model: its constructor defaults followed by
        ``init_weights`` (wan_transformer3d.py:1133-1155) -- xavier-uniform Linears and patch embedding with zero biases,
        N(0, 0.02) text / time embedding matrices, a zero head matrix, unit norm gains, zero norm3 bias, modulation
        ~ N(0, 1) / sqrt(dim) (:462, :533).  Same distributions; the random draws are this package's own."""
        if key.endswith("modulation"):
            return torch.randn(shape, generator=gen) / self.dim ** 0.5
        if key.endswith(".bias") or key == "head.head.weight":
            return torch.zeros(shape)
        if "norm" in key:
            return torch.ones(shape)
        if key.startswith(("text_embedding", "time_embedding")):
            return torch.randn(shape, generator=gen) * 0.02
        fan_out, fan_in = shape[0], int(math.prod(shape[1:]))
        bound = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, generator=gen) * 2 - 1) * bound

    def load_state_dict(self, state_dict, strict: bool = True, device=None):  # type: ignore[override]
        """Pack a REFERENCE-format state dict (keys as in wan_transformer3d.py's modules) for the
        kernels: bf16 [N,K] matrices (q|k fused), fp32 bias / norm / modulation vectors.

        Every tensor's shape is checked against the configuration (``expected_shapes``) BEFORE anything is packed:
        a mismatch raises ``StateDictShapeError`` (a ValueError and, like nn.Module's "size mismatch for ...", a
        RuntimeError) naming the key -- strict or not, as ``nn.Module.load_state_dict`` does.  ``strict=True``: missing or
        unexpected keys raise KeyError.  ``strict=False``: unexpected keys are reported; missing keys keep their current
        values on a loaded model (fast_infer.py:286-295) and get a fresh model's initial values on an empty one
        (wan_transformer3d.py:1288), and are reported in ``missing_keys`` either way."""
        sd = state_dict
        expect = self.expected_shapes()
        bad = [(k, tuple(v.shape), expect[k]) for k, v in sd.items() if k in expect and tuple(v.shape) != expect[k]]
        if bad:
            k, got, want = bad[0]
            raise StateDictShapeError(f"size mismatch for {k}: checkpoint tensor is {list(got)}, the model "
                                      f"(dim={self.dim}, ffn_dim={self.ffn_dim}, in_dim={self.in_dim}, text_dim={self.text_dim}) "
                                      f"expects {list(want)}" + (f" (+{len(bad) - 1} more)" if len(bad) > 1 else ""))
        missing = [k for k in expect if k not in sd]
        if strict and missing:
            raise KeyError(f"missing key in state_dict: {missing[0]}" + (f" (+{len(missing) - 1} more)" if len(missing) > 1 else ""))
        if strict and any(k not in expect for k in sd):
            extra = [k for k in sd if k not in expect]
            raise KeyError(f"unexpected keys in state_dict: {extra[:5]}{'...' if len(extra) > 5 else ''}")
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if dev.type != "cuda":
            raise RuntimeError("WanTransformer3DModel runs on a HIP device only (no CPU fallback)")
        if missing:
            if getattr(self, "_w", None):
                # nn.Module semantics of strict=False on a loaded model (fast_infer.py:286-295: a fine-tuned checkpoint on
                # top of from_pretrained): keys that are absent keep their current values
                current = self.state_dict()
                sd = {**{k: current[k] for k in missing}, **sd}
            else:
                gen = torch.Generator().manual_seed(0)
                sd = {**{k: self._fresh_value(k, expect[k], gen) for k in missing}, **sd}
        used = set()

        def get(k):
            if k not in sd:
                raise KeyError(f"missing key in state_dict: {k}")
            used.add(k)
            return sd[k]

        def mat(k):
            return get(k).detach().to(device=dev, dtype=torch.bfloat16).contiguous()

        def vec(k):
            return get(k).detach().to(device=dev, dtype=torch.float32).contiguous()

        def bvec(k):        # fp32 storage of a bf16-rounded vector
            return get(k).detach().to(device=dev, dtype=torch.bfloat16).to(torch.float32).contiguous()

        C = self.dim
        w = self._w = {}
        w["pe_w"] = get("patch_embedding.weight").detach().reshape(C, -1).to(device=dev, dtype=torch.bfloat16).contiguous()
        w["pe_b"] = vec("patch_embedding.bias")
        for i in ("0", "2"):
            w[f"te_w{i}"] = mat(f"text_embedding.{i}.weight")
            w[f"te_b{i}"] = vec(f"text_embedding.{i}.bias")
            # The time MLP runs under autocast(float32) in the reference (:913-929) on a model loaded with
            # torch_dtype=bf16 (fast_infer.py:282): fp32 arithmetic on bf16-ROUNDED parameters.  Round once here
            # (like every other matrix), keep fp32 storage for the fp32 torch matmuls of _time_embed.
            w[f"tm_w{i}"] = mat(f"time_embedding.{i}.weight").to(torch.float32)
            w[f"tm_b{i}"] = bvec(f"time_embedding.{i}.bias")
        w["tp_w"] = mat("time_projection.1.weight").to(torch.float32)
        w["tp_b"] = bvec("time_projection.1.bias")
        w["head_w"] = mat("head.head.weight")
        w["head_b"] = vec("head.head.bias")
        w["head_mod"] = vec("head.modulation").reshape(2, C)
        self.blocks = []
        for i in range(self.num_layers):
            p = f"blocks.{i}."
            b = _Block()
            b.w_qk = torch.cat([mat(p + "self_attn.q.weight"), mat(p + "self_attn.k.weight")], dim=0).contiguous()
            b.b_qk = torch.cat([vec(p + "self_attn.q.bias"), vec(p + "self_attn.k.bias")]).contiguous()
            b.w_v, b.b_v = mat(p + "self_attn.v.weight"), vec(p + "self_attn.v.bias")
            b.w_o, b.b_o = mat(p + "self_attn.o.weight"), vec(p + "self_attn.o.bias")
            b.nq, b.nk = vec(p + "self_attn.norm_q.weight"), vec(p + "self_attn.norm_k.weight")
            b.w_cq, b.b_cq = mat(p + "cross_attn.q.weight"), vec(p + "cross_attn.q.bias")
            b.w_ck, b.b_ck = mat(p + "cross_attn.k.weight"), vec(p + "cross_attn.k.bias")
            b.w_cv, b.b_cv = mat(p + "cross_attn.v.weight"), vec(p + "cross_attn.v.bias")
            b.w_co, b.b_co = mat(p + "cross_attn.o.weight"), vec(p + "cross_attn.o.bias")
            b.ncq, b.nck = vec(p + "cross_attn.norm_q.weight"), vec(p + "cross_attn.norm_k.weight")
            b.n3w, b.n3b = vec(p + "norm3.weight"), vec(p + "norm3.bias")
            b.w1, b.b1 = mat(p + "ffn.0.weight"), vec(p + "ffn.0.bias")
            b.w2, b.b2 = mat(p + "ffn.2.weight"), vec(p + "ffn.2.bias")
            b.modulation = vec(p + "modulation").reshape(6, C)
            b.f8 = None
            b._cw = None
            b.spw = None
            self.blocks.append(b)
        w["mod_all"] = torch.stack([b.modulation for b in self.blocks])        # [layers, 6, C]
        self._cdw = None
        extra = [k for k in sd.keys() if k not in used]
        ang = torch.view_as_real(self.freqs)          # [1024, 64, 2] = (cos, sin) of the fp64 angles
        self._rope_dev = (ang[..., 0].to(torch.float32).contiguous().to(dev),
                          ang[..., 1].to(torch.float32).contiguous().to(dev))
        self._device = dev
        self._ctx_cache = None
        self._graph_epoch += 1                         # new weight tensors
        self._reset_attention_scratch()                # the sticky "max-free attempt off" word described the OLD weights' scores
        if self._fp8:
            self.enable_fp8_linear(self._fp8, attn_smooth_k=self.fp8_attn_smooth_k)          # re-quantise from the new bf16 weights
        if getattr(self, "_sp_pad", None) is not None:
            self._pad_heads_for_ulysses(self.sp_world_size)                                  # the padded copies follow the new weights
        return IncompatibleKeys(missing, extra)

    def state_dict(self, *args, **kwargs):  # type: ignore[override]
        """The weights under the reference's key names (views of the packed device tensors: bf16 matrices, fp32
        vectors) -- what ``load_state_dict`` accepts and what ``safetensors.save_file`` can write."""
        if not getattr(self, "_w", None):
            return {}
        C, w = self.dim, self._w
        sd = {"patch_embedding.weight": w["pe_w"].view(C, self.in_dim, *self.patch_size), "patch_embedding.bias": w["pe_b"],
              "time_projection.1.weight": w["tp_w"], "time_projection.1.bias": w["tp_b"],
              "head.head.weight": w["head_w"], "head.head.bias": w["head_b"], "head.modulation": w["head_mod"].view(1, 2, C)}
        for i in ("0", "2"):
            sd[f"text_embedding.{i}.weight"], sd[f"text_embedding.{i}.bias"] = w[f"te_w{i}"], w[f"te_b{i}"]
            sd[f"time_embedding.{i}.weight"], sd[f"time_embedding.{i}.bias"] = w[f"tm_w{i}"], w[f"tm_b{i}"]
        for i, b in enumerate(self.blocks):
            p = f"blocks.{i}."
            sd.update({p + "modulation": b.modulation.view(1, 6, C),
                       p + "self_attn.q.weight": b.w_qk[:C], p + "self_attn.q.bias": b.b_qk[:C],
                       p + "self_attn.k.weight": b.w_qk[C:], p + "self_attn.k.bias": b.b_qk[C:],
                       p + "self_attn.v.weight": b.w_v, p + "self_attn.v.bias": b.b_v,
                       p + "self_attn.o.weight": b.w_o, p + "self_attn.o.bias": b.b_o,
                       p + "self_attn.norm_q.weight": b.nq, p + "self_attn.norm_k.weight": b.nk,
                       p + "cross_attn.q.weight": b.w_cq, p + "cross_attn.q.bias": b.b_cq,
                       p + "cross_attn.k.weight": b.w_ck, p + "cross_attn.k.bias": b.b_ck,
                       p + "cross_attn.v.weight": b.w_cv, p + "cross_attn.v.bias": b.b_cv,
                       p + "cross_attn.o.weight": b.w_co, p + "cross_attn.o.bias": b.b_co,
                       p + "cross_attn.norm_q.weight": b.ncq, p + "cross_attn.norm_k.weight": b.nck,
                       p + "norm3.weight": b.n3w, p + "norm3.bias": b.n3b,
                       p + "ffn.0.weight": b.w1, p + "ffn.0.bias": b.b1, p + "ffn.2.weight": b.w2, p + "ffn.2.bias": b.b2})
        return sd

    def linear_weights(self):
        """Reference module name (``blocks.3.self_attn.q`` ...) -> the packed [out, in] device weight of that
        ``nn.Linear`` (a VIEW: in-place edits take effect at the next forward; used by ``lora_utils.merge_lora``)."""
        if not getattr(self, "_w", None):
            raise RuntimeError("load_state_dict first")
        C, w = self.dim, self._w
        out = {"text_embedding.0": w["te_w0"], "text_embedding.2": w["te_w2"], "head.head": w["head_w"],
               "patch_embedding": w["pe_w"],
               # fp32 storage of bf16-rounded values (see load_state_dict); merge_lora re-rounds them
               "time_embedding.0": w["tm_w0"], "time_embedding.2": w["tm_w2"], "time_projection.1": w["tp_w"]}
        for i, b in enumerate(self.blocks):
            p = f"blocks.{i}."
            out.update({p + "self_attn.q": b.w_qk[:C], p + "self_attn.k": b.w_qk[C:], p + "self_attn.v": b.w_v,
                        p + "self_attn.o": b.w_o, p + "cross_attn.q": b.w_cq, p + "cross_attn.k": b.w_ck,
                        p + "cross_attn.v": b.w_cv, p + "cross_attn.o": b.w_co, p + "ffn.0": b.w1, p + "ffn.2": b.w2})
        return out

    @staticmethod
    def read_checkpoint(path: str, expect: Dict[str, Tuple[int, ...]], device=None, workers: int = 8):
        """The file and shape rules of the reference loader (wan_transformer3d.py:1259-1286): returns ``(kept, skipped)`` -- the
        tensors that will be loaded and the keys dropped by the "Size don't match, skip" rule (not a key of the model, or another
        size) -- after the ``patch_embedding.weight`` channel pad / truncate.  ``device`` (a HIP device): safetensors shards are read
        STRAIGHT onto it, several files at a time (``workers`` threads: the copies release the GIL) -- a 28.6 GB sharded 14B checkpoint
        then costs one pass over the bytes instead of a host copy of everything followed by 1 100 pageable transfers; pickles and
        ``device=None`` take the host path."""
        def pickle_file(fpath):
            obj = torch.load(fpath, map_location="cpu")
            return obj["state_dict"] if isinstance(obj, dict) and "state_dict" in obj else obj

        def load_file(fpath):
            from safetensors.torch import load_file as st_load
            return st_load(fpath) if device is None else st_load(fpath, device=str(torch.device(device)))

        def load_files(files):
            if device is None or len(files) < 2 or workers < 2:
                return [load_file(f) for f in files]
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(workers, len(files))) as pool:
                return list(pool.map(load_file, files))

        model_file = os.path.join(path, "diffusion_pytorch_model.bin")
        model_file_safetensors = model_file.replace(".bin", ".safetensors")
        if os.path.exists(model_file):
            sd = pickle_file(model_file)
        elif os.path.exists(model_file_safetensors):
            sd = load_file(model_file_safetensors)
        else:
            files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
            sd = {}
            for part in load_files(files):
                sd.update(part)
            if not files:
                pickles = sorted(f for ext in ("*.pth", "*.pt", "*.ckpt", "*.bin") for f in glob.glob(os.path.join(path, ext)))
                if not pickles:
                    raise FileNotFoundError(f"no diffusion_pytorch_model.bin, *.safetensors or *.pth under {path}")
                for fpath in pickles:
                    sd.update(pickle_file(fpath))
        sd = dict(sd)
        want = expect["patch_embedding.weight"]
        pe = sd.get("patch_embedding.weight")
        if pe is not None and tuple(pe.shape) != want and pe.dim() == 5 and (pe.shape[0],) + tuple(pe.shape[2:]) == (want[0],) + want[2:]:
            # :1274-1277 -- the checkpoint's input channels land in the leading channels, the rest (if any) is zero
            cin = min(pe.shape[1], want[1])
            fresh = torch.zeros(want, dtype=pe.dtype, device=pe.device)
            fresh[:, :cin] = pe[:, :cin]
            sd["patch_embedding.weight"] = fresh
        kept, skipped = {}, []
        for key, val in sd.items():
            if key in expect and tuple(val.shape) == expect[key]:
                kept[key] = val
            else:
                skipped.append(key)
        return kept, skipped

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, transformer_additional_kwargs={},
                        low_cpu_mem_usage=False, torch_dtype=torch.bfloat16):
        """The reference's checkpoint loader (wan_transformer3d.py:1157-1299), rule by rule:

        * ``config.json`` must exist (RuntimeError otherwise, :1166-1168); ``transformer_additional_kwargs`` override it,
          incl. the ``dict_mapping`` indirection (:1176-1178);
        * weights: ``diffusion_pytorch_model.bin`` (torch pickle, :1260-1261) if present, else
          ``diffusion_pytorch_model.safetensors`` (:1262-1264), else every ``*.safetensors`` shard merged (:1265-1272);
          a ``.pth`` / ``.pt`` / ``.ckpt`` pickle directly under the directory is accepted last (fast_infer.py:286-295 loads
          such files; a ``state_dict`` wrapper key is unwrapped as there);
        * ``patch_embedding.weight`` with another input-channel count is padded with zeros / truncated along dim 1 into
          the fresh model's tensor (:1274-1277);
        * keys the model does not have, or whose size differs, are dropped with the reference's message
          ``<key> Size don't match, skip`` (:1279-1286);
        * what is then missing keeps a fresh model's initial values and is REPORTED, not raised (``strict=False``, :1288-1290).

        ``low_cpu_mem_usage`` only changes how the reference materialises parameters (meta device); any failure there falls
        back to the path above (:1250-1253), so the loaded model is the same and the flag is accepted and ignored.  The
        kernels compute in bf16 (the reference CLI's ``weight_dtype``, fast_infer.py:282): other ``torch_dtype`` values raise."""
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        print(f"loaded 3D transformer's pretrained weights from {path} ...")
        config_file = os.path.join(path, "config.json")
        if not os.path.isfile(config_file):
            raise RuntimeError(f"{config_file} does not exist")
        with open(config_file) as f:
            cfg = json.load(f)
        if torch_dtype not in (None, torch.bfloat16):
            raise NotImplementedError(f"torch_dtype={torch_dtype}: the gfx950 kernels compute in bfloat16 (fast_infer.py:282)")
        extra_kw = dict(transformer_additional_kwargs)
        for key, target in dict(extra_kw.get("dict_mapping", {})).items():
            extra_kw[target] = cfg[key]
        import inspect
        ok = inspect.signature(cls.__init__).parameters
        kw = {k: v for k, v in cfg.items() if k in ok}
        kw.update({k: v for k, v in extra_kw.items() if k in ok})
        model = cls(**kw)

        expect = model.expected_shapes()
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        kept, skipped = cls.read_checkpoint(path, expect, device=dev)
        for key in skipped:
            print(key, "Size don't match, skip")
        m, u = model.load_state_dict(kept, strict=False)
        print(f"### missing keys: {len(m)}; \n### unexpected keys: {len(u)};")
        print(m)
        n_all = sum(int(math.prod(s_)) for s_ in expect.values())
        n_attn = sum(int(math.prod(s_)) for k_, s_ in expect.items() if "self_attn." in k_)
        print(f"### All Parameters: {n_all / 1e6} M")
        print(f"### self-attention Parameters: {n_attn / 1e6} M")
        return model

    # ------------------------------------------------------------------ reference API surface
    def enable_multi_gpus_inference(self):
        """wan_transformer3d.py:802-816: pick up the sequence-parallel group (Ulysses)."""
        sp = get_sp_group()
        if sp is None:
            raise RuntimeError("sequence-parallel group is not initialised (videocof_amd.dist.set_multi_gpus_devices)")
        self._sp, self.sp_world_size, self.sp_world_rank = sp, sp.world_size, sp.rank
        self._sp_pad = None
        if self.num_heads % sp.world_size:
            self._pad_heads_for_ulysses(sp.world_size)

    def _pad_heads_for_ulysses(self, P: int):
        """num_heads % P != 0 (the 12-head 1.3B model on 8 GPUs): the reference would reach for ``ring_degree`` (dist/fuser.py:46-49;
        ring attention is not built here).  Ulysses itself needs whole heads per rank, so the heads are PADDED to the next multiple of P
        and dealt round-robin: head h -> rank h % P, local slot h // P; a rank whose last slot has no head computes a dummy one (q = k =
        v = 0: a uniform softmax over zeros).  All of it is a re-arrangement of WEIGHTS done once, here: the self-attention q / k / v
        projections get their output rows permuted into wire order with zero rows for the dummy heads (so every kernel and every
        exchange of the equal-split path runs unchanged on C_pad channels), the RMSNorm gains follow, with gain and epsilon
        corrected for the mean now being taken over C_pad channels (x / sqrt(sum / C + eps) = x * sqrt(C / C_pad) / sqrt(sum / C_pad +
        eps * C / C_pad)), and the o projection gets the matching column permutation with zero columns.  Cost: the dummy heads' attention
        and the padding on the wires (12 heads on 8 ranks: 16 slots, what 6 ranks would take); the token-local two thirds of a layer
        split P ways regardless."""
        if self._fp8:
            raise NotImplementedError("fp8 projections with padded heads under sequence parallelism")
        H, C, d, dev = self.num_heads, self.dim, self.d, self._device
        Hp, Cp, src, valid = ulysses_head_padding(H, P, d)
        src, valid = src.to(dev), valid.to(dev)
        gain = math.sqrt(C / Cp)

        def rows(w):                         # [C, ...] -> [Cp, ...] in wire order, zero rows for the dummy heads
            out = w.index_select(0, src)
            out[~valid] = 0
            return out.contiguous()
        for blk in self.blocks:
            spw = SimpleNamespace()
            spw.w_q, spw.b_q = rows(blk.w_qk[:C]), rows(blk.b_qk[:C])
            spw.w_k, spw.b_k = rows(blk.w_qk[C:]), rows(blk.b_qk[C:])
            spw.w_v, spw.b_v = rows(blk.w_v), rows(blk.b_v)
            spw.nq, spw.nk = rows(blk.nq) * gain, rows(blk.nk) * gain
            spw.w_o = rows(blk.w_o.t().contiguous()).t().contiguous()          # [C, Cp]: columns in wire order
            blk.spw = spw
        self._sp_pad = SimpleNamespace(H=Hp, C=Cp, eps=self.eps * C / Cp, pad_heads=Hp - H)
        self._bufs, self._bufs_last = {}, None

    def enable_fp8_linear(self, layers=("qkv", "ffn"), attn_smooth_k: bool = True):
        """FP8 (OCP e4m3) projections, SURVEY.md 8f-4 -- an explicit LOSSY option, off by default and never used by a
        parity statement or the headline benchmark.  The reference's fp8 mode (``convert_model_weight_to_float8`` +
        ``convert_weight_dtype_wrapper``, videox_fun/utils/fp8_optimization.py:19-57; ``GPU_memory_mode =
        "model_cpu_offload_and_qfloat8"``) stores weights as e4m3 and up-casts them to bf16 for every matmul; here the named
        projections run on the fp8 matrix pipe: weights e4m3 with one scale per output channel (quantised once, now),
        activations e4m3 with one scale per token row (quantised inside the LN-modulate kernel, or by a row kernel for the
        GELU output), fp32 accumulation.  ``layers``: "qkv" (self-attention q | k and v), "ffn" (ffn.0 and ffn.2), "o" (the self- and
        cross-attention output projections: their bf16 input, the attention kernel's output, takes one row-quantising pass) and
        "cross" (the cross-attention query projection, fed by the quantising form of the norm3 kernel) -- all four together cover
        every per-token Linear of a block, as the reference's fp8 mode does (fp8_optimization.py:19-57); the step-invariant text
        K / V projections stay bf16.  "attn" is not a Linear: it moves the self-attention QK^T product to the fp8 matrix pipe
        (e4m3 q and k with static power-of-two scales ``fp8_attn_exponents``, written by the RMSNorm+RoPE kernel; softmax and
        P.V stay bf16 / fp32) -- the role of the reference's ``sageattn`` branch (attention_utils.py:152-211,
        ``attention_type = "SAGE_ATTENTION"``: 8-bit QK^T, 16-bit P.V).  ``attn_smooth_k`` (default on, as
        ``sageattn``'s ``smooth_k``): the e4m3 copy of k is taken of k minus its per-sample mean over the tokens, which the softmax
        cannot see and which keeps a channel with a large common offset from eating the 3 mantissa bits (one extra 0.5 ms pass
        per layer at the 14B shape); off: the RMSNorm+RoPE kernel writes the e4m3 operands directly.  "attn_pv" (with "attn"): the
        P.V product as well -- V^T is re-quantised per layer into MX e4m3 blocks of 32 keys (``wan_vt_quantize_mx``, 0.23 ms), P inside
        the kernel; SageAttention-2's operating point (``wan_attention_fwd_f8``, include/wan_hip.h a9'').  Under Ulysses sequence
        parallelism the projections run in e4m3 as well (q, k and V^T are projected from the same e4m3 token rows, their weight copies
        split by output rows); the wires stay bf16 and the two attention options run on each rank's arrived operands -- CFG batches
        and head-group pipelining as in the bf16 path (k is quantised once when it has arrived, every q group when its own exchange
        completes).  The e4m3 exponents of "attn" are CALIBRATED per layer on the first forward after this call
        (``fp8_attn_calibrate``: the largest |q| and |k - mean| of that layer's operands, one bit of head-room); under sequence
        parallelism every rank measures its own heads and the ranks agree through one all-reduce(max) per layer (once), so a
        sharded model uses the exponents a single device would have measured.
        The bf16 weights stay loaded (the last block under ``skip_source_frames`` and the sequence-parallel path use them).
        Measured error: tests/test_gpu_fp8.py, DESIGN.md section 13."""
        if getattr(self, "_sp_pad", None) is not None:
            raise NotImplementedError("fp8 projections with padded heads under sequence parallelism (num_heads % ulysses degree != 0)")
        layers = tuple(layers)
        if not set(layers) <= {"qkv", "ffn", "o", "cross", "attn", "attn_pv"} or not layers:
            raise ValueError(f"enable_fp8_linear: layers must be drawn from ('qkv', 'ffn', 'o', 'cross', 'attn', 'attn_pv'), got {layers}")
        if "attn_pv" in layers and "attn" not in layers:
            raise ValueError("enable_fp8_linear: 'attn_pv' (fp8 P.V) extends 'attn' (fp8 QK^T): name both")
        if "attn" in layers and self.d != 128:
            raise NotImplementedError("the fp8 QK^T attention kernel is built for head_dim 128")
        if self.dim % 128 or self.ffn_dim % 128:
            raise NotImplementedError("fp8 projections need dim and ffn_dim to be multiples of 128")
        self.fp8_attn_smooth_k = bool(attn_smooth_k)
        for blk in self.blocks:
            blk.f8 = {}
            if "qkv" in layers:
                blk.f8["qk"], blk.f8["v"] = ops.quantize_weight_fp8(blk.w_qk), ops.quantize_weight_fp8(blk.w_v)
            if "ffn" in layers:
                blk.f8["w1"], blk.f8["w2"] = ops.quantize_weight_fp8(blk.w1), ops.quantize_weight_fp8(blk.w2)
            if "o" in layers:
                blk.f8["o"], blk.f8["co"] = ops.quantize_weight_fp8(blk.w_o), ops.quantize_weight_fp8(blk.w_co)
            if "cross" in layers:
                blk.f8["cq"] = ops.quantize_weight_fp8(blk.w_cq)
            if "attn" in layers:
                blk.f8["attn"] = True
            if "attn_pv" in layers:
                blk.f8["attn_pv"] = True
        self._fp8 = layers
        self._bufs, self._bufs_last = {}, None
        self._graph_epoch += 1              # new e4m3 tensors: a graph captured before must not replay the old ones
        self._reset_attention_scratch()

    def disable_fp8_linear(self):
        for blk in self.blocks:
            blk.f8 = None
        self._fp8 = ()
        self._bufs, self._bufs_last = {}, None
        self._graph_epoch += 1
        self._reset_attention_scratch()

    def _reset_attention_scratch(self):
        """The sticky "max-free attempt off" word of the attention scratches describes the scores of the weights and of the kernel
        family that tripped it: whatever changes either (new weights, a LoRA merge, switching the fp8 attention kernels on or off)
        starts the four call sites afresh, so that a trip of the old configuration never routes the new one through the attempt
        plus a full fix-up for good."""
        for ws in (self._ws_self, self._ws_cross, self._ws_self_sfx, self._ws_cross_sfx):
            ws.reset()

    def clear_context_cache(self):
        """Drop the hoisted text K/V^T (0.8 GB at 14B) and the references that keep the prompt embeddings alive."""
        self._ctx_cache = None

    def enable_teacache(self, coefficients, num_steps: int, rel_l1_thresh: float, num_skip_start_steps: int = 0,
                        offload: bool = True):
        """wan_transformer3d.py:731-741.  Lossy and opt-in (videocof_amd/cache_utils.py); off by default."""
        from .cache_utils import TeaCache
        self.teacache = TeaCache(coefficients, num_steps, rel_l1_thresh=rel_l1_thresh,
                                 num_skip_start_steps=num_skip_start_steps, offload=offload)

    def share_teacache(self, transformer=None):
        self.teacache = transformer.teacache            # :743-747

    def disable_teacache(self):
        self.teacache = None

    def enable_cfg_skip(self, cfg_skip_ratio, num_steps):
        if cfg_skip_ratio:
            raise NotImplementedError("cfg_skip is identity on every supported config (SURVEY.md section 2, row 8)")
        self.cfg_skip_ratio, self.current_steps, self.num_inference_steps = None, 0, None

    def disable_cfg_skip(self):
        self.cfg_skip_ratio, self.current_steps, self.num_inference_steps = None, 0, None

    def share_cfg_skip(self, transformer=None):
        """wan_transformer3d.py:762-768 (the second transformer of a two-stage pipeline takes the first one's counters)."""
        self.cfg_skip_ratio = transformer.cfg_skip_ratio
        self.current_steps = transformer.current_steps
        self.num_inference_steps = transformer.num_inference_steps

    def enable_riflex(self, k=6, L_test=66, L_test_scale=4.886):
        """wan_transformer3d.py:775-789 replaces the temporal RoPE table by the RIFLEx one; only the Gradio UI reaches it
        (ui/wan_ui.py:247-250), no CLI of the path does, and it changes outputs: not built (OUT OF SCOPE)."""
        raise NotImplementedError("RIFLEx temporal frequencies are not built (SURVEY.md section 2: out of scope; no CLI of this path enables them)")

    def disable_riflex(self):
        """wan_transformer3d.py:791-800 restores the default table -- which is the only one this model ever holds."""
        return None

    def unpatchify(self, x, grid_sizes):
        """wan_transformer3d.py:1108-1131: per sample, the first prod(grid) token rows [L, prod(patch) * C_out] -> [C_out, F, H / 8, W / 8]
        (`fhwpqrc->cfphqwr`), on the layout kernel the forward itself uses.  ``grid_sizes``: [B, 3] tensor or sequence of (F, Hp, Wp)."""
        grids = grid_sizes.tolist() if torch.is_tensor(grid_sizes) else [tuple(g) for g in grid_sizes]
        out = []
        for u, v in zip(x, grids):
            o = ops.unpatchify(u.float().contiguous(), tuple(int(i) for i in v), tuple(self.patch_size), self.out_dim,
                               torch.float32 if u.dtype == torch.float32 else torch.bfloat16)
            out.append(o if o.dtype == u.dtype else o.to(u.dtype))
        return out

    # ------------------------------------------------------------------ pieces of forward
    def _time_embed(self, t: torch.Tensor):
        w = self._w
        s = sinusoidal_embedding_1d(self.freq_dim, t.to(self._device)).float()
        e = torch.addmm(w["tm_b2"], torch.nn.functional.silu(torch.addmm(w["tm_b0"], s, w["tm_w0"].t())), w["tm_w2"].t())
        e0 = torch.addmm(w["tp_b"], torch.nn.functional.silu(e), w["tp_w"].t()).unflatten(1, (6, self.dim))
        return e, e0          # fp32 [B,C], [B,6,C]

    def _text_embed(self, context: Sequence[torch.Tensor]) -> torch.Tensor:
        w = self._w
        B = len(context)
        ctx = torch.zeros(B * self.text_len, self.text_dim, device=self._device, dtype=torch.bfloat16)
        for b, u in enumerate(context):
            if u.shape[0] > self.text_len or u.shape[1] != self.text_dim:
                raise ValueError(f"context[{b}] has shape {tuple(u.shape)}; expected [<= {self.text_len}, {self.text_dim}]")
            ctx[b * self.text_len: b * self.text_len + u.shape[0]] = u.to(device=self._device, dtype=torch.bfloat16)
        hid = ops.gemm(ctx, w["te_w0"], w["te_b0"], ops.EPI_GELU_BF16)
        return ops.gemm(hid, w["te_w2"], w["te_b2"], ops.EPI_BF16)       # [B*512, C]

    def _context_kv(self, blk: _Block, ctx: torch.Tensor, B: int, out=None):
        """k [B, text_len, C] (RMS-normed) and v^T [B, C, text_len] of the text tokens (:321-322); `out` = an earlier
        result of the same batch size to overwrite in place (stable addresses for a captured graph)."""
        C, T = self.dim, self.text_len
        ck, cvt = out if out is not None else (None, None)
        ck = ops.gemm(ctx, blk.w_ck, blk.b_ck, ops.EPI_BF16, out=None if ck is None else ck.view(B * T, C))
        ops.rmsnorm_rope_(ck, blk.nck, None, None, self.d, self.eps)
        if cvt is None:
            cvt = torch.empty(B, C, T, device=self._device, dtype=torch.bfloat16)
        for b in range(B):
            ops.gemm(ctx[b * T:(b + 1) * T], blk.w_cv, blk.b_cv, ops.EPI_BF16_T, out=cvt[b])
        return ck.view(B, T, C), cvt

    def _hoisted_context(self, context, B, into=None):
        """Step-invariant text K/V of every block (cache_context), keyed by the IDENTITY of the context tensors; the
        entry keeps them alive, so a freed prompt buffer whose address the allocator hands to the next prompt can never
        match (tensors written by the ctypes kernels all carry _version 0, so (data_ptr, shape, version) is no key).
        On a miss the K/V of the previous prompt (same batch size) -- or the buffers `into`, which a captured graph
        reads -- are overwritten in place."""
        vers = tuple(u._version for u in context)
        if self._ctx_cache is not None:
            held, v0, kv = self._ctx_cache
            if (len(held) == len(context) and all(a is b for a, b in zip(held, context)) and v0 == vers
                    and (into is None or kv is into)):
                return kv
        old = into
        if old is None and self._ctx_cache is not None and self._ctx_cache[2][0][0].shape[0] == B:
            old = self._ctx_cache[2]
        self._ctx_cache = None
        ctx = self._text_embed(context)
        kv = [self._context_kv(blk, ctx, B, out=None if old is None else old[i]) for i, blk in enumerate(self.blocks)]
        self._ctx_cache = (list(context), vers, kv)
        return kv

    def _last_block_suffix(self, blk, em, xs, h, qk, vt, att, cq, ff, ctx_kv, rp, r0, L):
        """The last WanAttentionBlock for B = 1 when only rows >= r0 feed the output: K / V^T are built
        from every token, everything per-query (q, attention, o, cross-attention, FFN) only for the suffix."""
        C, H, Ll = self.dim, self.num_heads, xs.shape[0]
        ops.ln_modulate(xs, em[1], em[0], True, Ll, self.eps, out=h)
        ops.gemm(h, blk.w_qk[C:], blk.b_qk[C:], ops.EPI_BF16, out=qk[:, C:])            # k for all tokens
        ops.gemm(h[r0:], blk.w_qk[:C], blk.b_qk[:C], ops.EPI_BF16, out=qk[r0:, :C])      # q for the suffix
        ops.rmsnorm_rope_(qk[:r0, C:], blk.nk, None, None, self.d, self.eps, self._rope_dev, rp)
        rp2 = type(rp)(rp.F, rp.Hp, rp.Wp, rp.mode, rp.f_src, rp.ground_end, r0, Ll - r0, rp.max_pos)
        ops.rmsnorm_rope_(qk[r0:, :C], blk.nq, qk[r0:, C:], blk.nk, self.d, self.eps, self._rope_dev, rp2, x0_scale=self._qs)
        ops.gemm(h[:L], blk.w_v, blk.b_v, ops.EPI_BF16_T, out=vt[0])
        n = Ll - r0
        ops.attention_fwd(qk[r0:, :C].unsqueeze(0), qk[:, C:].unsqueeze(0), vt, H, k_len=L, out=att[r0:].unsqueeze(0), q_prescaled=True,
                          workspace=self._ws_self_sfx)
        ops.gemm(att[r0:], blk.w_o, blk.b_o, ops.EPI_RESID_F32, out=xs[r0:], gate=em[2], rows_per_batch=n)
        ops.ln_modulate(xs[r0:], blk.n3w, blk.n3b, False, n, self.eps, out=h[r0:])
        ops.gemm(h[r0:], blk.w_cq, blk.b_cq, ops.EPI_BF16, out=cq[r0:])
        ops.rmsnorm_rope_(cq[r0:], blk.ncq, None, None, self.d, self.eps, x0_scale=self._qs)
        ck, cvt = ctx_kv
        ops.attention_fwd(cq[r0:].unsqueeze(0), ck, cvt, H, out=att[r0:].unsqueeze(0), q_prescaled=True, workspace=self._ws_cross_sfx)
        ops.gemm(att[r0:], blk.w_co, blk.b_co, ops.EPI_RESID_F32, out=xs[r0:])
        ops.ln_modulate(xs[r0:], em[4], em[3], True, n, self.eps, out=h[r0:])
        ops.gemm(h[r0:], blk.w1, blk.b1, ops.EPI_GELU_BF16, out=ff[r0:])
        ops.gemm(ff[r0:], blk.w2, blk.b2, ops.EPI_RESID_F32, out=xs[r0:], gate=em[5], rows_per_batch=n)

    def _event_pair(self):
        if self._attn_events is None:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()          # on torch's current stream = the stream the kernel is launched on
        return a, b

    def _event_done(self, ev, rows):
        if ev is not None:
            ev[1].record()
            self._attn_events.append(ev)
            self._last_attn_rows = rows
            self._last_attn_variant = ops.get_tuning("last_attn_variant")      # what the dispatcher launched for THIS call

    def _comm_pair(self, tag):
        """HIP events on the compute stream around one EXPOSED stretch of the Ulysses exchanges; ``_comm_events`` collects
        (tag, start, end) with tag in "q_g0" (the waits before attention: k, V^T -- normally complete under the projections --
        and the first head group of q), "o_g1" (the inverse exchange of the last head group) and "all_gather" (the head output)."""
        if self._comm_events is None:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        return tag, a, b

    def _comm_done(self, ev):
        if ev is not None:
            ev[2].record()
            self._comm_events.append(ev)

    def _rope_map(self, grid, frame_split_indices, ground_frame_indices, token_offset, rows):
        """The temporal position map of rope_apply_qk (:160-179) as kernel parameters."""
        mode, f_src, g_end = 0, 0, 0
        if frame_split_indices is not None and len(frame_split_indices) > 0:
            if len(set(frame_split_indices)) != 1:
                raise NotImplementedError("all samples of a call must share frame_split_indices")
            f_src, mode = int(frame_split_indices[0]), 1
            if ground_frame_indices is not None and len(ground_frame_indices) > 0:
                if len(set(tuple(g) for g in ground_frame_indices)) != 1:
                    raise NotImplementedError("all samples of a call must share ground_frame_indices")
                g0, g1 = ground_frame_indices[0]
                if int(g0) != f_src:
                    raise ValueError("ground frames must start at frame_split_indices (pipeline_wan.py:716-718)")
                mode, g_end = 2, int(g1)
        return RopeParams(grid[0], grid[1], grid[2], mode, f_src, g_end, token_offset, rows, self.freqs.shape[0])

    def _workspaces(self, B, Ll, L, seq_len):
        """Per-forward activation buffers (module docstring).  Kept across calls of one shape: the caching allocator
        would hand the same blocks back anyway, and fixed addresses are what a captured hipGraph replays."""
        Ca = self._sp_pad.C if (self._usp and self._sp_pad is not None) else self.dim       # channels on the attention side of Ulysses
        key = (B, Ll, L, seq_len, self.sp_world_size, self._usp, str(self._device), tuple(self._fp8), Ca)
        b = self._bufs.get(key)
        if b is not None:
            self._bufs_last = key
            return b
        for k in [k for k, v in self._bufs.items() if not getattr(v, "pinned", False)]:
            del self._bufs[k]                                  # eager use keeps one shape; pinned sets belong to captured graphs
        C, dev, M = self.dim, self._device, B * Ll
        b = SimpleNamespace()
        b.h = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
        b.qk = torch.empty(M, 2 * C, device=dev, dtype=torch.bfloat16)
        b.att = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
        b.cq = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
        b.ff = torch.empty(M, self.ffn_dim, device=dev, dtype=torch.bfloat16)
        P = self.sp_world_size
        if not self._usp:
            # V^T pad columns [L, roundup(L, 64)) are never written and must stay finite: zero them once
            b.vt = torch.zeros(B, C, ops.round_up(L, 64), device=dev, dtype=torch.bfloat16)
        else:
            # Ulysses wire buffers (include/wan_hip.h a21), persistent: send / receive pairs for k, q, V^T and o, and the
            # head-sharded V^T the attention kernel reads (pad columns [P*Ll, ld) zeroed once, never written again)
            b.vt = None
            for name in ("kw_s", "kw_r", "qw_s", "qw_r", "vw_s", "vw_r", "ow_s", "ow_r"):
                setattr(b, name, torch.empty(M * Ca, device=dev, dtype=torch.bfloat16))
            b.vt_full = torch.zeros(B, Ca // P, ops.round_up(seq_len, 64), device=dev, dtype=torch.bfloat16)
            if Ca != C:                          # padded heads: the q | k projections and the gathered attention output are Ca wide
                b.qk_a = torch.empty(M, 2 * Ca, device=dev, dtype=torch.bfloat16)
                b.att_a = torch.empty(M, Ca, device=dev, dtype=torch.bfloat16)
        b.qk3 = b.qk.view(B, Ll, 2 * C)
        if self._fp8:
            b.hq = torch.empty(M, C, device=dev, dtype=ops.FP8)
            b.rs = torch.empty(M, device=dev, dtype=torch.float32)
            if "ffn" in self._fp8:
                b.ffq = torch.empty(M, self.ffn_dim, device=dev, dtype=ops.FP8)
                b.ffs = torch.empty(M, device=dev, dtype=torch.float32)
            if "o" in self._fp8:
                b.attq = torch.empty(M, C, device=dev, dtype=ops.FP8)
                b.atts = torch.empty(M, device=dev, dtype=torch.float32)
            if "attn" in self._fp8:
                b.q8 = torch.empty(M, C, device=dev, dtype=ops.FP8)
                b.k8 = torch.empty(M, C, device=dev, dtype=ops.FP8)
                b.kmean = torch.empty(B, C, device=dev, dtype=torch.float32)
                from ._lib import load
                b.kmean_ws = torch.empty(int(load().wan_col_mean_workspace_bytes(B, C)) // 4, device=dev, dtype=torch.float32)
                if "attn_pv" in self._fp8:
                    b.v8 = torch.empty_like(b.vt_full if self._usp else b.vt, dtype=ops.FP8)
                    b.v8s = torch.empty(int(load().wan_vt_mx_scale_bytes(B, self.num_heads // (P if self._usp else 1), L)), device=dev,
                                        dtype=torch.uint8)
        b.pinned = False
        self._bufs[key] = b
        self._bufs_last = key
        return b

    def release_workspaces(self, keep_pinned: bool = False):
        """Free the cached activation buffers (7 GB at 14B / L = 67 080) and the hoisted text K/V.  ``keep_pinned``: leave the
        sets a captured hipGraph replays from (``WanPipeline`` passes it when it owns a ``GraphedForward``); without it every
        captured graph of this model is invalidated (``_graph_epoch``)."""
        if keep_pinned:
            for k in [k for k, v in self._bufs.items() if not getattr(v, "pinned", False)]:
                del self._bufs[k]
        else:
            self._bufs = {}
            self._graph_epoch += 1
        self._bufs_last = None
        self._ctx_cache = None
        # the per-stream GEMM workspaces of eager launches (re-allocated on demand; the composites re-read the address on every call,
        # `_block_cws`).  The per-device CAPTURE workspace stays: graphs bake its address in, and graphs of OTHER models of this process
        # may still replay from it -- `ops.release_gemm_workspaces(include_capture=True)` is for a caller who knows every graph is gone.
        ops.release_gemm_workspaces(include_capture=False)

    def _run_block(self, blk: _Block, em, xs, bufs, ctx_kv, rp, B, Ll, L, seq_len):
        """One WanAttentionBlock (:464-515) in place on the fp32 residual stream xs [B*Ll, C].
        em: [6, B, C] = modulation + e0 (:495); ctx_kv: (k [B,512,C], v^T [B,C,512]) of the text tokens."""
        C, H, P, M = self.dim, self.num_heads, self.sp_world_size, B * Ll
        h, qk, att, cq, ff, vt, qk3 = bufs.h, bufs.qk, bufs.att, bufs.cq, bufs.ff, bufs.vt, bufs.qk3
        usp = self._usp
        f8 = blk.f8 or {}                                       # fp8 Linears (every projection below has both forms, Ulysses included)
        a8_sp = usp and "attn" in f8                            # the fp8 attention products also run on the arrived Ulysses operands
        if not usp and not f8 and self._attn_events is None and self.use_block_composite:
            # the same launch sequence as below, enqueued by ONE C call (wan_dit_block_forward): 1 FFI crossing instead of 15
            self._block_composite(blk, em, xs, bufs, ctx_kv, rp, B, Ll, L)
            return
        # ---- self attention (:495-499)
        if "qk" in f8:
            ops.ln_modulate_fp8(xs, em[1], em[0], True, Ll, self.eps, out=bufs.hq, out_scale=bufs.rs)
        else:
            ops.ln_modulate(xs, em[1], em[0], True, Ll, self.eps, out=h)
        a8 = "attn" in f8                 # QK^T on the fp8 matrix pipe: the norm+rope kernel writes e4m3 q / k instead of bf16
        qe, ke = f8.get("attn_exp") or self.fp8_attn_exponents

        def calibrate(q_bf16, k_bf16, mean, rows_per_batch):
            """Per-layer exponents from this call's operands (first forward after enable_fp8_linear; one host sync per layer, once)."""
            nonlocal qe, ke
            if not (self.fp8_attn_calibrate and "attn_exp" not in f8) or torch.cuda.is_current_stream_capturing():
                return
            # over the L valid rows of every sample only (pad rows [L, Ll) are never attended; the sequence-parallel path below
            # measures the same rows, so both paths pick the exponents of the same operands)
            kc = k_bf16.float().view(B, rows_per_batch, -1)[:, :L]
            if mean is not None:
                kc = kc - mean[:, None, :].float()
            aq, ak = float(q_bf16.float().view(B, rows_per_batch, -1)[:, :L].abs().max()), float(kc.abs().max())
            pick = lambda a: int(max(-8, min(8, math.floor(math.log2(448.0 / max(a, 1e-30))) - 1)))
            qe, ke = pick(aq), pick(ak)
            f8["attn_exp"] = (qe, ke)

        def norm_rope():
            if a8 and self.fp8_attn_smooth_k:
                # K smoothing (sageattn's smooth_k): the bf16 norm + rope as usual, the per-sample token mean of k, then e4m3 q and k - mean
                ops.rmsnorm_rope_(qk[:, :C], blk.nq, qk[:, C:], blk.nk, self.d, self.eps, self._rope_dev, rp, x0_scale=self._qs)
                ops.col_mean(qk[:, C:], Ll, L, B, out=bufs.kmean, workspace=bufs.kmean_ws)
                calibrate(qk[:, :C], qk[:, C:], bufs.kmean, Ll)
                ops.qk_quantize_fp8(qk[:, :C], qk[:, C:], Ll, bufs.kmean, 2.0 ** qe, 2.0 ** ke, bufs.q8, bufs.k8)
            elif a8:
                ops.rmsnorm_rope_fp8(qk[:, :C], blk.nq, qk[:, C:], blk.nk, self.d, self.eps, self._rope_dev, rp, bufs.q8, bufs.k8,
                                     x0_scale=self._qs * 2.0 ** qe, x1_scale=2.0 ** ke)
            else:
                ops.rmsnorm_rope_(qk[:, :C], blk.nq, qk[:, C:], blk.nk, self.d, self.eps, self._rope_dev, rp, x0_scale=self._qs)

        if not usp and "qk" in f8:
            ops.gemm_fp8(bufs.hq, bufs.rs, *f8["qk"], blk.b_qk, ops.EPI_BF16, out=qk)
            norm_rope()
            for b in range(B):
                ops.gemm_fp8(bufs.hq[b * Ll:(b + 1) * Ll][:L], bufs.rs[b * Ll:(b + 1) * Ll][:L], *f8["v"], blk.b_v,
                             ops.EPI_BF16_T, out=vt[b])
        elif not usp:
            ops.gemm(h, blk.w_qk, blk.b_qk, ops.EPI_BF16, out=qk)
            norm_rope()
            for b in range(B):
                ops.gemm(h[b * Ll:(b + 1) * Ll][:L], blk.w_v, blk.b_v, ops.EPI_BF16_T, out=vt[b])
        if not usp:
            ev = self._event_pair()
            if a8 and "attn_pv" in f8:
                ops.vt_quantize_mx(vt, H, L, v8=bufs.v8, scales=bufs.v8s)
                ops.attention_fwd_f8(bufs.q8.view(B, Ll, C), bufs.k8.view(B, Ll, C), bufs.v8, bufs.v8s, vt, H, qe, ke, k_len=L,
                                     out=att.view(B, Ll, C), workspace=self._ws_self)
            elif a8:
                ops.attention_fwd_qk8(bufs.q8.view(B, Ll, C), bufs.k8.view(B, Ll, C), vt, H, qe, ke, k_len=L, out=att.view(B, Ll, C),
                                      workspace=self._ws_self)
            else:
                ops.attention_fwd(qk3[:, :, :C], qk3[:, :, C:], vt, H, k_len=L, out=att.view(B, Ll, C), q_prescaled=True,
                                  workspace=self._ws_self)
            self._event_done(ev, B * Ll)
            o_in = att
        else:
            # Ulysses.  Every projection writes its result straight into the send layout of its own exchange (k, q: the
            # RMSNorm+RoPE kernel's wire output; V^T: the transposed GEMM epilogue with ldo = B * Ll) and is followed at once
            # by that exchange (async, on RCCL's stream): the k exchange runs under the V projection, the V^T exchange under
            # the q projection, only the q exchange is exposed.  The arrived q / k buffers ARE the attention operands
            # ([P*Ll][B][C/P], uniform strides), the attention output IS the send buffer of the inverse exchange; the only
            # re-layout passes per layer are wan_sp_unpack_vt and wan_sp_unpack_heads (2 x M*C bf16 each way).
            # Padded heads (num_heads % P != 0, _pad_heads_for_ulysses): the attention side runs on Ca = C_pad channels / Ha = H_pad heads
            # with the wire-ordered, zero-padded weight copies; otherwise Ca = C and everything below is the model's own tensors.
            pad = self._sp_pad
            Ca, Ha = (pad.C, pad.H) if pad is not None else (C, H)
            eps_a = pad.eps if pad is not None else self.eps
            spw = blk.spw if pad is not None else None
            nq_a, nk_a = (spw.nq, spw.nk) if pad is not None else (blk.nq, blk.nk)
            qk_a = bufs.qk_a if pad is not None else qk
            sp, Cl, Lt = self._sp, Ca // P, P * Ll
            if "qk" in f8:          # e4m3 operands: the q | k weight copy and its per-output-channel scales split by rows like the bf16 one
                w8, ws8 = f8["qk"]
                proj = lambda rows, bias, out: ops.gemm_fp8(bufs.hq, bufs.rs, w8[rows], ws8[rows], bias, ops.EPI_BF16, out=out)
                proj_vt = lambda b, out: ops.gemm_fp8(bufs.hq[b * Ll:(b + 1) * Ll], bufs.rs[b * Ll:(b + 1) * Ll], *f8["v"], blk.b_v,
                                                      ops.EPI_BF16_T, out=out)
            elif pad is not None:
                proj = lambda rows, bias, out: ops.gemm(h, spw.w_k if rows.start else spw.w_q, spw.b_k if rows.start else spw.b_q,
                                                        ops.EPI_BF16, out=out)
                proj_vt = lambda b, out: ops.gemm(h[b * Ll:(b + 1) * Ll], spw.w_v, spw.b_v, ops.EPI_BF16_T, out=out)
            else:
                proj = lambda rows, bias, out: ops.gemm(h, blk.w_qk[rows], bias, ops.EPI_BF16, out=out)
                proj_vt = lambda b, out: ops.gemm(h[b * Ll:(b + 1) * Ll], blk.w_v, blk.b_v, ops.EPI_BF16_T, out=out)
            proj(slice(Ca, 2 * Ca), blk.b_qk[C:], qk_a[:, Ca:])
            ops.rmsnorm_rope_sp(qk_a[:, Ca:], nk_a, None, None, self.d, eps_a, self._rope_dev, rp, bufs.kw_s, None, P, B)
            wait_k = sp.exchange(bufs.kw_r, bufs.kw_s, async_op=True)
            vsend = bufs.vw_s.view(Ca, B, Ll)
            for b in range(B):
                proj_vt(b, vsend[:, b])
            wait_v = sp.exchange(bufs.vw_r, bufs.vw_s, async_op=True)
            proj(slice(0, Ca), blk.b_qk[:C], qk_a[:, :Ca])
            # Head groups (q and o only; k and V^T are already under the projections).  The RMSNorm of q spans ALL heads of a token
            # (wan_transformer3d.py:264-267: WanRMSNorm(dim)), so no head group of q exists before the whole projection does -- the
            # pipeline is between the exchanges and the attention launches, not inside the projection: with groups g0 | g1 the
            # exposed transfers per layer are q(g0) and o(g1), ONE exchange's worth instead of two.
            Hl = Ha // P
            h0 = Hl // 2 if (self.sp_head_groups >= 2 and Hl >= 2) else 0
            split, n0 = h0 * self.d, Lt * B * h0 * self.d        # group 0: channels [0, split) of every slab, n0 elements of wire
            ops.rmsnorm_rope_sp(qk_a[:, :Ca], nq_a, None, None, self.d, eps_a, self._rope_dev, rp, bufs.qw_s, None, P, B,
                                x0_scale=self._qs, split=split)
            if h0:
                wait_q = sp.exchange(bufs.qw_r[:n0], bufs.qw_s[:n0], async_op=True)
                wait_q1 = sp.exchange(bufs.qw_r[n0:], bufs.qw_s[n0:], async_op=True)
            else:
                wait_q = sp.exchange(bufs.qw_r, bufs.qw_s, async_op=True)
            cev = self._comm_pair("q_g0")       # exposed: whatever of the exchanges the projections did not cover (q: group 0 only)
            wait_k()
            wait_v()
            ops.sp_unpack_vt(bufs.vw_r, bufs.vt_full, P, Ll)
            # The head groups of this rank: (first channel, channels, heads, wire elements before it).  A group's wire buffer reads as
            # [P*Ll][B][its channels]; k / V^T of a group are column / row slices of the whole arrived ones.
            groups = [(0, split, h0, 0), (split, Cl - split, Hl - h0, n0)] if h0 else [(0, Cl, Hl, 0)]
            q_waits = [wait_q, wait_q1] if h0 else [wait_q]
            wire_g = lambda w, c0, cg, off: w[off:off + Lt * B * cg].view(Lt, B, cg).permute(1, 0, 2)     # [B, P*Ll, cg]: row stride B*cg, sample stride cg
            k_all = bufs.kw_r.view(Lt, B, Cl).permute(1, 0, 2)
            if a8_sp:
                # fp8 attention on the arrived operands (bf16 wires; every rank holds ALL tokens of its heads, so the K mean is local).
                # The token-major wire [P*Ll][B][Cl] is one [P*Ll, B*Cl] matrix: a column is one (sample, channel) pair, so one column
                # mean / one quantisation pass covers every sample of the CFG batch.  k is quantised once, when it has arrived; every
                # head group of q when ITS exchange completes (so group 1 still travels under the attention of group 0).
                from ._lib import load
                k2d = bufs.kw_r.view(Lt, B * Cl)
                mean = None
                if self.fp8_attn_smooth_k:
                    mean = ops.col_mean(k2d, Lt, L, 1, out=bufs.kmean.view(-1)[:B * Cl].view(1, B * Cl), workspace=bufs.kmean_ws)
                # Whether to enter the collective below must be decided by state that is THE SAME ON EVERY RANK, or the ranks dead-lock:
                # `fp8_attn_calibrate` (set by enable_fp8_linear, which every rank calls alike) and "attn_exp_agreed" (set only here,
                # by the collective itself) -- not by a rank-local preset of "attn_exp" or by whether this rank happens to be capturing.
                if self.fp8_attn_calibrate and not f8.get("attn_exp_agreed"):
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("fp8 attention under sequence parallelism calibrates its exponents on the first EAGER forward "
                                           "(one all-reduce per layer); run one before capturing a graph")
                    # per-layer exponents, agreed by the ranks: each rank holds other heads, so the operands' maxima are reduced
                    # (max) over the group -- the pair every rank then uses is the one a single device would have measured
                    for w_ in q_waits:
                        w_()
                    kc = k2d[:L].float() if mean is None else k2d[:L].float() - mean.float()
                    q_amax = torch.stack([bufs.qw_r[off:off + L * B * cg].float().abs().max() for (_, cg, _, off) in groups]).max()
                    amax = torch.stack([q_amax, kc.abs().max()])       # (over the L valid token rows of every sample)
                    if self.sp_world_size > 1:
                        amax = sp.all_reduce_max(amax)
                    aq, ak = (float(v) for v in amax.tolist())
                    pick = lambda a_: int(max(-8, min(8, math.floor(math.log2(448.0 / max(a_, 1e-30))) - 1)))
                    if "attn_exp" not in f8:            # (a preset pair stays: exponents need not agree across ranks for correctness)
                        qe, ke = pick(aq), pick(ak)
                        f8["attn_exp"] = (qe, ke)
                    f8["attn_exp_agreed"] = True
                k8w = bufs.k8.view(-1)[:Lt * B * Cl]
                ops.qk_quantize_fp8(None, k2d, Lt, mean, 1.0, 2.0 ** ke, None, k8w)
                k8_all = k8w.view(Lt, B, Cl).permute(1, 0, 2)
                pv8 = "attn_pv" in f8
                hs = int(load().wan_vt_mx_scale_bytes(1, 1, L)) if pv8 else 0          # scale bytes per (sample, head)
            wait_o = []
            ev = self._event_pair()
            for gi, (c0, cg, hg, off) in enumerate(groups):
                q_waits[gi]()                   # group 0: exposed; group 1 arrived under the attention of group 0
                if gi == 0:
                    self._comm_done(cev)
                o_g = wire_g(bufs.ow_s, c0, cg, off)
                if a8_sp:
                    q8g = bufs.q8.view(-1)[off:off + Lt * B * cg]
                    ops.qk_quantize_fp8(bufs.qw_r[off:off + Lt * B * cg].view(Lt, B * cg), None, Lt, None, 2.0 ** qe, 1.0, q8g, None)
                    q8v = q8g.view(Lt, B, cg).permute(1, 0, 2)
                    vt_g = bufs.vt_full[:, c0:c0 + cg]
                    if pv8:
                        # a group's V^T is quantised on its own (rows [c0, c0 + cg) of every sample; its scales are a [B][hg] block)
                        v8g, s8g = bufs.v8[:, c0:c0 + cg], bufs.v8s[B * (c0 // self.d) * hs:B * (c0 // self.d + hg) * hs]
                        ops.vt_quantize_mx(vt_g, hg, L, v8=v8g, scales=s8g)
                        ops.attention_fwd_f8(q8v, k8_all[..., c0:c0 + cg], v8g, s8g, vt_g, hg, qe, ke, k_len=L, out=o_g,
                                             workspace=self._ws_self)
                    else:
                        ops.attention_fwd_qk8(q8v, k8_all[..., c0:c0 + cg], vt_g, hg, qe, ke, k_len=L, out=o_g, workspace=self._ws_self)
                else:
                    ops.attention_fwd(wire_g(bufs.qw_r, c0, cg, off), k_all[..., c0:c0 + cg], bufs.vt_full[:, c0:c0 + cg], hg, k_len=L,
                                      out=o_g, q_prescaled=True, workspace=self._ws_self)
                if gi + 1 < len(groups):        # ... the output of group 0 leaves under the attention of group 1
                    wait_o.append(sp.exchange(bufs.ow_r[off:off + Lt * B * cg], bufs.ow_s[off:off + Lt * B * cg], async_op=True))
            self._event_done(ev, B * seq_len)
            cev = self._comm_pair("o_g1")       # exposed: the inverse exchange sits between attention and the o projection (group 1 only)
            c0, cg, hg, off = groups[-1]
            sp.exchange(bufs.ow_r[off:off + Lt * B * cg], bufs.ow_s[off:off + Lt * B * cg])
            for w_ in wait_o:
                w_()
            self._comm_done(cev)
            if pad is not None:                 # [B*Ll, Ca] in wire order; the o projection's padded weight has the matching columns
                ops.sp_unpack_heads(bufs.ow_r, bufs.att_a, P, Ll, B, split=split)
                ops.gemm(bufs.att_a, spw.w_o, blk.b_o, ops.EPI_RESID_F32, out=xs, gate=em[2], rows_per_batch=Ll)
                o_in = None
            else:
                ops.sp_unpack_heads(bufs.ow_r, att, P, Ll, B, split=split)
                o_in = att
        if o_in is None:
            pass
        elif "o" in f8:
            ops.quantize_rows_fp8(o_in, out=bufs.attq, out_scale=bufs.atts)
            ops.gemm_fp8(bufs.attq, bufs.atts, *f8["o"], blk.b_o, ops.EPI_RESID_F32, out=xs, gate=em[2], rows_per_batch=Ll)
        else:
            ops.gemm(o_in, blk.w_o, blk.b_o, ops.EPI_RESID_F32, out=xs, gate=em[2], rows_per_batch=Ll)
        if usp and not f8 and self._attn_events is None and self.use_block_composite:
            # the token-local two thirds of the layer (norm3 -> cross-attention -> FFN) as ONE C call (wan_dit_block_tail_forward):
            # the same launches as below, bit for bit
            from ._lib import check, load
            import ctypes
            self._block_cw(blk)
            self._block_cws(bufs, bufs, B, Ll, L)
            ck, cvt = ctx_kv
            check(load().wan_dit_block_tail_forward(ctypes.c_void_p(xs.data_ptr()), ctypes.c_void_p(em.data_ptr()),
                                                    ctypes.c_void_p(ck.data_ptr()), ctypes.c_void_p(cvt.data_ptr()),
                                                    ctypes.byref(blk._cw), ctypes.byref(bufs.cws), B, Ll, ops._stream()),
                  "wan_dit_block_tail_forward")
            return
        # ---- cross attention (:504), text rows are NOT masked (context_lens=None, :936)
        if "cq" in f8:
            ops.ln_modulate_fp8(xs, blk.n3w, blk.n3b, False, M, self.eps, out=bufs.hq, out_scale=bufs.rs)
            ops.gemm_fp8(bufs.hq, bufs.rs, *f8["cq"], blk.b_cq, ops.EPI_BF16, out=cq)
        else:
            ops.ln_modulate(xs, blk.n3w, blk.n3b, False, M, self.eps, out=h)
            ops.gemm(h, blk.w_cq, blk.b_cq, ops.EPI_BF16, out=cq)
        ops.rmsnorm_rope_(cq, blk.ncq, None, None, self.d, self.eps, x0_scale=self._qs)
        ck, cvt = ctx_kv
        ops.attention_fwd(cq.view(B, Ll, C), ck, cvt, H, out=att.view(B, Ll, C), q_prescaled=True, workspace=self._ws_cross)
        if "co" in f8:
            ops.quantize_rows_fp8(att, out=bufs.attq, out_scale=bufs.atts)
            ops.gemm_fp8(bufs.attq, bufs.atts, *f8["co"], blk.b_co, ops.EPI_RESID_F32, out=xs)
        else:
            ops.gemm(att, blk.w_co, blk.b_co, ops.EPI_RESID_F32, out=xs)
        # ---- FFN (:507-511)
        if "w1" in f8:
            ops.ln_modulate_fp8(xs, em[4], em[3], True, Ll, self.eps, out=bufs.hq, out_scale=bufs.rs)
            ops.gemm_fp8(bufs.hq, bufs.rs, *f8["w1"], blk.b1, ops.EPI_GELU_BF16, out=ff)
            ops.quantize_rows_fp8(ff, out=bufs.ffq, out_scale=bufs.ffs)
            ops.gemm_fp8(bufs.ffq, bufs.ffs, *f8["w2"], blk.b2, ops.EPI_RESID_F32, out=xs, gate=em[5], rows_per_batch=Ll)
        else:
            ops.ln_modulate(xs, em[4], em[3], True, Ll, self.eps, out=h)
            ops.gemm(h, blk.w1, blk.b1, ops.EPI_GELU_BF16, out=ff)
            ops.gemm(ff, blk.w2, blk.b2, ops.EPI_RESID_F32, out=xs, gate=em[5], rows_per_batch=Ll)

    def _block_cw(self, blk: _Block):
        """The block's weights as a ``wan_block_weights`` (built once; the tensors it points at live as long as the block)."""
        from ._lib import BlockWeights
        import ctypes
        if getattr(blk, "_cw", None) is None:
            p = lambda t: ctypes.c_void_p(t.data_ptr())
            blk._cw = BlockWeights(self.dim, self.ffn_dim, self.num_heads, self.text_len, float(self.eps),
                                   p(blk.w_qk), p(blk.w_v), p(blk.w_o), p(blk.w_cq), p(blk.w_co), p(blk.w1), p(blk.w2),
                                   p(blk.b_qk), p(blk.b_v), p(blk.b_o), p(blk.b_cq), p(blk.b_co), p(blk.b1), p(blk.b2),
                                   p(blk.nq), p(blk.nk), p(blk.ncq), p(blk.n3w), p(blk.n3b))
        return blk._cw

    def _block_cws(self, bufs, holder, B, Ll, L):
        """``wan_block_workspace`` over the cached activation buffers, kept on `holder.cws`."""
        from ._lib import BlockWorkspace, load
        import ctypes
        if getattr(holder, "cws", None) is None:
            lib = load()
            p = lambda t: ctypes.c_void_p(t.data_ptr())
            H = self.num_heads
            bufs.nself = int(lib.wan_attention_workspace_bytes(B, Ll, L, H, 128))
            bufs.ncross = int(lib.wan_attention_workspace_bytes(B, Ll, self.text_len, H, 128))
            vt = getattr(bufs, "vt", None)                       # (None under Ulysses: the tail composite does not touch qk / vt)
            holder.cws = BlockWorkspace(p(bufs.h), p(bufs.qk), p(bufs.att), p(bufs.cq), p(bufs.ff), p(vt) if vt is not None else None,
                                        vt.stride(1) if vt is not None else 0, None, bufs.nself, None, bufs.ncross, None, 0)
        # the attention scratches belong to the call sites (AttentionWorkspace objects that may be re-allocated when another
        # shape asks for more): take their CURRENT addresses on every call, never a cached pointer
        holder.cws.attn_ws_self = self._ws_self.get(self._device, max(bufs.nself, 16)).data_ptr()
        holder.cws.attn_ws_cross = self._ws_cross.get(self._device, max(bufs.ncross, 16)).data_ptr()
        # the persistent GEMM's workspace: the same per-device buffer ops.gemm hands to the per-op path (so that the composite and
        # the Python launch sequence run the SAME kernels and stay bit-identical); sized by the widest Linear of the block
        M = B * Ll
        gws = None
        # every GEMM shape of wan_dit_block_forward: the four per-token Linears at M = B * Ll and the V^T projection, which runs per
        # sample at M = L valid tokens (a smaller M can ask for MORE workspace: the split-K form of small shapes) -- the composite must
        # never fall back to another kernel than the per-op path takes (ops.gemm sizes its own request), or the two stop being bit-identical
        for (m_, n_, k_) in ((M, 2 * self.dim, self.dim), (M, self.ffn_dim, self.dim), (M, self.dim, self.ffn_dim), (M, self.dim, self.dim),
                             (L, self.dim, self.dim)):
            t = ops.gemm_workspace(self._device, m_, n_, k_)
            if t is not None and (gws is None or t.numel() > gws.numel()):
                gws = t
        holder.cws.gemm_ws = gws.data_ptr() if gws is not None else None
        holder.cws.gemm_ws_bytes = gws.numel() if gws is not None else 0
        return holder.cws

    def _forward_composite(self, x, emod, ehead, kvs, rp, bufs, B, Ll, L, out_dtype):
        """The token path of forward as ONE C call (wan_dit_forward, include/wan_hip.h a11'): used when nothing hooks into
        the block loop (no TeaCache, probes, fp8 projections, suffix-only last block, kernel events) on a single device."""
        from ._lib import BlockWeights, DitWeights, DitWorkspace, check, load
        import ctypes
        lib, w, n = load(), self._w, self.num_layers
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        pt, ph, pw = self.patch_size
        _, Cin, F, Hh, Ww = x.shape
        if self._cdw is None:
            arr = (BlockWeights * n)(*[self._block_cw(b) for b in self.blocks])
            self._cdw = (DitWeights(n, self.in_dim, self.out_dim, pt, ph, pw, arr, p(w["pe_w"]), p(w["pe_b"]), p(w["head_w"]), p(w["head_b"])), arr)
        if getattr(bufs, "dws", None) is None:
            M, dev = B * Ll, self._device
            bufs.xs = torch.empty(M, self.dim, device=dev, dtype=torch.float32)
            bufs.tok = torch.empty(L, Cin * pt * ph * pw, device=dev, dtype=torch.bfloat16)
            bufs.yt = torch.empty(M, w["head_w"].shape[0], device=dev, dtype=torch.float32)
            bufs.dws = DitWorkspace()
            bufs.dws.x, bufs.dws.tokens, bufs.dws.head_out = bufs.xs.data_ptr(), bufs.tok.data_ptr(), bufs.yt.data_ptr()
        bufs.dws.block = self._block_cws(bufs, bufs, B, Ll, L)
        ck = (ctypes.c_void_p * n)(*[kv[0].data_ptr() for kv in kvs])
        cvt = (ctypes.c_void_p * n)(*[kv[1].data_ptr() for kv in kvs])
        out = torch.empty(B, self.out_dim, F, Hh, Ww, device=self._device, dtype=out_dtype)
        check(lib.wan_dit_forward(p(x), 0 if x.dtype == torch.float32 else 1, p(out), 0 if out_dtype == torch.float32 else 1,
                                  p(emod), p(ehead), ck, cvt, ctypes.byref(self._cdw[0]), ctypes.byref(bufs.dws),
                                  p(self._rope_dev[0]), p(self._rope_dev[1]), ctypes.byref(rp), B, F, Hh, Ww, Ll,
                                  min(int(self.mask_source_frames), F // pt) * pt, ops._stream()), "wan_dit_forward")
        return out

    def _block_composite(self, blk: _Block, em, xs, bufs, ctx_kv, rp, B, Ll, L):
        from ._lib import check, load
        import ctypes
        lib = load()
        self._block_cw(blk)
        self._block_cws(bufs, bufs, B, Ll, L)
        ck, cvt = ctx_kv
        check(lib.wan_dit_block_forward(ctypes.c_void_p(xs.data_ptr()), ctypes.c_void_p(em.data_ptr()),
                                        ctypes.c_void_p(ck.data_ptr()), ctypes.c_void_p(cvt.data_ptr()),
                                        ctypes.byref(blk._cw), ctypes.byref(bufs.cws), ctypes.c_void_p(self._rope_dev[0].data_ptr()),
                                        ctypes.c_void_p(self._rope_dev[1].data_ptr()), ctypes.byref(rp), B, Ll, L,
                                        ops._stream()), "wan_dit_block_forward")

    @torch.no_grad()
    def head_forward(self, x: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
        """``Head.forward`` (:535-548): x fp32 [B, L, C], e fp32 [B, C] (the time embedding, not its projection)
        -> fp32 [B, L, patch_volume * out_dim]."""
        B, Ll, C = x.shape
        w = self._w
        xs = x.to(device=self._device, dtype=torch.float32).reshape(B * Ll, C).contiguous()
        eh = (w["head_mod"][None] + e.to(self._device, torch.float32)[:, None]).permute(1, 0, 2).contiguous()   # [2, B, C]
        h = ops.ln_modulate(xs, eh[1], eh[0], True, Ll, self.eps)
        return ops.gemm(h, w["head_w"], w["head_b"], ops.EPI_F32).view(B, Ll, -1)

    @torch.no_grad()
    def block_forward(self, x: torch.Tensor, e: torch.Tensor, context: torch.Tensor, grid_sizes, block_index: int = 0,
                      frame_split_indices=None, ground_frame_indices=None) -> torch.Tensor:
        """``WanAttentionBlock.forward`` (:464-515) of block ``block_index`` on a given residual stream.
        x fp32 [B, L, C]; e fp32 [B, 6, C] (the time projection; the block adds its own ``modulation``, :494);
        context [B, text_len, C] (already through ``text_embedding``); grid_sizes (F, Hp, Wp) with F*Hp*Wp <= L.
        Returns the new residual stream, fp32 [B, L, C].  Single device only."""
        if self.sp_world_size != 1 or self.force_ulysses:
            raise NotImplementedError("block_forward is the single-device composite")
        self._usp = False
        B, Ll, C = x.shape
        grid = tuple(int(v) for v in grid_sizes)
        L = grid[0] * grid[1] * grid[2]
        if L > Ll or C != self.dim or tuple(e.shape) != (B, 6, C) or tuple(context.shape) != (B, self.text_len, C):
            raise ValueError("block_forward: shapes disagree")
        blk = self.blocks[block_index]
        xs = x.to(device=self._device, dtype=torch.float32).reshape(B * Ll, C).clone()
        em = (blk.modulation[None] + e.to(self._device, torch.float32)).permute(1, 0, 2).contiguous()      # [6, B, C]
        ctx = context.to(device=self._device, dtype=torch.bfloat16).reshape(B * self.text_len, C).contiguous()
        rp = self._rope_map(grid, frame_split_indices, ground_frame_indices, 0, Ll)
        bufs = self._workspaces(B, Ll, L, Ll)
        self._run_block(blk, em, xs, bufs, self._context_kv(blk, ctx, B), rp, B, Ll, L, Ll)
        return xs.view(B, Ll, C)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x, t, context, seq_len, clip_fea=None, y=None, y_camera=None, full_ref=None,
                subject_ref=None, cond_flag=True, frame_split_indices=None, ground_frame_indices=None):
        if not self.blocks:
            raise RuntimeError("weights are not loaded (call load_state_dict / from_pretrained)")
        if any(v is not None for v in (clip_fea, y, y_camera, full_ref, subject_ref)):
            raise NotImplementedError("i2v / camera / reference conditioning are other model families "
                                      "(SURVEY.md section 2, rows 13-14)")
        if isinstance(x, (list, tuple)):
            if len({tuple(u.shape) for u in x}) != 1:
                raise NotImplementedError("samples of one call must share a latent shape")
            x = torch.stack(list(x))
        if not x.is_cuda:
            raise RuntimeError("input latents are on the CPU; the HIP path has no CPU fallback")
        dtype = x.dtype
        self._dtype = dtype if dtype in (torch.bfloat16, torch.float32) else self._dtype
        B, Cin, F, Hh, Ww = x.shape
        pt, ph, pw = self.patch_size
        if Cin != self.in_dim:
            raise ValueError(f"latent has {Cin} channels, model expects {self.in_dim}")
        grid = (F // pt, Hh // ph, Ww // pw)
        L = grid[0] * grid[1] * grid[2]
        if t.dim() != 1:
            raise NotImplementedError("per-token timesteps (Wan2.2-5B) are never produced by WanPipeline")
        if t.numel() != B or len(context) != B:
            raise ValueError(f"batch mismatch: x has {B} samples, t {t.numel()}, context {len(context)}")
        # Samples with DIFFERENT CoF position maps in one call (rope_apply_qk takes them per sample, :160-179; WanPipeline never produces
        # this): the kernels take one map per launch, and a sample's result does not depend on its batch mates, so the call is served
        # group by group -- samples that share a map together, the outputs back in the caller's order.
        if B > 1 and frame_split_indices is not None and len(frame_split_indices) > 0:
            gfi = ground_frame_indices if (ground_frame_indices is not None and len(ground_frame_indices) > 0) else None
            if len(frame_split_indices) != B or (gfi is not None and len(gfi) != B):
                raise ValueError(f"frame_split_indices / ground_frame_indices must have one entry per sample ({B})")
            keys = [(int(frame_split_indices[b]), tuple(int(v) for v in gfi[b]) if gfi is not None else None) for b in range(B)]
            if len(set(keys)) > 1:
                if self.teacache is not None:
                    raise NotImplementedError("TeaCache keeps one residual per call: samples of a call must share their CoF position map")
                outs = [None] * B
                for key in dict.fromkeys(keys):
                    idx = [b for b in range(B) if keys[b] == key]
                    sel = torch.tensor(idx, device=x.device)
                    y = self.forward(x.index_select(0, sel), t.index_select(0, sel.to(t.device)), [context[b] for b in idx], seq_len,
                                     cond_flag=cond_flag, frame_split_indices=[key[0]] * len(idx),
                                     ground_frame_indices=[key[1]] * len(idx) if key[1] is not None else None)
                    for j, b in enumerate(idx):
                        outs[b] = y[j]
                return torch.stack(outs)
        P, rank = self.sp_world_size, self.sp_world_rank
        usp = self._usp = self._sp is not None and (P > 1 or self.force_ulysses)
        if usp:
            # :904-905 pads to a multiple of P; we pad to 8*P so every shard keeps 16-byte aligned rows
            seq_len = int(math.ceil(seq_len / (8 * P))) * 8 * P
        assert L <= seq_len, f"sequence of {L} tokens exceeds seq_len={seq_len}"         # :906
        Ll = seq_len // P
        C, H, dev, w = self.dim, self.num_heads, self._device, self._w
        M = B * Ll

        # -- time / text embeddings ------------------------------------------------------------
        e, e0 = self._time_embed(t)
        emod = (w["mod_all"][:, None] + e0[None]).permute(0, 2, 1, 3).contiguous()      # [layers, 6, B, C]
        ehead = (w["head_mod"][None] + e[:, None]).permute(1, 0, 2).contiguous()         # [2, B, C]

        if self.cache_context:
            ctx_kv = self._hoisted_context(context, B)
        else:
            self._ctx_cache = None
            ctx = self._text_embed(context)
            ctx_kv = [None] * self.num_layers

        r0 = 0          # rows whose output is needed after the last block start here (non-zero only with skip_source_frames)
        if self.skip_source_frames and B == 1 and not usp:
            r0 = min(int(self.skip_source_frames), grid[0]) * grid[1] * grid[2]
        if (not usp and self.use_block_composite and self.use_forward_composite and not self._fp8 and self.teacache is None
                and self._probe_layer is None and self._attn_events is None and r0 == 0
                and dtype in (torch.float32, torch.bfloat16)):
            kvs = ctx_kv if ctx_kv[0] is not None else [self._context_kv(blk, ctx, B) for blk in self.blocks]
            rp = self._rope_map(grid, frame_split_indices, ground_frame_indices, 0, Ll)
            return self._forward_composite(x.contiguous(), emod, ehead, kvs, rp, self._workspaces(B, Ll, L, seq_len), B, Ll, L, dtype)

        # -- patch embedding into the fp32 residual stream (this rank's token rows only) --------
        xs = torch.zeros(M, C, device=dev, dtype=torch.float32)
        lo, hi = rank * Ll, min((rank + 1) * Ll, L)
        for b in range(B):
            tok = ops.patchify(x[b].contiguous() if x.dtype in (torch.float32, torch.bfloat16)
                               else x[b].float().contiguous(), self.patch_size)
            if hi > lo:
                ops.gemm(tok[lo:hi], w["pe_w"], w["pe_b"], ops.EPI_F32, out=xs[b * Ll: b * Ll + (hi - lo)])

        rp = self._rope_map(grid, frame_split_indices, ground_frame_indices, rank * Ll, Ll)

        # -- workspaces ----------------------------------------------------------------------------
        bufs = self._workspaces(B, Ll, L, seq_len)
        h = bufs.h

        # TeaCache (:956-1031): skip the blocks and re-apply the residual they produced the last time they ran
        run_blocks, ori_x = True, None
        if self.teacache is not None:
            run_blocks = self.teacache.decide(e0, cond_flag)
            self.should_calc = run_blocks
            if not run_blocks:
                prev = self.teacache.previous_residual_cond if cond_flag else self.teacache.previous_residual_uncond
                xs.add_(prev[-xs.shape[0]:])
            else:
                ori_x = xs.clone()
        for li, blk in enumerate(self.blocks if run_blocks else ()):
            kv = ctx_kv[li] if ctx_kv[li] is not None else self._context_kv(blk, ctx, B)
            if self._probe_layer == li:
                self._probe = xs.clone()
            if r0 and li == self.num_layers - 1:
                self._last_block_suffix(blk, emod[li], xs, bufs.h, bufs.qk, bufs.vt, bufs.att, bufs.cq, bufs.ff, kv, rp, r0, L)
                break
            self._run_block(blk, emod[li], xs, bufs, kv, rp, B, Ll, L, seq_len)
        if ori_x is not None:
            if cond_flag:
                self.teacache.previous_residual_cond = xs - ori_x
            else:
                self.teacache.previous_residual_uncond = xs - ori_x

        # -- head (:535-548) + unpatchify (:1108-1131)
        if r0:
            yt = torch.zeros(1, Ll, w["head_w"].shape[0], device=dev, dtype=torch.float32)
            ops.ln_modulate(xs[r0:], ehead[1], ehead[0], True, Ll - r0, self.eps, out=h[r0:])
            ops.gemm(h[r0:], w["head_w"], w["head_b"], ops.EPI_F32, out=yt[0, r0:])
        else:
            ops.ln_modulate(xs, ehead[1], ehead[0], True, Ll, self.eps, out=h)
            yt = ops.gemm(h, w["head_w"], w["head_b"], ops.EPI_F32).view(B, Ll, -1)
        if usp:
            cev = self._comm_pair("all_gather")
            yt = self._sp.all_gather_tokens(yt)                                        # :1085-1086
            self._comm_done(cev)
        out_dtype = dtype if dtype in (torch.float32, torch.bfloat16) else torch.float32
        out = torch.empty(B, self.out_dim, grid[0] * pt, grid[1] * ph, grid[2] * pw, device=dev, dtype=out_dtype)
        for b in range(B):
            ops.unpatchify(yt[b], grid, self.patch_size, self.out_dim, out_dtype,
                           zero_frames=min(int(self.mask_source_frames), grid[0]) * pt, out=out[b])
        if self.teacache is not None:
            self.teacache.step_done(cond_flag)                                        # :1101-1104
        return out.to(dtype)
