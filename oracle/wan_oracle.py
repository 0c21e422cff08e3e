"""CPU oracle for the Wan2.1-DiT denoising path  --  TEST INFRASTRUCTURE ONLY.

This file restates, in plain functional PyTorch (fp32 on CPU, fp64 where the
reference uses fp64), the algorithm of the reference's hot path so that the HIP
kernels can be checked on the GPU box where ``/root/reference`` does not exist.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; the product package ``videocof_amd`` never does.

Parity pinning: every function below is checked in ``tests/test_oracle_golden.py``
against fixtures under ``tests/golden/`` that were produced by importing the
reference itself in the build container (``oracle/gen_golden.py``), and -- when
the reference tree is present -- live in ``tests/test_oracle_vs_reference.py``.

The functions are device-agnostic torch: the parity tests run them on the CPU; the
full-size checks (L = 67 080 tokens at 14B width, ``tests/test_gpu_fullsize.py``)
evaluate the SAME functions in fp32/fp64 on the GPU through torch's own kernels --
still independent of ``libwan_hip.so`` -- after checking on a small case that the
on-device evaluation reproduces the CPU one.

All ``file:line`` citations are relative to ``/root/reference``.
State-dict key names are the reference's ``nn.Module`` parameter names.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------
@dataclass
class DiTConfig:
    """Constructor arguments of WanTransformer3DModel that matter on the T2V path
    (videox_fun/models/wan_transformer3d.py:579-604)."""
    dim: int = 2048
    ffn_dim: int = 8192
    num_heads: int = 16
    num_layers: int = 32
    in_dim: int = 16
    out_dim: int = 16
    text_dim: int = 4096
    text_len: int = 512
    freq_dim: int = 256
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    eps: float = 1e-6

    @property
    def head_dim(self) -> int:
        return self.dim // self.num_heads


WAN_1_3B = DiTConfig(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30)
WAN_14B = DiTConfig(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40)


# ----------------------------------------------------------------------------
# a2: sinusoidal time embedding          wan_transformer3d.py:31-41
# ----------------------------------------------------------------------------
def sinusoidal_embedding_1d(dim: int, position: Tensor) -> Tensor:
    half = dim // 2
    pos = position.to(torch.float64)
    inv = torch.pow(torch.tensor(10000.0, dtype=torch.float64),
                    -torch.arange(half, dtype=torch.float64) / half).to(pos.device)
    ang = pos[:, None] * inv[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=1)  # fp64; caller casts (.float())


# ----------------------------------------------------------------------------
# a6: RoPE table                         wan_transformer3d.py:44-52, 692-699
# ----------------------------------------------------------------------------
def rope_axis_dims(head_dim: int) -> Tuple[int, int, int]:
    """Number of complex pairs per axis: (t, h, w) = (c-2*(c//3), c//3, c//3) with
    c = head_dim//2 (wan_transformer3d.py:141); 22/21/21 for head_dim 128."""
    c = head_dim // 2
    return c - 2 * (c // 3), c // 3, c // 3


def rope_angles(head_dim: int, max_pos: int = 1024, theta: float = 10000.0) -> Tensor:
    """fp64 angles [max_pos, head_dim//2]; the reference stores polar(1, angle)
    as complex128.  Axis blocks use their own dim in the exponent
    (rope_params(1024, d-4*(d//6)), rope_params(1024, 2*(d//6)) x2)."""
    d = head_dim
    out = []
    for axis_dim in (d - 4 * (d // 6), 2 * (d // 6), 2 * (d // 6)):
        inv = 1.0 / torch.pow(torch.tensor(theta, dtype=torch.float64),
                              torch.arange(0, axis_dim, 2, dtype=torch.float64) / axis_dim)
        out.append(torch.arange(max_pos, dtype=torch.float64)[:, None] * inv[None, :])
    return torch.cat(out, dim=1)


def temporal_positions(f: int, frame_split: Optional[int],
                       ground: Optional[Tuple[int, int]]) -> List[int]:
    """Temporal RoPE index of every latent frame (wan_transformer3d.py:153-191).

    CoF (split + ground): src -> 1..f_src, every ground frame -> 0, tgt -> 1..f_tgt.
    Paired (split only):  src -> 0..f_src-1, tgt -> 0..f_tgt-1.
    Default:              0..f-1.
    """
    if frame_split is None:
        return list(range(f))
    f_src = frame_split
    if ground is not None:
        g0, g1 = ground
        f_ground = g1 - g0
        f_tgt = f - f_src - f_ground
        return list(range(1, f_src + 1)) + [0] * f_ground + list(range(1, f_tgt + 1))
    f_tgt = f - f_src
    return list(range(f_src)) + list(range(f_tgt))


# ----------------------------------------------------------------------------
# a7: rope_apply                         wan_transformer3d.py:135-205
# ----------------------------------------------------------------------------
def rope_apply(x: Tensor, grid: Tuple[int, int, int], angles: Tensor,
               frame_split: Optional[int] = None,
               ground: Optional[Tuple[int, int]] = None,
               token_offset: int = 0, total_tokens: Optional[int] = None) -> Tensor:
    """x: [L, N, D] (one sample).  Interleaved pairs (2i, 2i+1) are rotated by
    the angle of (pos_t | h | w) for pair i in the (22|21|21) split.  Rows
    >= f*h*w pass through.  Arithmetic in fp64 like the reference's
    complex64 x complex128 product, result cast back to x.dtype.

    token_offset/total_tokens: evaluate rows [token_offset, token_offset+L) of a
    longer sequence (sequence-parallel shard); semantics of
    videox_fun/dist/wan_xfuser.py:22-63 generalised to the CoF position map.
    """
    f, h, w = grid
    L, n, d = x.shape
    c = d // 2
    ct, ch, cw = rope_axis_dims(d)
    seq = f * h * w
    dev = x.device           # the restatement is plain torch: it runs wherever its inputs live (CPU in the parity tests;
    angles = angles.to(dev)  # the full-size checks evaluate the same code in fp32/fp64 on the GPU through torch)
    pos_t = torch.tensor(temporal_positions(f, frame_split, ground), dtype=torch.long, device=dev)
    tok = torch.arange(token_offset, token_offset + L, device=dev)
    valid = tok < seq
    tokc = tok.clamp(max=seq - 1)
    fi = tokc // (h * w)
    hi = (tokc // w) % h
    wi = tokc % w
    ang = torch.cat([angles[pos_t[fi], :ct],
                     angles[hi, ct:ct + ch],
                     angles[wi, ct + ch:ct + ch + cw]], dim=1)  # [L, c] fp64
    cos, sin = ang.cos()[:, None, :], ang.sin()[:, None, :]
    xr = x.to(torch.float32).to(torch.float64).reshape(L, n, c, 2)
    a, b = xr[..., 0], xr[..., 1]
    out = torch.stack([a * cos - b * sin, a * sin + b * cos], dim=-1).reshape(L, n, d)
    out = torch.where(valid[:, None, None], out, x.to(torch.float64))
    return out.to(x.dtype)


# ----------------------------------------------------------------------------
# a5 / a4: norms                         wan_transformer3d.py:214-243
# ----------------------------------------------------------------------------
def rms_norm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """x * rsqrt(mean(x^2, -1) + eps) * w over the FULL channel dim (before the
    head split), wan_transformer3d.py:227-230."""
    return x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps) * weight


def layer_norm(x: Tensor, eps: float, weight: Optional[Tensor] = None,
               bias: Optional[Tensor] = None) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = (x - mu).pow(2).mean(dim=-1, keepdim=True)
    y = (x - mu) * torch.rsqrt(var + eps)
    if weight is not None:
        y = y * weight + bias
    return y


def ln_modulate(x: Tensor, scale: Tensor, shift: Tensor, eps: float) -> Tensor:
    """LN(x; no affine) * (1 + scale) + shift   (wan_transformer3d.py:495,507,547)."""
    return layer_norm(x, eps) * (1 + scale) + shift


def gelu_tanh(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def linear(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    return x @ sd[prefix + ".weight"].t() + sd[prefix + ".bias"]


# ----------------------------------------------------------------------------
# a9: attention                          attention_utils.py:152-211 (SDPA branch)
# ----------------------------------------------------------------------------
_MAX_SCORES = 1 << 29          # score elements materialised at once (2 GiB fp32)


def attention(q: Tensor, k: Tensor, v: Tensor, k_len: Optional[int] = None) -> Tensor:
    """q [Lq,N,D], k,v [Lk,N,D] -> [Lq,N,D]; softmax(q k^T / sqrt(D)) v, non-causal.
    ``k_len`` trims keys like the flash-attn branch (attention_utils.py:95-100)."""
    if k_len is not None:
        k, v = k[:k_len], v[:k_len]
    d = q.shape[-1]
    lq, n, lk = q.shape[0], q.shape[1], k.shape[0]
    step = max(1, min(lq, _MAX_SCORES // max(1, n * lk)))      # query rows per slab; softmax is per row, so slabs are exact
    outs = []
    for r in range(0, lq, step):
        s = torch.einsum("qnd,knd->nqk", q[r:r + step], k) * (1.0 / math.sqrt(d))
        p = torch.softmax(s, dim=-1)
        outs.append(torch.einsum("nqk,knd->qnd", p, v))
    return outs[0] if len(outs) == 1 else torch.cat(outs)


# ----------------------------------------------------------------------------
# a8, a10, a11, a12: one WanAttentionBlock  wan_transformer3d.py:464-515
# ----------------------------------------------------------------------------
def self_attention(h: Tensor, sd, pre: str, cfg: DiTConfig, grid, angles,
                   frame_split, ground, seq_len_valid: int) -> Tensor:
    """WanSelfAttention.forward (wan_transformer3d.py:271-305) for one sample [L,C]."""
    L = h.shape[0]
    n, d = cfg.num_heads, cfg.head_dim
    q = rms_norm(linear(h, sd, pre + ".q"), sd[pre + ".norm_q.weight"], cfg.eps).view(L, n, d)
    k = rms_norm(linear(h, sd, pre + ".k"), sd[pre + ".norm_k.weight"], cfg.eps).view(L, n, d)
    v = linear(h, sd, pre + ".v").view(L, n, d)
    q = rope_apply(q, grid, angles, frame_split, ground)
    k = rope_apply(k, grid, angles, frame_split, ground)
    o = attention(q, k, v)  # SDPA branch ignores k_lens; no BASELINE config pads
    return linear(o.reshape(L, n * d), sd, pre + ".o")


def cross_attention(h: Tensor, ctx: Tensor, sd, pre: str, cfg: DiTConfig) -> Tensor:
    """WanT2VCrossAttention.forward (wan_transformer3d.py:308-336); context rows are
    the 512 zero-padded text rows, NOT masked (context_lens=None, :936)."""
    L = h.shape[0]
    n, d = cfg.num_heads, cfg.head_dim
    q = rms_norm(linear(h, sd, pre + ".q"), sd[pre + ".norm_q.weight"], cfg.eps).view(L, n, d)
    k = rms_norm(linear(ctx, sd, pre + ".k"), sd[pre + ".norm_k.weight"], cfg.eps).view(-1, n, d)
    v = linear(ctx, sd, pre + ".v").view(-1, n, d)
    o = attention(q, k, v)
    return linear(o.reshape(L, n * d), sd, pre + ".o")


def block_forward(x: Tensor, e0: Tensor, ctx: Tensor, sd, i: int, cfg: DiTConfig,
                  grid, angles, frame_split, ground, seq_len_valid: int) -> Tensor:
    """x [L,C]; e0 [6,C] (time projection of this sample); ctx [512,C]."""
    pre = f"blocks.{i}"
    e = sd[pre + ".modulation"][0] + e0                        # :492
    h = ln_modulate(x, e[1], e[0], cfg.eps)                    # :495
    y = self_attention(h, sd, pre + ".self_attn", cfg, grid, angles,
                       frame_split, ground, seq_len_valid)
    x = x + y * e[2]                                           # :499
    h = layer_norm(x, cfg.eps, sd[pre + ".norm3.weight"], sd[pre + ".norm3.bias"])
    x = x + cross_attention(h, ctx, sd, pre + ".cross_attn", cfg)   # :504
    h = ln_modulate(x, e[4], e[3], cfg.eps)                    # :507
    y = linear(gelu_tanh(linear(h, sd, pre + ".ffn.0")), sd, pre + ".ffn.2")
    return x + y * e[5]                                        # :511


# ----------------------------------------------------------------------------
# a1, a3, a13, a14, a15: full forward    wan_transformer3d.py:818-1105
# ----------------------------------------------------------------------------
def patchify(x: Tensor, cfg: DiTConfig) -> Tuple[Tensor, Tuple[int, int, int]]:
    """[Cin,F,H,W] -> tokens [L, Cin*pt*ph*pw] in (f,h,w) order with the K index
    (c, pt, ph, pw), c slowest -- the im2col of Conv3d(k=s=patch) (:662-663,870-879)."""
    pt, ph, pw = cfg.patch_size
    c, f, hh, ww = x.shape
    g = (f // pt, hh // ph, ww // pw)
    t = x.reshape(c, g[0], pt, g[1], ph, g[2], pw).permute(1, 3, 5, 0, 2, 4, 6)
    return t.reshape(g[0] * g[1] * g[2], c * pt * ph * pw), g


def unpatchify(u: Tensor, grid, cfg: DiTConfig) -> Tensor:
    """[L, pt*ph*pw*Cout] (c fastest) -> [Cout, F*pt, H*ph, W*pw]  (:1108-1131)."""
    pt, ph, pw = cfg.patch_size
    f, h, w = grid
    c = cfg.out_dim
    u = u[: f * h * w].reshape(f, h, w, pt, ph, pw, c).permute(6, 0, 3, 1, 4, 2, 5)
    return u.reshape(c, f * pt, h * ph, w * pw)


def time_embed(t: Tensor, sd, cfg: DiTConfig) -> Tuple[Tensor, Tensor]:
    """e [B,C], e0 [B,6,C]  (:913-929): fp64 sinusoid -> .float() -> Linear,SiLU,Linear."""
    s = sinusoidal_embedding_1d(cfg.freq_dim, t).float()
    e = linear(F.silu(linear(s, sd, "time_embedding.0")), sd, "time_embedding.2")
    e0 = linear(F.silu(e), sd, "time_projection.1").unflatten(1, (6, cfg.dim))
    return e, e0


def text_embed(context: Sequence[Tensor], sd, cfg: DiTConfig) -> Tensor:
    """zero-pad each prompt to text_len rows, Linear-GELU(tanh)-Linear (:936-942)."""
    ctx = torch.stack([torch.cat([u, u.new_zeros(cfg.text_len - u.shape[0], u.shape[1])])
                       for u in context])
    return linear(gelu_tanh(linear(ctx, sd, "text_embedding.0")), sd, "text_embedding.2")


def head_forward(x: Tensor, e: Tensor, sd, cfg: DiTConfig) -> Tensor:
    """Head.forward (:535-548): modulation [1,2,C] + e (time embedding, NOT e0)."""
    m = sd["head.modulation"][0] + e[None, :]
    return linear(ln_modulate(x, m[1], m[0], cfg.eps), sd, "head.head")


class TeaCacheOracle:
    """Restatement of TeaCache (videox_fun/models/cache_utils.py:21-76) and of its hook in the forward
    (wan_transformer3d.py:956-1031, 1101-1104), conditional branch only (WanPipeline batches cond/uncond in one call)."""

    def __init__(self, coefficients, num_steps: int, rel_l1_thresh: float, num_skip_start_steps: int):
        self.poly = np.poly1d(coefficients)
        self.num_steps, self.thresh, self.skip_start = num_steps, rel_l1_thresh, num_skip_start_steps
        self.cnt, self.acc, self.prev_e0, self.residual = 0, 0.0, None, None
        self.decisions: List[bool] = []

    def run_blocks(self, e0: Tensor) -> bool:
        if self.cnt < self.skip_start:                       # :962-965
            calc, self.acc = True, 0.0
        else:                                                # :967-974
            d = float((e0 - self.prev_e0).abs().mean() / self.prev_e0.abs().mean())
            self.acc += float(self.poly(d))
            calc = not (self.acc < self.thresh)
            if calc:
                self.acc = 0.0
        self.prev_e0 = e0
        self.decisions.append(calc)
        return calc

    def done(self) -> None:                                  # :1101-1104
        self.cnt += 1
        if self.cnt == self.num_steps:
            self.cnt, self.acc, self.prev_e0, self.residual = 0, 0.0, None, None


def dit_forward(sd: Dict[str, Tensor], cfg: DiTConfig, x: Tensor, t: Tensor,
                context: Sequence[Tensor], seq_len: int,
                frame_split_indices: Optional[List[int]] = None,
                ground_frame_indices: Optional[List[Tuple[int, int]]] = None,
                return_tokens: bool = False, teacache: Optional[TeaCacheOracle] = None) -> Tensor:
    """WanTransformer3DModel.forward for the T2V/CoF path.  x [B,Cin,F,H,W]."""
    B = x.shape[0]
    angles = rope_angles(cfg.head_dim)
    w_pe = sd["patch_embedding.weight"].reshape(cfg.dim, -1)
    e, e0 = time_embed(t, sd, cfg)
    ctx = text_embed(context, sd, cfg)
    outs = []
    calc = teacache.run_blocks(e0) if teacache is not None else True
    residual = []
    for b in range(B):
        tok, grid = patchify(x[b], cfg)
        L = tok.shape[0]
        assert L <= seq_len                                   # :906
        h = tok @ w_pe.t() + sd["patch_embedding.bias"]
        if seq_len > L:                                       # zero pad (:907-910)
            h = torch.cat([h, h.new_zeros(seq_len - L, cfg.dim)])
        fs = frame_split_indices[b] if frame_split_indices is not None and b < len(frame_split_indices) else None
        gr = ground_frame_indices[b] if (fs is not None and ground_frame_indices is not None
                                         and b < len(ground_frame_indices)) else None
        if calc:
            h_in = h
            for i in range(cfg.num_layers):
                h = block_forward(h, e0[b], ctx[b], sd, i, cfg, grid, angles, fs, gr, L)
            residual.append(h - h_in)                         # :1028-1031
        else:
            h = h + teacache.residual[b]                      # :983-984
        y = head_forward(h, e[b], sd, cfg)
        outs.append(y if return_tokens else unpatchify(y, grid, cfg))
    if teacache is not None:
        if calc:
            teacache.residual = residual
        teacache.done()
    return torch.stack(outs)


# ----------------------------------------------------------------------------
# a17: FlowUniPCMultistepScheduler       fm_solvers_unipc.py
# ----------------------------------------------------------------------------
class UniPCOracle:
    """Restatement of the flow-matching UniPC (bh2, predict_x0, solver_order 2,
    lower_order_final) as used by fast_infer.py:328-337 / pipeline_wan.py:613-615,740.
    Scalar math follows the reference's fp32 torch scalars so trajectories match
    to rounding."""

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2):
        self.N = num_train_timesteps
        self.solver_order = solver_order
        alphas = np.linspace(1, 1 / self.N, self.N)[::-1].copy()
        sig = torch.from_numpy(1.0 - alphas).to(torch.float32)
        self.sigma_max, self.sigma_min = sig[0].item(), sig[-1].item()   # :128-130 (shift=1 at ctor)

    def set_timesteps(self, n: int, shift: float):
        sig = np.linspace(self.sigma_max, self.sigma_min, n + 1).copy()[:-1]   # :185-188
        sig = shift * sig / (1 + (shift - 1) * sig)                            # :195-196
        self.timesteps = torch.from_numpy(sig * self.N).to(torch.int64)        # :208-211 (truncation)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.model_outputs: List[Optional[Tensor]] = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.step_index = 0
        self.this_order = None

    # -- shared coefficient algebra (:378-470 / :520-612) --------------------
    def _lam(self, s):
        return torch.log(1 - s) - torch.log(s)

    def _update(self, x, m0, m_prev, sigma_t, sigma_s0, sigma_prev, order, model_t=None):
        alpha_t = 1 - sigma_t
        h = self._lam(sigma_t) - self._lam(sigma_s0)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        B_h = torch.expm1(hh)
        x_t_ = sigma_t / sigma_s0 * x - alpha_t * h_phi_1 * m0
        res = 0
        if model_t is None:       # predictor, UniP
            if order == 2:
                rk = (self._lam(sigma_prev) - self._lam(sigma_s0)) / h
                res = 0.5 * ((m_prev - m0) / rk)              # rhos_p = [0.5]  (:437-438)
        else:                      # corrector, UniC
            if order == 1:
                res = 0.5 * (model_t - m0)                    # rhos_c = [0.5]  (:597-598)
            else:
                rk = (self._lam(sigma_prev) - self._lam(sigma_s0)) / h
                rks = torch.stack([rk, torch.tensor(1.0)])
                R, b = [], []
                h_phi_k = h_phi_1 / hh - 1
                fact = 1
                for i in range(1, order + 1):
                    R.append(torch.pow(rks, i - 1))
                    b.append(h_phi_k * fact / B_h)
                    fact *= i + 1
                    h_phi_k = h_phi_k / hh - 1 / fact
                rhos = torch.linalg.solve(torch.stack(R), torch.stack(b)).to(x.dtype)
                res = rhos[0] * ((m_prev - m0) / rk) + rhos[1] * (model_t - m0)
        return (x_t_ - alpha_t * B_h * res).to(x.dtype)

    def step(self, v: Tensor, sample: Tensor) -> Tensor:
        i = self.step_index
        sig = self.sigmas
        x0 = sample - sig[i] * v                                   # convert_model_output :318-320
        if i > 0 and self.last_sample is not None:                 # corrector :689-703
            sample = self._update(self.last_sample, self.model_outputs[-1],
                                  self.model_outputs[-2], sig[i], sig[i - 1],
                                  sig[i - 2] if i >= 2 else None, self.this_order, model_t=x0)
        self.model_outputs = self.model_outputs[1:] + [x0]
        order = min(self.solver_order, len(self.timesteps) - i)    # lower_order_final :713-716
        self.this_order = min(order, self.lower_order_nums + 1)    # :720
        self.last_sample = sample
        prev = self._update(sample, x0, self.model_outputs[-2], sig[i + 1], sig[i],
                            sig[i - 1] if i >= 1 else None, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev


# ----------------------------------------------------------------------------
# a16: the CoF denoise loop               pipeline_wan.py:630-740
# ----------------------------------------------------------------------------
def cof_layout(source_frames: int, reasoning_frames: int, ratio: int = 4) -> Tuple[int, int]:
    """(condition_count, ground_latent_count)  pipeline_wan.py:630-631, 637."""
    cc = 1 if source_frames == 1 else (source_frames - 1) // ratio + 1
    g = 1 if reasoning_frames <= 1 else (reasoning_frames - 1) // ratio + 1
    return cc, g


def cof_denoise(sd, cfg: DiTConfig, src_latents: Tensor, noise: Tensor,
                context: Sequence[Tensor], steps: int, shift: float,
                condition_count: int, ground_count: int,
                guidance_scale: float = 1.0,
                negative_context: Optional[Sequence[Tensor]] = None,
                model_fn=None) -> List[Tensor]:
    """Returns latents after every step.  src_latents [B,16,Fs,h,w], noise
    [B,16,Fs+G,h,w] (pipeline_wan.py:411-417)."""
    lat = torch.cat([src_latents, noise], dim=2)
    B, _, Ftot, hh, ww = lat.shape
    seq_len = math.ceil(hh * ww / (cfg.patch_size[1] * cfg.patch_size[2]) * Ftot)   # :689
    sched = UniPCOracle()
    sched.set_timesteps(steps, shift)
    cfg_on = guidance_scale > 1.0
    out = []
    fwd = model_fn or (lambda x, t, c, fsi, gfi: dit_forward(sd, cfg, x, t, c, seq_len, fsi, gfi))
    for t in sched.timesteps:
        inp = torch.cat([lat] * 2) if cfg_on else lat
        nb = inp.shape[0]
        ctx = (list(negative_context) + list(context)) if cfg_on else list(context)
        v = fwd(inp, t.expand(nb), ctx, [condition_count] * nb,
                [(condition_count, condition_count + ground_count)] * nb)
        if cfg_on:
            vu, vt = v.chunk(2)
            v = vu + guidance_scale * (vt - vu)                  # :731-733
        v = v.clone()
        v[:, :, :condition_count] = 0                            # :736
        lat = sched.step(v, lat)
        out.append(lat)
    return out
