"""CPU oracle for WanVAE encode/decode -- TEST INFRASTRUCTURE ONLY.

Functional fp32 PyTorch restatement of ``videox_fun/models/wan_vae.py`` operating on a
reference-format state dict (keys ``model.encoder...`` as saved by ``AutoencoderKLWan``).
Pinned by ``tests/golden/vae_*.npz`` (captured from the reference by ``oracle/gen_golden_vae.py``)
in ``tests/test_oracle_golden_vae.py``.

Own formulation of the temporal feature cache: the reference keeps, per CausalConv3d, the last
``CACHE_T = 2`` input frames with a None / 1-frame / 2-frame state machine (wan_vae.py:18,
206-221, 322-335); here every cached conv owns a 2-frame history that starts as zeros, which is
the same function (a missing frame is exactly the zero padding the reference applies,
wan_vae.py:32-38).  The two exceptions are restated literally:
  * upsample3d: the first chunk skips ``time_conv`` altogether ('Rep', :108-112) and its frame never
    enters the history (:124-130);
  * downsample3d: the first chunk bypasses ``time_conv`` and seeds a 1-frame history (:148-152);
    later chunks convolve cat([last cached frame, x]) with stride 2 and no padding (:155-163).
All tensors are [C, T, H, W] (batch 1).
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

VAE_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
            0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]      # wan_vae.py:630-633
VAE_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
           3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]            # wan_vae.py:634-637


def rms_norm_c(x: Tensor, gamma: Tensor) -> Tensor:
    """RMS_norm (wan_vae.py:43-58): F.normalize over channels (eps 1e-12) * sqrt(C) * gamma."""
    c = x.shape[0]
    n = x.pow(2).sum(dim=0, keepdim=True).sqrt().clamp_min(1e-12)
    return x / n * math.sqrt(c) * gamma.reshape(c, *([1] * (x.dim() - 1)))


class WanVAEOracle:
    def __init__(self, sd: Dict[str, Tensor], dim: int = 96, z_dim: int = 16,
                 dim_mult: Sequence[int] = (1, 2, 4, 4), num_res_blocks: int = 2,
                 temporal_downsample: Sequence[bool] = (False, True, True)):
        self.sd = {k[len("model."):] if k.startswith("model.") else k: v.float() for k, v in sd.items()}
        self.dim, self.z_dim, self.dim_mult = dim, z_dim, list(dim_mult)
        self.nres = num_res_blocks
        self.tdown = list(temporal_downsample)
        self.tup = self.tdown[::-1]                       # wan_vae.py:504
        self.hist: Dict[str, Tensor] = {}
        self.seen: Dict[str, bool] = {}

    # ------------------------------------------------------------------ primitives
    def clear_cache(self):
        self.hist, self.seen = {}, {}

    def causal_conv(self, x: Tensor, p: str) -> Tensor:
        """CausalConv3d with stride 1 (wan_vae.py:21-40) and the per-conv 2-frame history."""
        w, b = self.sd[p + ".weight"], self.sd[p + ".bias"]
        kt, kh, kw = w.shape[2:]
        if kt == 1:
            return F.conv3d(x[None], w, b)[0]
        h = self.hist.get(p)
        if h is None:
            h = x.new_zeros(x.shape[0], kt - 1, *x.shape[2:])
        xin = torch.cat([h, x], dim=1)
        self.hist[p] = xin[:, -(kt - 1):].clone()
        return F.conv3d(xin[None], w, b, padding=(0, kh // 2, kw // 2))[0]

    def residual_block(self, x: Tensor, p: str) -> Tensor:
        """ResidualBlock (wan_vae.py:190-224): shortcut + conv(silu(norm(conv(silu(norm x)))))."""
        h = self.causal_conv(x, p + ".shortcut") if (p + ".shortcut.weight") in self.sd else x
        y = F.silu(rms_norm_c(x, self.sd[p + ".residual.0.gamma"]))
        y = self.causal_conv(y, p + ".residual.2")
        y = F.silu(rms_norm_c(y, self.sd[p + ".residual.3.gamma"]))
        y = self.causal_conv(y, p + ".residual.6")
        return y + h

    def attention_block(self, x: Tensor, p: str) -> Tensor:
        """AttentionBlock (wan_vae.py:227-266): per-frame single-head attention over h*w, head dim C."""
        c, t, hh, ww = x.shape
        y = rms_norm_c(x, self.sd[p + ".norm.gamma"].reshape(c))
        qkv = F.conv2d(y.permute(1, 0, 2, 3), self.sd[p + ".to_qkv.weight"], self.sd[p + ".to_qkv.bias"])
        q, k, v = qkv.reshape(t, 3, c, hh * ww).unbind(1)            # channel order q | k | v
        s = torch.einsum("tci,tcj->tij", q, k) / math.sqrt(c)
        o = torch.einsum("tij,tcj->tci", torch.softmax(s, dim=-1), v).reshape(t, c, hh, ww)
        o = F.conv2d(o, self.sd[p + ".proj.weight"], self.sd[p + ".proj.bias"])
        return o.permute(1, 0, 2, 3) + x

    def spatial_conv(self, x: Tensor, p: str, mode: str) -> Tensor:
        """The Conv2d of Resample applied per frame (wan_vae.py:143-145): nearest-exact 2x + 3x3
        (:81-83) or ZeroPad2d((0,1,0,1)) + 3x3 stride 2 (:92-94)."""
        w, b = self.sd[p + ".resample.1.weight"], self.sd[p + ".resample.1.bias"]
        f = x.permute(1, 0, 2, 3)
        if mode.startswith("up"):
            f = F.interpolate(f, scale_factor=(2.0, 2.0), mode="nearest-exact")
            f = F.conv2d(f, w, b, padding=1)
        else:
            f = F.conv2d(F.pad(f, (0, 1, 0, 1)), w, b, stride=2)
        return f.permute(1, 0, 2, 3)

    def resample(self, x: Tensor, p: str, mode: str) -> Tensor:
        c, t = x.shape[:2]
        if mode == "upsample3d":
            if not self.seen.get(p):                      # 'Rep': first chunk skips time_conv (:110-112)
                self.seen[p] = True
            else:
                y = self.causal_conv(x, p + ".time_conv")                 # [2C, T, H, W]
                y = y.reshape(2, c, t, *x.shape[2:])
                x = torch.stack((y[0], y[1]), dim=2).reshape(c, 2 * t, *x.shape[2:])   # (:138-141)
        x = self.spatial_conv(x, p, mode)
        if mode == "downsample3d":
            w, b = self.sd[p + ".time_conv.weight"], self.sd[p + ".time_conv.bias"]
            if not self.seen.get(p):                      # first chunk: seed the history, no conv (:150-152)
                self.seen[p] = True
                self.hist[p] = x[:, -1:].clone()
            else:
                xin = torch.cat([self.hist[p], x], dim=1)
                self.hist[p] = x[:, -1:].clone()
                x = F.conv3d(xin[None], w, b, stride=(2, 1, 1))[0]        # (:160-161)
        return x

    # ------------------------------------------------------------------ encoder / decoder (one chunk)
    def encoder_chunk(self, x: Tensor) -> Tensor:
        """Encoder3d.forward (wan_vae.py:322-370) on one temporal chunk."""
        dims = [self.dim * u for u in [1] + self.dim_mult]
        x = self.causal_conv(x, "encoder.conv1")
        idx = 0
        for i in range(len(self.dim_mult)):
            for _ in range(self.nres):
                x = self.residual_block(x, f"encoder.downsamples.{idx}")
                idx += 1
            if i != len(self.dim_mult) - 1:
                x = self.resample(x, f"encoder.downsamples.{idx}", "downsample3d" if self.tdown[i] else "downsample2d")
                idx += 1
        x = self.residual_block(x, "encoder.middle.0")
        x = self.attention_block(x, "encoder.middle.1")
        x = self.residual_block(x, "encoder.middle.2")
        x = F.silu(rms_norm_c(x, self.sd["encoder.head.0.gamma"]))
        return self.causal_conv(x, "encoder.head.2")

    def decoder_chunk(self, x: Tensor) -> Tensor:
        """Decoder3d.forward (wan_vae.py:427-476) on one latent frame."""
        x = self.causal_conv(x, "decoder.conv1")
        x = self.residual_block(x, "decoder.middle.0")
        x = self.attention_block(x, "decoder.middle.1")
        x = self.residual_block(x, "decoder.middle.2")
        idx = 0
        for i in range(len(self.dim_mult)):
            for _ in range(self.nres + 1):
                x = self.residual_block(x, f"decoder.upsamples.{idx}")
                idx += 1
            if i != len(self.dim_mult) - 1:
                x = self.resample(x, f"decoder.upsamples.{idx}", "upsample3d" if self.tup[i] else "upsample2d")
                idx += 1
        x = F.silu(rms_norm_c(x, self.sd["decoder.head.0.gamma"]))
        return self.causal_conv(x, "decoder.head.2")

    # ------------------------------------------------------------------ public
    def encode(self, video: Tensor) -> Tensor:
        """video [3,T,H,W] in [-1,1] -> [2*z, t, h, w]; chunks of 1,4,4,... frames (wan_vae.py:520-548);
        first z channels = normalised mean = ``DiagonalGaussianDistribution.mode()``."""
        self.clear_cache()
        t = video.shape[1]
        outs = [self.encoder_chunk(video[:, :1])]
        for i in range(1, 1 + (t - 1) // 4):
            outs.append(self.encoder_chunk(video[:, 1 + 4 * (i - 1):1 + 4 * i]))
        out = self.causal_conv(torch.cat(outs, dim=1), "conv1")
        mu, logvar = out.chunk(2, dim=0)
        mean = torch.tensor(VAE_MEAN).reshape(-1, 1, 1, 1)
        inv_std = (1.0 / torch.tensor(VAE_STD)).reshape(-1, 1, 1, 1)
        self.clear_cache()
        return torch.cat([(mu - mean) * inv_std, logvar], dim=0)

    def decode(self, z: Tensor) -> Tensor:
        """z [z, t, h, w] -> [3, 1+4(t-1), 8h, 8w], one latent frame per decoder call
        (wan_vae.py:550-575), output clamped to [-1, 1] (:669)."""
        self.clear_cache()
        mean = torch.tensor(VAE_MEAN).reshape(-1, 1, 1, 1)
        inv_std = (1.0 / torch.tensor(VAE_STD)).reshape(-1, 1, 1, 1)
        x = self.causal_conv(z / inv_std + mean, "conv2")
        outs = [self.decoder_chunk(x[:, i:i + 1]) for i in range(x.shape[1])]
        self.clear_cache()
        return torch.cat(outs, dim=1).clamp(-1, 1)
