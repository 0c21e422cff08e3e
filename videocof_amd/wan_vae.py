"""MI355X-native WanVAE behind the reference's ``AutoencoderKLWan`` surface.

Drop-in for ``videox_fun/models/wan_vae.py:620-706`` as ``WanPipeline`` uses it
(``pipeline_wan.py:138,325-326,406-407,424,630,687``): ``encode(x)[0].mode()``,
``decode(z).sample``, ``.config.temporal_compression_ratio`` /
``spatial_compression_ratio`` / ``latent_channels``, ``.dtype``; reference-format
checkpoints (keys ``encoder.conv1.weight`` ..., with or without the ``model.`` prefix).

Every convolution runs in ``wan_conv_cl`` (implicit GEMM on the matrix cores) on channels-last
bf16 activations ``[T, H, W, C]``; RMS_norm+SiLU, the attention softmax and the layout changes at
the boundary are HIP kernels too; the middle AttentionBlock's two products use ``wan_gemm_bf16``.

Temporal chunking: encode 1,4,4,... frames as the reference does (:527-539).  The reference decodes one latent
frame per call (:561-573); here latent frame 0 goes alone (it is the 'Rep' chunk whose upsample3d skips time_conv,
:108-112) and the following frames in chunks of ``decode_chunk`` (default 4): every temporal operator of the decoder is
causal with a 2-frame history, so a chunk of n frames is the same arithmetic on the same operands as n chunks of one --
the results are bit-identical (``tests/test_gpu_vae.py::test_decode_chunking_is_bit_identical``) -- while the 60 x 104
latent-resolution stage gets 4x the workgroups (it filled half the chip) and the launch count drops from 2 072 to ~600.
Each cached CausalConv3d owns a 2-frame history buffer; a missing history frame is the zero padding the reference applies
(:32-38).  'Rep' and the downsample3d seeding (:148-152) are kept literally.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops

__all__ = ["AutoencoderKLWan", "DiagonalGaussianDistribution", "AutoencoderKLOutput", "DecoderOutput"]

_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
         0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
        3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


class DiagonalGaussianDistribution:
    """What ``encode(x)[0]`` / ``.latent_dist`` is (wan_vae.py:655-668 builds diffusers' class of this name from the encoder's 32 channels):
    mean | log-variance halves of dim 1, the log-variance clamped to [-30, 20]; ``mode()`` is what the pipeline takes
    (pipeline_wan.py:406-407), ``sample`` / ``kl`` / ``nll`` complete the surface."""

    def __init__(self, parameters: torch.Tensor, deterministic: bool = False):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.deterministic = deterministic
        self.std, self.var = torch.exp(0.5 * self.logvar), torch.exp(self.logvar)
        if deterministic:
            self.std = self.var = torch.zeros_like(self.mean)

    def mode(self) -> torch.Tensor:
        return self.mean

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        dev = self.mean.device
        noise = torch.randn(self.mean.shape, generator=generator, device=generator.device if generator is not None else dev,
                            dtype=self.mean.dtype).to(dev)          # drawn on the generator's device, as diffusers' randn_tensor does
        return self.mean + self.std * noise

    def kl(self, other: Optional["DiagonalGaussianDistribution"] = None) -> torch.Tensor:
        if self.deterministic:
            return torch.zeros(1)
        if other is None:
            t = self.mean.pow(2) + self.var - 1.0 - self.logvar
        else:
            t = (self.mean - other.mean).pow(2) / other.var + self.var / other.var - 1.0 - self.logvar + other.logvar
        return 0.5 * t.sum(dim=[1, 2, 3])

    def nll(self, sample: torch.Tensor, dims=(1, 2, 3)) -> torch.Tensor:
        if self.deterministic:
            return torch.zeros(1)
        return 0.5 * (math.log(2.0 * math.pi) + self.logvar + (sample - self.mean).pow(2) / self.var).sum(dim=list(dims))


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist

    def __getitem__(self, i):
        return (self.latent_dist,)[i]


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


class _Conv:
    __slots__ = ("w", "b", "cin", "cout", "k")


class AutoencoderKLWan(nn.Module):
    def __init__(self, latent_channels: int = 16, temporal_compression_ratio: int = 4,
                 spatial_compression_ratio: int = 8):
        super().__init__()
        if (latent_channels, temporal_compression_ratio, spatial_compression_ratio) != (16, 4, 8):
            raise NotImplementedError("only the Wan2.1 VAE (z=16, 4x temporal, 8x spatial) is built")
        self.config = SimpleNamespace(latent_channels=16, temporal_compression_ratio=4, spatial_compression_ratio=8)
        self.latent_channels, self.temporal_compression_ratio, self.spatial_compression_ratio = 16, 4, 8
        self.dim, self.z_dim, self.dim_mult, self.nres = 96, 16, [1, 2, 4, 4], 2
        self.tdown = [False, True, True]                 # wan_vae.py:610
        self.tup = self.tdown[::-1]
        self.mean = torch.tensor(_MEAN, dtype=torch.float32)
        self.std = torch.tensor(_STD, dtype=torch.float32)
        self._c: Dict[str, _Conv] = {}
        self._g: Dict[str, torch.Tensor] = {}
        self._hist: Dict[str, torch.Tensor] = {}
        self._seen: Dict[str, bool] = {}
        self._device = torch.device("cpu")
        self._dtype = torch.bfloat16
        self.decode_chunk = 4                            # latent frames per decoder call after the first (1 = the reference's loop)

    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict, strict: bool = True, device=None):  # type: ignore[override]
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if dev.type != "cuda":
            raise RuntimeError("AutoencoderKLWan runs on a HIP device only (no CPU fallback)")
        sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in state_dict.items()}
        self._c, self._g = {}, {}
        for key, v in sd.items():
            if key.endswith(".gamma"):
                self._g[key[:-len(".gamma")]] = v.detach().reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
            elif key.endswith(".weight"):
                name = key[:-len(".weight")]
                w = v.detach().float()
                if w.dim() == 4:
                    w = w[:, :, None]                               # Conv2d -> (1, kh, kw)
                cout, cin, kt, kh, kw = w.shape
                cin_p, cout_p = (cin + 7) // 8 * 8, (cout + 3) // 4 * 4
                wp = torch.zeros(cout_p, kt, kh, kw, cin_p)
                wp[:cout, ..., :cin] = w.permute(0, 2, 3, 4, 1)
                K = kt * kh * kw * cin_p
                Kp = (K + 63) // 64 * 64
                flat = torch.zeros(cout_p, Kp)
                flat[:, :K] = wp.reshape(cout_p, K)
                b = torch.zeros(cout_p)
                b[:cout] = sd[name + ".bias"].detach().float()
                c = _Conv()
                c.w = flat.to(device=dev, dtype=torch.bfloat16).contiguous()
                c.b = b.to(dev)
                c.cin, c.cout, c.k = cin_p, cout_p, (kt, kh, kw)
                self._c[name] = c
        need = ["encoder.conv1", "decoder.conv1", "conv1", "conv2", "encoder.head.2", "decoder.head.2"]
        missing = [n for n in need if n not in self._c]
        if missing and strict:
            raise KeyError(f"missing keys in state_dict: {missing}")
        self._device = dev
        self.clear_cache()
        from .wan_transformer3d import IncompatibleKeys
        return IncompatibleKeys(missing, [])

    @classmethod
    def from_pretrained(cls, pretrained_model_path, additional_kwargs={}):
        """wan_vae.py:684-706: a single .safetensors / .pth file with un-prefixed keys."""
        import inspect
        ok = set(inspect.signature(cls.__init__).parameters) - {"self"}
        model = cls(**{k: v for k, v in dict(additional_kwargs).items() if k in ok})
        if pretrained_model_path.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(pretrained_model_path)
        else:
            sd = torch.load(pretrained_model_path, map_location="cpu")
        model.load_state_dict(sd, strict=False)
        return model

    # ------------------------------------------------------------------ primitives
    def clear_cache(self):
        self._hist, self._seen = {}, {}

    def _push_hist(self, name: str, x: torch.Tensor, keep: int = 2):
        old = self._hist.get(name)
        if x.shape[0] >= keep:
            # a VIEW of the chunk's last frames, not a copy: every op of this module writes a fresh output tensor, so the
            # chunk is never modified after the conv that read it; the view keeps it alive until the next chunk replaces it
            self._hist[name] = x[-keep:]
        else:
            if old is None:
                old = torch.zeros(keep, *x.shape[1:], device=x.device, dtype=x.dtype)
            self._hist[name] = torch.cat([old, x])[-keep:].contiguous()

    def _causal(self, x: torch.Tensor, name: str, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
        """CausalConv3d, stride 1 (wan_vae.py:21-40) with its history."""
        c = self._c[name]
        T, H, W, _ = x.shape
        kt, kh, kw = c.k
        if kt == 1:
            return ops.conv_cl(x, c.w, c.b, c.cout, c.k, out_thw=(T, H, W), pad=(0, kh // 2, kw // 2), resid=resid)
        out = ops.conv_cl(x, c.w, c.b, c.cout, c.k, out_thw=(T, H, W), pad=(kt - 1, kh // 2, kw // 2),
                          hist=self._hist.get(name), resid=resid)
        self._push_hist(name, x)
        return out

    def _res(self, x: torch.Tensor, p: str) -> torch.Tensor:
        h = self._causal(x, p + ".shortcut") if (p + ".shortcut") in self._c else x
        y = ops.rmsnorm_silu_cl(x, self._g[p + ".residual.0"], True)
        y = self._causal(y, p + ".residual.2")
        y = ops.rmsnorm_silu_cl(y, self._g[p + ".residual.3"], True)
        return self._causal(y, p + ".residual.6", resid=h)

    def _attn(self, x: torch.Tensor, p: str) -> torch.Tensor:
        """AttentionBlock (wan_vae.py:227-266): per frame, one head of width C over h*w."""
        T, H, W, C = x.shape
        L = H * W
        Lp = ops.round_up(L, 64)
        y = ops.rmsnorm_silu_cl(x, self._g[p + ".norm"], False)
        qc = self._c[p + ".to_qkv"]
        qkv = ops.conv_cl(y, qc.w, qc.b, qc.cout, qc.k, out_thw=(T, H, W)).view(T, L, 3 * C)
        o = torch.empty(T, L, C, device=x.device, dtype=torch.bfloat16)
        kpad = torch.zeros(Lp, C, device=x.device, dtype=torch.bfloat16)
        for t in range(T):
            kpad[:L] = qkv[t, :, C:2 * C]
            s = ops.gemm(qkv[t, :, :C], kpad, None, ops.EPI_F32)                       # [L, Lp]
            pr = ops.softmax_rows(s, L, Lp, 1.0 / math.sqrt(C))
            vt = ops.transpose_pad(qkv[t, :, 2 * C:], Lp)                              # [C, Lp]
            ops.gemm(pr, vt, None, ops.EPI_BF16, out=o[t])
        pc = self._c[p + ".proj"]
        return ops.conv_cl(o.view(T, H, W, C), pc.w, pc.b, pc.cout, pc.k, out_thw=(T, H, W), resid=x)

    def _resample(self, x: torch.Tensor, p: str, mode: str) -> torch.Tensor:
        T, H, W, C = x.shape
        if mode == "upsample3d":
            if not self._seen.get(p):
                self._seen[p] = True                                       # 'Rep' (:110-112)
            else:
                tc = self._c[p + ".time_conv"]
                name = p + ".time_conv"
                y = ops.conv_cl(x, tc.w, tc.b, tc.cout, tc.k, out_thw=(T, H, W), pad=(2, 0, 0),
                                hist=self._hist.get(name), time_interleave=True)      # [2T,H,W,C]
                self._push_hist(name, x)
                x = y
                T = 2 * T
        rc = self._c[p + ".resample.1"]
        if mode.startswith("up"):
            x = ops.conv_cl(x, rc.w, rc.b, rc.cout, rc.k, out_thw=(T, 2 * H, 2 * W), pad=(0, 1, 1), upsample2x=True)
        else:
            Ho, Wo = (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1
            x = ops.conv_cl(x, rc.w, rc.b, rc.cout, rc.k, out_thw=(T, Ho, Wo), stride=(1, 2, 2), pad=(0, 0, 0))
        if mode == "downsample3d":
            name = p + ".time_conv"
            if not self._seen.get(p):
                self._seen[p] = True                                       # seed, no conv (:150-152)
                self._hist[name] = x[-1:]
            else:
                tc = self._c[name]
                T2 = x.shape[0]
                y = ops.conv_cl(x, tc.w, tc.b, tc.cout, tc.k, out_thw=((T2 + 1 - 3) // 2 + 1, x.shape[1], x.shape[2]),
                                stride=(2, 1, 1), pad=(1, 0, 0), hist=self._hist[name])
                self._hist[name] = x[-1:]
                x = y
        return x

    # ------------------------------------------------------------------ encoder / decoder chunk
    def _encoder_chunk(self, x: torch.Tensor) -> torch.Tensor:
        x = self._causal(x, "encoder.conv1")
        idx = 0
        for i in range(len(self.dim_mult)):
            for _ in range(self.nres):
                x = self._res(x, f"encoder.downsamples.{idx}")
                idx += 1
            if i != len(self.dim_mult) - 1:
                x = self._resample(x, f"encoder.downsamples.{idx}", "downsample3d" if self.tdown[i] else "downsample2d")
                idx += 1
        x = self._res(x, "encoder.middle.0")
        x = self._attn(x, "encoder.middle.1")
        x = self._res(x, "encoder.middle.2")
        x = ops.rmsnorm_silu_cl(x, self._g["encoder.head.0"], True)
        return self._causal(x, "encoder.head.2")

    def _decoder_chunk(self, x: torch.Tensor) -> torch.Tensor:
        x = self._causal(x, "decoder.conv1")
        x = self._res(x, "decoder.middle.0")
        x = self._attn(x, "decoder.middle.1")
        x = self._res(x, "decoder.middle.2")
        idx = 0
        for i in range(len(self.dim_mult)):
            for _ in range(self.nres + 1):
                x = self._res(x, f"decoder.upsamples.{idx}")
                idx += 1
            if i != len(self.dim_mult) - 1:
                x = self._resample(x, f"decoder.upsamples.{idx}", "upsample3d" if self.tup[i] else "upsample2d")
                idx += 1
        x = ops.rmsnorm_silu_cl(x, self._g["decoder.head.0"], True)
        return self._causal(x, "decoder.head.2")

    # ------------------------------------------------------------------ public surface
    @torch.no_grad()
    def _encode_one(self, video: torch.Tensor) -> torch.Tensor:
        """[3,T,H,W] -> params [32,t,h,w] (normalised mean | logvar), wan_vae.py:520-548."""
        if video.shape[2] % 8 or video.shape[3] % 8:
            raise ValueError("`height` and `width` have to be divisible by 8")
        self.clear_cache()
        T = video.shape[1]
        outs = [self._encoder_chunk(ops.video_to_cl(video[:, :1]))]
        for i in range(1, 1 + (T - 1) // 4):
            outs.append(self._encoder_chunk(ops.video_to_cl(video[:, 1 + 4 * (i - 1):1 + 4 * i])))
        out = self._causal(torch.cat(outs), "conv1")
        self.clear_cache()
        params = ops.cl_to_video(out, 2 * self.z_dim, torch.float32, False)
        mu, logvar = params.chunk(2, dim=0)
        mean = self.mean.to(params.device).view(-1, 1, 1, 1)
        inv_std = (1.0 / self.std).to(params.device).view(-1, 1, 1, 1)
        return torch.cat([(mu - mean) * inv_std, logvar], dim=0)

    @torch.no_grad()
    def _decode_one(self, z: torch.Tensor, out_dtype) -> torch.Tensor:
        """[16,t,h,w] -> [3, 1+4(t-1), 8h, 8w] clamped to [-1,1], wan_vae.py:550-575, 669."""
        self.clear_cache()
        mean = self.mean.to(z.device).view(-1, 1, 1, 1)
        std = self.std.to(z.device).view(-1, 1, 1, 1)
        zc = (z.float() * std + mean).permute(1, 2, 3, 0).contiguous().to(torch.bfloat16)
        x = self._causal(zc, "conv2")
        n = max(1, int(self.decode_chunk))
        outs = [self._decoder_chunk(x[:1].contiguous())]
        outs += [self._decoder_chunk(x[i:i + n].contiguous()) for i in range(1, x.shape[0], n)]
        self.clear_cache()
        return ops.cl_to_video(torch.cat(outs), 3, out_dtype, True)

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        if not x.is_cuda:
            raise RuntimeError("input video is on the CPU; the HIP path has no CPU fallback")
        self._dtype = x.dtype if x.dtype in (torch.float32, torch.bfloat16) else self._dtype
        h = torch.stack([self._encode_one(u) for u in x]).to(x.dtype)
        post = DiagonalGaussianDistribution(h)
        return AutoencoderKLOutput(post) if return_dict else (post,)

    def decode(self, z: torch.Tensor, return_dict: bool = True):
        if not z.is_cuda:
            raise RuntimeError("latents are on the CPU; the HIP path has no CPU fallback")
        dec = torch.stack([self._decode_one(u, z.dtype) for u in z])
        return DecoderOutput(dec) if return_dict else (dec,)
