#!/usr/bin/env python3
"""Generate tests/golden/vae_*.npz by running the REFERENCE WanVAE (build container only).
Real channel widths (dim 96, z 16, 126.9 M parameters), small spatial extents.  Goldens (9) and
(10) of SURVEY.md section 8c."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_import import load_reference                                  # noqa: E402
from videocof_amd.weights import deterministic_vae_state_dict, det_uniform     # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


@torch.no_grad()
def main():
    torch.set_num_threads(8)
    ns = load_reference()
    vae = ns.vae.AutoencoderKLWan()
    sd = deterministic_vae_state_dict()
    missing, unexpected = vae.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    vae.eval()
    m = vae.model

    # (9a) CausalConv3d over three chunks (1, 1, 2 frames) with the reference's cache protocol
    conv = m.decoder.upsamples[8].residual[2]            # 192 -> 192, 3x3x3
    x = det_uniform("v9.conv.x", (1, 192, 4, 6, 10), 1.0)
    cache, outs = None, []
    for sl in (slice(0, 1), slice(1, 2), slice(2, 4)):
        xc = x[:, :, sl]
        cx = xc[:, :, -2:].clone()
        if cx.shape[2] < 2 and cache is not None:
            cx = torch.cat([cache[:, :, -1:], cx], dim=2)
        outs.append(conv(xc, cache))
        cache = cx
    save("vae_g9_causal_conv", x=x, out=torch.cat(outs, dim=2), full=conv(x))

    # (9b) ResidualBlock with shortcut (192 -> 384), 2 chunks; AttentionBlock; Resample up3d / up2d / down3d / down2d
    def run_chunks(layer, x, splits, n_cache):
        fc, outs = [None] * n_cache, []
        for sl in splits:
            outs.append(layer(x[:, :, sl], fc, [0]))
        return torch.cat(outs, dim=2)

    rb = m.decoder.upsamples[4]
    xr = det_uniform("v9.rb.x", (1, 192, 3, 6, 10), 1.0)
    save("vae_g9_resblock", x=xr, out=run_chunks(rb, xr, (slice(0, 1), slice(1, 3)), 2))
    ab = m.decoder.middle[1]
    xa = det_uniform("v9.ab.x", (1, 384, 2, 5, 7), 1.0)
    save("vae_g9_attn", x=xa, out=ab(xa))
    up3 = m.decoder.upsamples[3]
    xu = det_uniform("v9.up3.x", (1, 384, 3, 4, 6), 1.0)
    save("vae_g9_up3d", x=xu, out=run_chunks(up3, xu, (slice(0, 1), slice(1, 2), slice(2, 3)), 1))
    up2 = m.decoder.upsamples[11]
    xu2 = det_uniform("v9.up2.x", (1, 192, 2, 4, 6), 1.0)
    save("vae_g9_up2d", x=xu2, out=up2(xu2, [None], [0]))
    dn3 = m.encoder.downsamples[5]
    xd = det_uniform("v9.dn3.x", (1, 192, 5, 6, 10), 1.0)
    save("vae_g9_down3d", x=xd, out=run_chunks(dn3, xd, (slice(0, 1), slice(1, 5)), 1))
    dn2 = m.encoder.downsamples[2]
    xd2 = det_uniform("v9.dn2.x", (1, 96, 2, 7, 10), 1.0)          # odd height: exercises ZeroPad2d((0,1,0,1))
    save("vae_g9_down2d", x=xd2, out=dn2(xd2, [None], [0]))

    # (10) full encode (T = 9 -> chunks 1,4,4; and T = 1) and decode (t = 3 and t = 1)
    video = det_uniform("v10.video", (1, 3, 9, 32, 48), 1.0)
    lat = vae.encode(video)[0].parameters
    save("vae_g10_encode", video=video, params=lat, mode=vae.encode(video)[0].mode(),
         params_t1=vae.encode(video[:, :, :1])[0].parameters)
    z = det_uniform("v10.z", (1, 16, 3, 4, 6), 1.5)
    save("vae_g10_decode", z=z, out=vae.decode(z).sample, out_t1=vae.decode(z[:, :, :1]).sample)
    print("temporal/spatial ratios:", vae.config.temporal_compression_ratio, vae.config.spatial_compression_ratio)


if __name__ == "__main__":
    main()
