"""WanVAE CPU oracle vs fixtures captured from the reference VAE (oracle/gen_golden_vae.py).
fp32 vs fp32: rel-L2 <= 1e-5."""
import pytest
import torch

from oracle.vae_oracle import WanVAEOracle
from videocof_amd.weights import deterministic_vae_state_dict, vae_param_shapes


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def vae():
    return WanVAEOracle(deterministic_vae_state_dict())


def test_param_inventory():
    s = vae_param_shapes()
    assert len(s) == 194
    assert sum(int(torch.tensor(v).prod()) for v in s.values()) == 126892531     # 126.9 M
    assert s["model.decoder.upsamples.3.time_conv.weight"] == (768, 384, 3, 1, 1)
    assert s["model.decoder.upsamples.4.shortcut.weight"] == (384, 192, 1, 1, 1)
    assert s["model.encoder.downsamples.2.resample.1.weight"] == (96, 96, 3, 3)


def test_g9_causal_conv_chunked_equals_full(golden, vae):
    g = golden("vae_g9_causal_conv")
    x = torch.from_numpy(g["x"])[0]
    p = "decoder.upsamples.8.residual.2"
    vae.clear_cache()
    outs = [vae.causal_conv(x[:, sl], p) for sl in (slice(0, 1), slice(1, 2), slice(2, 4))]
    assert rel_l2(torch.cat(outs, dim=1), g["out"][0]) < 1e-5
    assert rel_l2(g["out"], g["full"]) < 1e-5          # the cache protocol is exact causal streaming
    vae.clear_cache()
    assert rel_l2(vae.causal_conv(x, p), g["full"][0]) < 1e-5


def test_g9_resblock_attn(golden, vae):
    g = golden("vae_g9_resblock")
    x = torch.from_numpy(g["x"])[0]
    vae.clear_cache()
    outs = [vae.residual_block(x[:, sl], "decoder.upsamples.4") for sl in (slice(0, 1), slice(1, 3))]
    assert rel_l2(torch.cat(outs, dim=1), g["out"][0]) < 1e-5
    g = golden("vae_g9_attn")
    assert rel_l2(vae.attention_block(torch.from_numpy(g["x"])[0], "decoder.middle.1"), g["out"][0]) < 1e-5


def test_g9_resample_modes(golden, vae):
    g = golden("vae_g9_up3d")
    x = torch.from_numpy(g["x"])[0]
    vae.clear_cache()
    outs = [vae.resample(x[:, i:i + 1], "decoder.upsamples.3", "upsample3d") for i in range(3)]
    assert [o.shape[1] for o in outs] == [1, 2, 2]        # first chunk is not doubled in time ('Rep')
    assert rel_l2(torch.cat(outs, dim=1), g["out"][0]) < 1e-5
    g = golden("vae_g9_up2d")
    vae.clear_cache()
    assert rel_l2(vae.resample(torch.from_numpy(g["x"])[0], "decoder.upsamples.11", "upsample2d"), g["out"][0]) < 1e-5
    g = golden("vae_g9_down3d")
    x = torch.from_numpy(g["x"])[0]
    vae.clear_cache()
    outs = [vae.resample(x[:, sl], "encoder.downsamples.5", "downsample3d") for sl in (slice(0, 1), slice(1, 5))]
    assert [o.shape[1] for o in outs] == [1, 2]
    assert rel_l2(torch.cat(outs, dim=1), g["out"][0]) < 1e-5
    g = golden("vae_g9_down2d")
    vae.clear_cache()
    out = vae.resample(torch.from_numpy(g["x"])[0], "encoder.downsamples.2", "downsample2d")
    assert out.shape[-2:] == (3, 5) and rel_l2(out, g["out"][0]) < 1e-5


def test_g10_encode_decode(golden, vae):
    g = golden("vae_g10_encode")
    video = torch.from_numpy(g["video"])[0]
    params = vae.encode(video)
    assert params.shape == (32, 3, 4, 6)
    assert rel_l2(params, g["params"][0]) < 1e-5
    assert rel_l2(params[:16], g["mode"][0]) < 1e-5
    assert rel_l2(vae.encode(video[:, :1]), g["params_t1"][0]) < 1e-5
    g = golden("vae_g10_decode")
    z = torch.from_numpy(g["z"])[0]
    out = vae.decode(z)
    assert out.shape == (3, 9, 32, 48) and float(out.abs().max()) <= 1.0
    assert rel_l2(out, g["out"][0]) < 1e-5
    assert rel_l2(vae.decode(z[:, :1]), g["out_t1"][0]) < 1e-5
