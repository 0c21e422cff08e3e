"""The CPU oracle against fixtures captured from the reference itself
(tests/golden/*.npz, produced by oracle/gen_golden.py).  fp32 restatement vs
fp32 reference: rel-L2 <= 1e-5 (SURVEY.md section 8c tolerance guidance)."""
import math

import numpy as np
import pytest
import torch

from oracle import wan_oracle as O
from videocof_amd.weights import deterministic_dit_state_dict

TINY = dict(dim=256, ffn_dim=512, num_layers=2, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
CFG = O.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def sd():
    return deterministic_dit_state_dict(**TINY)


def test_g1_sinusoid(golden):
    g = golden("dit_g1_sinusoid")
    out = O.sinusoidal_embedding_1d(256, torch.from_numpy(g["t"]))
    assert out.dtype == torch.float64
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=1e-12)


def test_g2_rope_table(golden):
    g = golden("dit_g2_freqs")
    ang = O.rope_angles(128)
    assert tuple(ang.shape) == tuple(g["shape"]) == (1024, 64)
    assert O.rope_axis_dims(128) == (22, 21, 21)
    rows = g["rows"]
    np.testing.assert_allclose(ang[rows].cos().numpy(), g["real"], atol=1e-12)
    np.testing.assert_allclose(ang[rows].sin().numpy(), g["imag"], atol=1e-12)
    assert abs(float(ang.cos().sum()) - float(g["sum_real"])) < 1e-6
    assert abs(float(ang.sin().sum()) - float(g["sum_imag"])) < 1e-6


@pytest.mark.parametrize("mode,fs,gr", [("default", None, None), ("paired", 3, None),
                                        ("cof", 3, (3, 4)), ("cof_g2", 2, (2, 4))])
def test_g3_rope_apply(golden, mode, fs, gr):
    g = golden("dit_g3_rope")
    x = torch.from_numpy(g["x"])[0]
    out = O.rope_apply(x, tuple(g["grid"][0]), O.rope_angles(128), fs, gr)
    assert rel_l2(out, g[mode][0]) < 1e-6
    # pad rows pass through untouched
    assert torch.equal(out[-4:], x[-4:])


def test_temporal_positions():
    assert O.temporal_positions(7, None, None) == [0, 1, 2, 3, 4, 5, 6]
    assert O.temporal_positions(7, 3, None) == [0, 1, 2, 0, 1, 2, 3]
    assert O.temporal_positions(7, 3, (3, 4)) == [1, 2, 3, 0, 1, 2, 3]
    assert O.temporal_positions(43, 21, (21, 22)) == list(range(1, 22)) + [0] + list(range(1, 22))


def test_g4_norms_head_unpatchify(golden, sd):
    g = golden("dit_g4_norms")
    x = torch.from_numpy(g["x"])[0]
    e = sd["blocks.0.modulation"][0] + torch.from_numpy(g["e6"])[0]
    assert rel_l2(O.rms_norm(x, sd["blocks.0.self_attn.norm_q.weight"], 1e-6), g["rms_q"][0]) < 1e-6
    assert rel_l2(O.ln_modulate(x, e[1], e[0], 1e-6), g["ln_mod"][0]) < 1e-6
    assert rel_l2(O.layer_norm(x, 1e-6, sd["blocks.0.norm3.weight"], sd["blocks.0.norm3.bias"]),
                  g["ln_affine"][0]) < 1e-6
    assert rel_l2(O.head_forward(x, torch.from_numpy(g["ehead"])[0], sd, CFG), g["head"][0]) < 1e-5
    u = torch.from_numpy(g["u"])[0]
    assert torch.equal(O.unpatchify(u, (7, 3, 5), CFG), torch.from_numpy(g["unpatch"]))


def test_patchify_is_conv3d(sd):
    x = torch.randn(16, 3, 8, 12)
    tok, grid = O.patchify(x, CFG)
    ref = torch.nn.functional.conv3d(x[None], sd["patch_embedding.weight"], sd["patch_embedding.bias"],
                                     stride=(1, 2, 2)).flatten(2).transpose(1, 2)[0]
    mine = tok @ sd["patch_embedding.weight"].reshape(256, -1).t() + sd["patch_embedding.bias"]
    assert grid == (3, 4, 6)
    assert rel_l2(mine, ref) < 1e-5


def test_g5_block(golden, sd):
    g = golden("dit_g5_block")
    grid = tuple(int(v) for v in g["grid"])
    out = O.block_forward(torch.from_numpy(g["x"])[0], torch.from_numpy(g["e"])[0],
                          torch.from_numpy(g["ctx"])[0], sd, 0, CFG, grid, O.rope_angles(128),
                          3, (3, 4), math.prod(grid))
    assert rel_l2(out, g["out"][0]) < 1e-5


def test_g6_forward(golden, sd):
    g = golden("dit_g6_forward")
    lat = torch.from_numpy(g["lat"])
    ctx = [torch.from_numpy(g["ctx"])]
    out = O.dit_forward(sd, CFG, lat, torch.tensor([899]), ctx, 420, [3], [(3, 4)])
    assert rel_l2(out, g["out_cof"]) < 1e-5
    out = O.dit_forward(sd, CFG, lat, torch.tensor([499]), ctx, 420)
    assert rel_l2(out, g["out_t2v"]) < 1e-5
    out = O.dit_forward(sd, CFG, torch.from_numpy(g["lat2"]), torch.tensor([749, 749]),
                        [ctx[0], torch.from_numpy(g["ctx2"])], 420, [3, 3], [(3, 4), (3, 4)])
    assert rel_l2(out, g["out_b2"]) < 1e-5


def test_g7_unipc(golden):
    g = golden("dit_g7_unipc")
    s = O.UniPCOracle()
    s.set_timesteps(4, 3.0)
    assert s.timesteps.tolist() == g["timesteps"].tolist() == [999, 899, 749, 499]
    np.testing.assert_array_equal(s.sigmas.numpy(), g["sigmas"])
    cur = torch.from_numpy(g["x"])
    orders = []
    for i in range(4):
        cur = s.step(torch.from_numpy(g["v"][i]), cur)
        orders.append(s.this_order)
        assert rel_l2(cur, g["traj"][i]) < 2e-6, i
    assert orders == g["orders"].tolist() == [1, 2, 2, 1]
    g50 = golden("dit_g7_sched50")
    s.set_timesteps(50, 5.0)
    assert s.timesteps.tolist() == g50["timesteps"].tolist()
    np.testing.assert_array_equal(s.sigmas.numpy(), g50["sigmas"])


def test_g7b_unipc_12_steps(golden):
    """Steady second-order steps + lower_order_final (orders 1, 2 x 10, 1), shift 5 as in inference.py."""
    g = golden("dit_g7b_unipc12")
    s = O.UniPCOracle()
    s.set_timesteps(12, 5.0)
    assert s.timesteps.tolist() == g["timesteps"].tolist()
    np.testing.assert_array_equal(s.sigmas.numpy(), g["sigmas"])
    cur, orders = torch.from_numpy(g["x"]), []
    for i in range(12):
        cur = s.step(torch.from_numpy(g["v"][i]), cur)
        orders.append(s.this_order)
        assert rel_l2(cur, g["traj"][i]) < 5e-6, i
    assert orders == g["orders"].tolist() == [1] + [2] * 10 + [1]


def test_g8_cof_loop(golden, sd):
    g = golden("dit_g8_cof_loop")
    assert O.cof_layout(9, 4) == (3, 1)
    assert O.cof_layout(81, 4) == (21, 1)
    steps = O.cof_denoise(sd, CFG, torch.from_numpy(g["src"]), torch.from_numpy(g["noise"]),
                          [torch.from_numpy(g["ctx"])], 4, 3.0, 3, 1)
    for i in range(4):
        assert rel_l2(steps[i], g["steps"][i]) < 1e-5, i
    # source latents are algebraically fixed (parity checklist 11)
    assert (steps[-1][:, :, :3] - torch.from_numpy(g["src"])).abs().max() < 1e-5


def test_g8b_cfg_loop(golden, sd):
    g = golden("dit_g8_cof_loop")
    gb = golden("dit_g8b_cfg_loop")
    steps = O.cof_denoise(sd, CFG, torch.from_numpy(g["src"]), torch.from_numpy(g["noise"]),
                          [torch.from_numpy(g["ctx"])], 3, 5.0, 3, 1, guidance_scale=5.0,
                          negative_context=[torch.from_numpy(gb["neg"])])
    for i in range(3):
        assert rel_l2(steps[i], gb["steps"][i]) < 2e-5, i


def test_g11_sp_rope_slice(golden):
    """Rank-sliced RoPE of videox_fun/dist/wan_xfuser.py:22-63 == rows of the full-sequence map."""
    g = golden("dit_g11_sp_rope")
    x = torch.from_numpy(g["x"])[0]
    grid = tuple(g["grid"][0])
    Lr = x.shape[0]
    for r in range(2):
        out = O.rope_apply(x, grid, O.rope_angles(128), token_offset=r * Lr)
        assert rel_l2(out, g[f"rank{r}"][0]) < 1e-6


def test_g14_teacache_sequence(golden, sd):
    """TeaCache: the oracle's restatement against an 8-step sequence captured from the reference model with enable_teacache
    (decisions, accumulated distances via the decisions, outputs), num_skip_start_steps = 1 and 3."""
    g = golden("dit_g14_teacache")
    lat0, dl, ctx = torch.from_numpy(g["lat0"]), torch.from_numpy(g["dlat"]), [torch.from_numpy(g["ctx"])]
    for key, skip in (("", 1), ("_skip3", 3)):
        tc = O.TeaCacheOracle(g["coeff"], len(g["ts"]), float(g["thresh"]), skip)
        for i, t in enumerate(g["ts"]):
            out = O.dit_forward(sd, CFG, lat0 + i * dl, torch.tensor([int(t)]), ctx, 420, [3], [(3, 4)], teacache=tc)
            assert rel_l2(out, g["out" + key][i]) < 1e-5, (key, i)
        assert tc.decisions == g["calc" + key].tolist()
        assert tc.cnt == 0 and tc.prev_e0 is None           # reset after num_steps forwards
