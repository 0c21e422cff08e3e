"""merge_lora on reference-format state dicts vs the reference's own merge_lora (fixture g12,
lora_utils.py:371-500): three key styles, alpha scaling, text-encoder entries ignored."""
import numpy as np
import torch

from videocof_amd.lora_utils import merge_lora_state_dict, unmerge_lora_state_dict
from videocof_amd.weights import deterministic_dit_state_dict, det_uniform

TINY = dict(dim=256, ffn_dim=512, num_layers=2, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)


def _lora():
    r, C = 4, 256
    return {
        "diffusion_model.blocks.0.self_attn.q.lora_down.weight": det_uniform("l.a.down", (r, C), 0.3),
        "diffusion_model.blocks.0.self_attn.q.lora_up.weight": det_uniform("l.a.up", (C, r), 0.3),
        "diffusion_model.blocks.0.self_attn.q.alpha": torch.tensor(2.0),
        "blocks.1.ffn.0.lora_A.default.weight": det_uniform("l.b.down", (r, C), 0.3),
        "blocks.1.ffn.0.lora_B.default.weight": det_uniform("l.b.up", (512, r), 0.3),
        "lora_unet__blocks_1_cross_attn_o.lora_down.weight": det_uniform("l.c.down", (r, C), 0.3),
        "lora_unet__blocks_1_cross_attn_o.lora_up.weight": det_uniform("l.c.up", (C, r), 0.3),
        "lora_unet__blocks_1_cross_attn_o.alpha": torch.tensor(8.0),
        "lora_te_text_model_encoder_layers_0_mlp_fc1.lora_down.weight": torch.zeros(r, 8),
    }


def test_merge_matches_reference(golden):
    g = golden("dit_g12_lora")
    sd = deterministic_dit_state_dict(**TINY)
    orig = {k: v.clone() for k, v in sd.items()}
    n = merge_lora_state_dict(sd, _lora(), float(g["multiplier"]))
    assert n == 3
    np.testing.assert_allclose(sd["blocks.0.self_attn.q.weight"].numpy(), g["q"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(sd["blocks.1.ffn.0.weight"].numpy(), g["ffn0"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(sd["blocks.1.cross_attn.o.weight"].numpy(), g["o"], rtol=0, atol=1e-6)
    assert torch.equal(sd["blocks.0.self_attn.k.weight"], torch.from_numpy(g["untouched"]))
    assert not torch.equal(sd["blocks.0.self_attn.q.weight"], orig["blocks.0.self_attn.q.weight"])
    unmerge_lora_state_dict(sd, _lora(), float(g["multiplier"]))
    for k in ("blocks.0.self_attn.q.weight", "blocks.1.ffn.0.weight", "blocks.1.cross_attn.o.weight"):
        assert float((sd[k] - orig[k]).abs().max()) < 1e-6


def test_merge_keeps_dtype_and_rejects_shape_mismatch():
    sd = {k: v.bfloat16() for k, v in deterministic_dit_state_dict(**TINY).items()}
    merge_lora_state_dict(sd, _lora(), 1.0)
    assert sd["blocks.0.self_attn.q.weight"].dtype == torch.bfloat16
    bad = {"diffusion_model.blocks.0.self_attn.q.lora_down.weight": torch.zeros(4, 100),
           "diffusion_model.blocks.0.self_attn.q.lora_up.weight": torch.zeros(256, 4)}
    import pytest
    with pytest.raises(ValueError, match="does not match"):
        merge_lora_state_dict(sd, bad, 1.0)
    # a bare dotted name is cut at its FIRST dot by the reference (lora_utils.py:394): layer "blocks", no pair -> dropped, there and here
    before = sd["blocks.0.self_attn.q.weight"].clone()
    bare = {"blocks.0.self_attn.q.lora_down.weight": torch.ones(4, 256), "blocks.0.self_attn.q.lora_up.weight": torch.ones(256, 4)}
    assert merge_lora_state_dict(sd, bare, 1.0) == 0 and torch.equal(sd["blocks.0.self_attn.q.weight"], before)
