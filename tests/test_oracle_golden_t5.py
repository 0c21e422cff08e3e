"""umT5 CPU oracle vs the fixture captured from the reference's WanT5EncoderModel
(oracle/gen_golden_t5.py).  fp32 vs fp32: rel-L2 <= 1e-5; bucket ids bit-exact."""
import pytest
import torch

from oracle import t5_oracle as T
from oracle.gen_golden_t5 import TINY
from videocof_amd.weights import deterministic_t5_state_dict, t5_param_shapes


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def enc():
    return T.T5EncoderOracle(deterministic_t5_state_dict(**TINY), TINY["num_heads"], TINY["num_layers"], TINY["num_buckets"])


def test_umt5_xxl_parameter_inventory():
    s = t5_param_shapes(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32)
    n = sum(int(torch.tensor(v).prod()) for v in s.values())
    assert len(s) == 2 + 24 * 10
    assert n == 256384 * 4096 + 4096 + 24 * (2 * 4096 + 4 * 4096 * 4096 + 3 * 4096 * 10240 + 32 * 64)    # 5.68 B


def test_relative_position_buckets_bit_exact(golden):
    g = golden("t5_g13_encoder")
    rel = torch.from_numpy(g["rel"])
    assert torch.equal(T.relative_position_bucket(rel, 32), torch.from_numpy(g["buckets"]))
    lut = T.bucket_lut(512, 32)
    assert lut.dtype == torch.int32 and lut.numel() == 1023
    assert torch.equal(lut.long(), torch.from_numpy(g["buckets"])[700 - 511: 700 + 512])
    assert int(lut[511]) == 0 and int(lut[512]) == 17 and int(lut[510]) == 1 and int(lut.max()) == 31


def test_position_bias_tensor(golden, enc):
    g = golden("t5_g13_encoder")
    assert rel_l2(enc.pos_bias(0, 20), g["bias20"]) < 1e-7


def test_single_block_with_key_mask(golden, enc):
    g = golden("t5_g13_encoder")
    y = enc.block(1, torch.from_numpy(g["blk_x"]), torch.from_numpy(g["blk_mask"]))
    assert rel_l2(y, g["blk_y"]) < 1e-5


def test_encoder_forward_padded_batch(golden, enc):
    g = golden("t5_g13_encoder")
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    out = enc.forward(ids, mask)
    assert out.shape == (3, 72, TINY["dim"])
    assert rel_l2(out, g["out"]) < 1e-5
    assert rel_l2(enc.forward(ids[:1], None), g["out_nomask"]) < 1e-5
    # the mask matters: unmasked result differs on the valid rows of sample 0
    assert rel_l2(out[0, :37], g["out_nomask"][0, :37]) > 1e-3
    # valid rows do not depend on what sits in the padded positions
    ids2 = ids.clone()
    ids2[1, 11:] = 5
    assert rel_l2(enc.forward(ids2, mask)[1, :11], out[1, :11]) < 1e-6
