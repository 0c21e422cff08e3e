"""-m gpu: every HIP kernel, called through the C ABI (videocof_amd.ops -> libwan_hip.so),
against the CPU oracle on seeded inputs and against the reference-captured fixtures.

Tolerances (bf16 kernels vs the fp32 oracle): a single bf16 rounding is
2^-9 relative, so element-wise ops must land within rel-L2 4e-3 of the oracle evaluated on the
same bf16-rounded inputs; fp32-output epilogues within 1e-5; attention within 6e-3.
"""
import math

import numpy as np
import pytest
import torch

from oracle import wan_oracle as O
from videocof_amd import _lib, ops
from videocof_amd._lib import RopeParams
from videocof_amd.attention_utils import attention

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def bf(x):
    return x.to(torch.bfloat16)


@pytest.fixture(scope="module")
def rope_dev():
    ang = O.rope_angles(128)
    return ang.cos().float().contiguous().to(DEV), ang.sin().float().contiguous().to(DEV)


def test_extension_is_loaded():
    lib = _lib.load()
    assert lib.wan_abi_version() == _lib.ABI_VERSION            # (the number itself is pinned by tests/test_abi.py)
    maps = open("/proc/self/maps").read()
    assert "libwan_hip.so" in maps


@pytest.mark.parametrize("dim", [256, 1536, 5120])
def test_ln_modulate_vs_oracle(dim):
    g = torch.Generator().manual_seed(dim)
    x = torch.randn(75, dim, generator=g) * 2 + 0.3
    sc, sh = torch.randn(3, dim, generator=g) * 0.5, torch.randn(3, dim, generator=g) * 0.5
    out = ops.ln_modulate(x.to(DEV), sc.to(DEV), sh.to(DEV), True, 25, 1e-6)
    ref = torch.cat([O.ln_modulate(x[b * 25:(b + 1) * 25], sc[b], sh[b], 1e-6) for b in range(3)])
    assert out.dtype == torch.bfloat16 and rel_l2(out, ref) < 4e-3
    out = ops.ln_modulate(x.to(DEV), sc[:1].to(DEV), sh[:1].to(DEV), False, 75, 1e-6)
    assert rel_l2(out, O.layer_norm(x, 1e-6, sc[0], sh[0])) < 4e-3


def test_g4_norm_fixtures(golden):
    """norm1+modulate, norm3 (affine), WanRMSNorm exactly as the reference computed them."""
    from videocof_amd.weights import deterministic_dit_state_dict
    g = golden("dit_g4_norms")
    sd = deterministic_dit_state_dict(dim=256, ffn_dim=512, num_layers=2, text_dim=64)
    x = torch.from_numpy(g["x"])[0]
    e = sd["blocks.0.modulation"][0] + torch.from_numpy(g["e6"])[0]
    out = ops.ln_modulate(x.to(DEV), e[1:2].contiguous().to(DEV), e[0:1].contiguous().to(DEV), True, 37, 1e-6)
    assert rel_l2(out, g["ln_mod"][0]) < 4e-3
    out = ops.ln_modulate(x.to(DEV), sd["blocks.0.norm3.weight"][None].to(DEV),
                          sd["blocks.0.norm3.bias"][None].to(DEV), False, 37, 1e-6)
    assert rel_l2(out, g["ln_affine"][0]) < 4e-3
    xb = bf(x).to(DEV).clone()
    ops.rmsnorm_rope_(xb, sd["blocks.0.self_attn.norm_q.weight"].to(DEV), None, None, 128, 1e-6)
    assert rel_l2(xb, O.rms_norm(bf(x).float(), sd["blocks.0.self_attn.norm_q.weight"], 1e-6)) < 4e-3
    assert rel_l2(xb, g["rms_q"][0]) < 8e-3          # vs reference on un-rounded input


@pytest.mark.parametrize("mode,fs,gr", [(0, None, None), (1, 3, None), (2, 3, (3, 4)), (2, 2, (2, 4))])
def test_g3_rope_fixture(golden, rope_dev, mode, fs, gr):
    """RoPE alone (unit RMS weight is not an identity, so compare against oracle = rms(1) then rope,
    and pin the oracle itself to the reference fixture in test_oracle_golden)."""
    g = golden("dit_g3_rope")
    x = torch.from_numpy(g["x"])[0]                      # [116, 2, 128], last 4 rows are padding
    rows, C = x.shape[0], 256
    xb = bf(x).reshape(rows, C)
    ones = torch.ones(C)
    q = xb.to(DEV).clone()
    k = (xb * 0.5).to(DEV).clone()
    rp = RopeParams(7, 4, 4, mode, fs or 0, gr[1] if gr else 0, 0, rows, 1024)
    ops.rmsnorm_rope_(q, ones.to(DEV), k, (ones * 2).to(DEV), 128, 1e-6, rope_dev, rp)
    ang = O.rope_angles(128)
    refq = O.rope_apply(O.rms_norm(xb.float(), ones, 1e-6).view(rows, 2, 128), (7, 4, 4), ang, fs, gr)
    refk = O.rope_apply(O.rms_norm(xb.float() * 0.5, ones * 2, 1e-6).view(rows, 2, 128), (7, 4, 4), ang, fs, gr)
    assert rel_l2(q, refq.reshape(rows, C)) < 4e-3
    assert rel_l2(k, refk.reshape(rows, C)) < 4e-3


def test_rope_token_offset_is_a_slice_of_the_full_map(rope_dev):
    """Sequence-parallel shard r == rows [r*Ll, (r+1)*Ll) of the single-device result (CoF mode)."""
    g = torch.Generator().manual_seed(5)
    F, Hp, Wp, C = 7, 6, 10, 384
    L = F * Hp * Wp
    x = bf(torch.randn(L, C, generator=g))
    w = (torch.rand(C, generator=g) + 0.5)
    full = x.to(DEV).clone()
    ops.rmsnorm_rope_(full, w.to(DEV), None, None, 128, 1e-6, rope_dev, RopeParams(F, Hp, Wp, 2, 3, 4, 0, L, 1024))
    Ll = L // 4
    for r in range(4):
        part = x[r * Ll:(r + 1) * Ll].to(DEV).clone()
        ops.rmsnorm_rope_(part, w.to(DEV), None, None, 128, 1e-6, rope_dev,
                          RopeParams(F, Hp, Wp, 2, 3, 4, r * Ll, Ll, 1024))
        assert torch.equal(part, full[r * Ll:(r + 1) * Ll])


@pytest.mark.parametrize("mode,key,fs,gr", [(0, "default", None, None), (2, "cof", 3, (3, 4))])
def test_g3_rope_fixture_direct(golden, rope_dev, mode, key, fs, gr):
    """The reference-captured rope_apply output itself (not the oracle): RoPE is linear in the row and WanRMSNorm with a
    unit gain scales the row by 1/rms, so kernel(x, w = 1) == fixture(x) / rms(x)."""
    g = golden("dit_g3_rope")
    x = torch.from_numpy(g["x"])[0].reshape(116, 256)
    xb = bf(x)
    q = xb.to(DEV).clone()
    ops.rmsnorm_rope_(q, torch.ones(256, device=DEV), None, None, 128, 1e-6, rope_dev,
                      RopeParams(7, 4, 4, mode, fs or 0, gr[1] if gr else 0, 0, 116, 1024))
    rms = torch.sqrt(xb.float().pow(2).mean(dim=1, keepdim=True) + 1e-6)
    ref = torch.from_numpy(g[key])[0].reshape(116, 256).float() / rms
    assert rel_l2(q, ref) < 5e-3          # x itself is bf16-rounded on the way in (2^-9) + the bf16 output rounding


def test_g11_sp_rope_fixture(golden, rope_dev):
    """The reference's sequence-parallel rope_apply (dist/wan_xfuser.py:22-63), captured with rank r of 2: the kernel with
    token_offset = r * L/2 on the local rows must reproduce it (same 1/rms argument as above)."""
    g = golden("dit_g11_sp_rope")
    x = torch.from_numpy(g["x"])[0].reshape(56, 256)
    xb = bf(x)
    rms = torch.sqrt(xb.float().pow(2).mean(dim=1, keepdim=True) + 1e-6)
    outs = []
    for r in range(2):
        q = xb.to(DEV).clone()
        ops.rmsnorm_rope_(q, torch.ones(256, device=DEV), None, None, 128, 1e-6, rope_dev,
                          RopeParams(7, 4, 4, 0, 0, 0, r * 56, 56, 1024))
        ref = torch.from_numpy(g[f"rank{r}"])[0].reshape(56, 256).float() / rms
        assert rel_l2(q, ref) < 5e-3, r
        outs.append(q)
    assert not torch.equal(outs[0], outs[1])       # the two shards really sit at different positions


@pytest.mark.parametrize("M,N,K", [(300, 384, 256), (1, 128, 64), (515, 64, 1024), (129, 1536, 192)])
def test_gemm_epilogues_vs_oracle(M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a, w = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) * 0.1)
    bias = torch.randn(N, generator=g) * 0.5
    acc = a.double() @ w.double().t() + bias.double()
    ad, wd, bd = a.to(DEV), w.to(DEV), bias.to(DEV)
    assert rel_l2(ops.gemm(ad, wd, bd, ops.EPI_BF16), acc) < 4e-3
    assert rel_l2(ops.gemm(ad, wd, bd, ops.EPI_GELU_BF16), O.gelu_tanh(acc)) < 4e-3
    assert rel_l2(ops.gemm(ad, wd, bd, ops.EPI_F32), acc) < 1e-5
    assert rel_l2(ops.gemm(ad, wd, None, ops.EPI_F32), acc - bias.double()) < 1e-5
    x0 = torch.randn(M, N, generator=g)
    gate = torch.randn(2, N, generator=g)
    rpb = (M + 1) // 2
    x = x0.to(DEV).clone()
    ops.gemm(ad, wd, bd, ops.EPI_RESID_F32, out=x, gate=gate.to(DEV), rows_per_batch=rpb)
    gsel = gate[(torch.arange(M) // rpb)]
    assert rel_l2(x, x0.double() + acc * gsel.double()) < 1e-5
    x = x0.to(DEV).clone()
    ops.gemm(ad, wd, bd, ops.EPI_RESID_F32, out=x)
    assert rel_l2(x, x0.double() + acc) < 1e-5
    vt = ops.gemm(ad, wd, bd, ops.EPI_BF16_T)
    assert vt.shape == (N, ops.round_up(M, 64))
    assert rel_l2(vt[:, :M].t(), acc) < 4e-3
    assert float(vt[:, M:].abs().max()) == 0.0 if vt.shape[1] > M else True
    # strided A (a view into a wider buffer), as the q|k buffer is used
    wide = torch.zeros(M, K + 64, dtype=torch.bfloat16, device=DEV)
    wide[:, 64:] = ad
    assert rel_l2(ops.gemm(wide[:, 64:], wd, bd, ops.EPI_F32), acc) < 1e-5


def test_gemm_argument_errors():
    a = torch.zeros(8, 100, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(16, 100, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.gemm(a, w, None, ops.EPI_BF16)
    with pytest.raises(ValueError, match="a is"):
        ops.gemm(a, torch.zeros(16, 64, dtype=torch.bfloat16, device=DEV), None, ops.EPI_BF16)
    with pytest.raises(ValueError, match="expected torch.bfloat16"):
        ops.gemm(a.float(), w, None, ops.EPI_BF16)


def _attn_ref(q, k, v, k_len=None):
    return torch.stack([O.attention(q[b].float(), k[b].float(), v[b].float(), k_len) for b in range(q.shape[0])])


@pytest.mark.parametrize("B,Lq,Lk,H,qs", [(1, 300, 420, 2, 1.0), (2, 64, 64, 1, 1.0), (1, 257, 8, 3, 1.0),
                                          (1, 520, 512, 2, 3.0), (1, 33, 1000, 1, 6.0), (2, 420, 420, 2, 2.0)])
def test_attention_seam_vs_oracle(B, Lq, Lk, H, qs):
    """attention() with the reference's [B,L,N,D] layout (attention_utils.py:152-168)."""
    g = torch.Generator().manual_seed(Lq * 7 + Lk)
    q = bf(torch.randn(B, Lq, H, 128, generator=g) * qs)
    k = bf(torch.randn(B, Lk, H, 128, generator=g))
    v = bf(torch.randn(B, Lk, H, 128, generator=g) + torch.arange(128) * 0.01)     # asymmetric in d
    out = attention(q.to(DEV), k.to(DEV), v.to(DEV))
    assert out.shape == q.shape and out.dtype == torch.bfloat16
    ref = _attn_ref(q, k, v)
    assert rel_l2(out, ref) < 6e-3
    assert float((out.float().cpu() - ref).abs().max()) < 4e-2


def test_cross_attention_persistent_form_walks_many_blocks():
    """Short key streams (cross-attention: 512 text rows = 8 KV tiles) run on the PERSISTENT form of the 4-wave kernel as soon as a
    launch has more query blocks than the chip has CUs: one resident workgroup per CU walks blocks w, w + grid, ... with the next
    block's K / V requests and Q fragments issued before the current block's output stores (csrc/attn_fwd.hip, PERSIST).  Checked:
    against the oracle (ragged last block: Lq % 256 != 0; B = 2; heads pinned to XCDs), bit for bit against the one-workgroup-per-block
    launch of the same kernel family (tuning key attn_persist = 0), with a key count that ends inside the peeled last tile, plain
    (not pre-scaled) q -- the packed-shift form -- and ragged per-sample key counts from device memory."""
    g = torch.Generator().manual_seed(11)
    B, Lq, Lk, H = 2, 256 * 20 + 37, 512, 8           # 21 query blocks x 16 (batch, head) pairs = 336 workgroups' worth of blocks
    q = bf(torch.randn(B, Lq, H, 128, generator=g) * 2.0)
    k = bf(torch.randn(B, Lk, H, 128, generator=g))
    v = bf(torch.randn(B, Lk, H, 128, generator=g) + torch.arange(128) * 0.01)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    assert ops.get_tuning("attn_persist") == 1
    out = attention(qd, kd, vd)
    ref = _attn_ref(q, k, v)
    assert rel_l2(out, ref) < 6e-3 and float((out.float().cpu() - ref).abs().max()) < 4e-2
    again = attention(qd, kd, vd)
    assert torch.equal(out, again)                      # run to run (a block that started before its requests landed would differ)
    out300 = attention(qd, kd, vd, k_lens=torch.tensor([300, 300]))
    ragged = attention(qd, kd, vd, k_lens=torch.tensor([300, 77], device=DEV))
    # pre-scaled q through the raw op (the form the DiT uses): [B, L, C] operands, V^T
    qs = (qd.float() * (128 ** -0.5) * 1.4426950408889634).bfloat16().reshape(B, Lq, H * 128)
    vt = torch.zeros(B, H * 128, Lk, device=DEV, dtype=torch.bfloat16)
    for b in range(B):
        ops.transpose_pad(vd[b].reshape(Lk, H * 128), Lk, out=vt[b])
    pre = ops.attention_fwd(qs, kd.reshape(B, Lk, H * 128), vt, H, q_prescaled=True)
    ops.set_tuning("attn_persist", 0)
    try:
        assert torch.equal(attention(qd, kd, vd), out), "persistent form != one workgroup per block"
        assert torch.equal(attention(qd, kd, vd, k_lens=torch.tensor([300, 300])), out300)
        assert torch.equal(attention(qd, kd, vd, k_lens=torch.tensor([300, 77], device=DEV)), ragged)
        assert torch.equal(ops.attention_fwd(qs, kd.reshape(B, Lk, H * 128), vt, H, q_prescaled=True), pre)
    finally:
        ops.set_tuning("attn_persist", 1)
    assert rel_l2(out300, _attn_ref(q, k, v, 300)) < 6e-3
    assert rel_l2(ragged[1:], _attn_ref(q[1:], k[1:], v[1:], 77)) < 6e-3
    assert rel_l2(pre.view(B, Lq, H, 128), ref) < 8e-3


def test_attention_k_lens_masks_keys_and_q_lens_zero_rows():
    g = torch.Generator().manual_seed(3)
    q, k, v = (bf(torch.randn(2, 200, 2, 128, generator=g)) for _ in range(3))
    out = attention(q.to(DEV), k.to(DEV), v.to(DEV), k_lens=torch.tensor([130, 130]), q_lens=torch.tensor([200, 150]))
    ref = _attn_ref(q, k, v, 130)
    ref[1, 150:] = 0
    assert rel_l2(out, ref) < 6e-3
    assert float(out[1, 150:].abs().max()) == 0.0
    out = attention(q.to(DEV), k.to(DEV), v.to(DEV), k_lens=torch.tensor([130, 77]))       # ragged batch
    assert rel_l2(out[1:], _attn_ref(q[1:], k[1:], v[1:], 77)) < 6e-3
    out32 = attention(q.float().to(DEV), k.float().to(DEV), v.float().to(DEV))            # fp32 in -> fp32 out
    assert out32.dtype == torch.float32 and rel_l2(out32, _attn_ref(q, k, v)) < 6e-3


def test_attention_ragged_batch_runs_as_one_launch():
    """flash_attention() with ragged k_lens (attention_utils.py:95-146 packs the samples behind cu_seqlens): here ONE
    wan_attention_fwd_varlen launch whose workgroups read their sample's key count from device memory.  Same bits as one call per
    sample on the cut tensors; NaN in the padding of k and v never enters; a sample without keys gives zero rows."""
    g = torch.Generator().manual_seed(5)
    B, Lq, Lk, H = 3, 300, 1500, 2
    kl = [1500, 777, 64]
    q = bf(torch.randn(B, Lq, H, 128, generator=g))
    k = bf(torch.randn(B, Lk, H, 128, generator=g))
    v = bf(torch.randn(B, Lk, H, 128, generator=g) + torch.arange(128) * 0.01)
    kp, vp = k.clone(), v.clone()
    for b in range(B):
        kp[b, kl[b]:] = float("nan")
        vp[b, kl[b]:] = float("nan")
    out = attention(q.to(DEV), kp.to(DEV), vp.to(DEV), k_lens=torch.tensor(kl, device=DEV))        # lengths stay on the device
    assert ops.get_tuning("last_attn_variant") & _lib.ATTN_VARIANT_SPLIT_TAIL == 0
    assert torch.isfinite(out.float()).all()
    for b in range(B):
        one = attention(q[b:b + 1].to(DEV), k[b:b + 1, :kl[b]].to(DEV), v[b:b + 1, :kl[b]].to(DEV))
        assert torch.equal(out[b:b + 1], one), b
        assert rel_l2(out[b:b + 1], _attn_ref(q[b:b + 1], k[b:b + 1], v[b:b + 1], kl[b])) < 6e-3
    out_l = attention(q.to(DEV), kp.to(DEV), vp.to(DEV), k_lens=kl, q_lens=[300, 10, 300])           # host list: same launch
    assert torch.equal(out_l[0], out[0]) and torch.equal(out_l[2], out[2]) and torch.equal(out_l[1, :10], out[1, :10])
    assert float(out_l[1, 10:].abs().max()) == 0.0
    out0 = attention(q[:2].to(DEV), kp[:2].to(DEV), vp[:2].to(DEV), k_lens=torch.tensor([777, 0], device=DEV))
    assert float(out0[1].abs().max()) == 0.0
    assert rel_l2(out0[:1], _attn_ref(q[:1], k[:1], v[:1], 777)) < 6e-3


def test_attention_ragged_batch_long_launch_skips_the_split_tail():
    """A shape whose uniform launch takes the split-KV tail round (5 heads x 53 query blocks = 265 workgroups on 256 CUs): the
    ragged form walks the per-sample key count in every workgroup instead, and agrees with the uniform call within tolerance."""
    g = torch.Generator().manual_seed(6)
    L, H, kv = 13568, 5, 10000
    q = bf(torch.randn(1, L, H * 128, generator=g)).to(DEV)
    k = bf(torch.randn(1, L, H * 128, generator=g)).to(DEV)
    v = bf(torch.randn(1, L, H * 128, generator=g)).to(DEV)
    vt = torch.stack([ops.transpose_pad(v[0, :kv], ops.round_up(L, 64))])
    uni = ops.attention_fwd(q, k, vt, H, k_len=kv)
    tail = ops.get_tuning("last_attn_variant") & _lib.ATTN_VARIANT_SPLIT_TAIL
    rag = ops.attention_fwd(q, k, vt, H, k_lens=torch.tensor([kv], device=DEV, dtype=torch.int32))
    assert ops.get_tuning("last_attn_variant") & _lib.ATTN_VARIANT_SPLIT_TAIL == 0
    assert tail != 0, "the uniform launch of this shape is expected to take the tail round (256-CU part)"
    assert rel_l2(rag, uni) < 2e-3
    with pytest.raises(ValueError):
        ops.attention_fwd(q, k, vt, H, k_lens=torch.tensor([kv, kv], device=DEV, dtype=torch.int32))


def test_attention_online_softmax_rescale_branch():
    """A key tile whose scores jump far above the running max forces the rescale path
    (guide rule 26): spike one key against every query late in the sequence."""
    g = torch.Generator().manual_seed(11)
    q = bf(torch.randn(1, 96, 1, 128, generator=g))
    k = bf(torch.randn(1, 640, 1, 128, generator=g))
    v = bf(torch.randn(1, 640, 1, 128, generator=g))
    k[0, 500, 0] = bf(q[0, :, 0].float().mean(0) * 40)         # huge score at tile 7
    k[0, 70, 0] = bf(q[0, 5, 0].float() * 8)
    out = attention(q.to(DEV), k.to(DEV), v.to(DEV))
    ref = _attn_ref(q, k, v)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 6e-3


@pytest.mark.parametrize("Lq,Lk,H,qs,spike", [(300, 420, 2, 1.0, False), (257, 8, 1, 1.0, False), (96, 640, 1, 1.0, True),
                                              (33, 1000, 1, 6.0, False), (520, 577, 2, 3.0, True)])
def test_attention_prescaled_q_path(Lq, Lk, H, qs, spike):
    """WAN_ATTN_Q_PRESCALED: q carries softmax_scale*log2(e) and the running max lives in the MFMA
    accumulator.  Same function as the plain path: compare with the oracle on the un-scaled q, incl. the
    rescale branch (late spike), a first tile far BELOW zero (scores ~ -60) and a ragged last tile."""
    g = torch.Generator().manual_seed(Lq + 3 * Lk)
    q = torch.randn(1, Lq, H, 128, generator=g) * qs
    k = bf(torch.randn(1, Lk, H, 128, generator=g))
    v = bf(torch.randn(1, Lk, H, 128, generator=g) + torch.arange(128) * 0.01)
    if spike:
        k[0, Lk - 140, 0] = bf(q[0, :, 0].mean(0) * 40)                 # huge score late in the stream
        k[0, :64] = bf(-q[0, :64].mean(0, keepdim=True).expand(64, H, 128) * 30)    # tile 0: very negative scores
    c = ops.q_prescale(128)
    q_pre = bf(q * c)                                                   # what wan_rmsnorm_rope(x0_scale=c) stores
    q_eff = q_pre.float() / c                                           # the q the kernel effectively sees
    C = H * 128
    out = ops.attention_fwd(q_pre.view(1, Lq, C).to(DEV), k.view(1, Lk, C).to(DEV),
                            ops.transpose_pad(v.view(Lk, C).to(DEV))[None], H, q_prescaled=True)
    ref = _attn_ref(q_eff, k, v).view(1, Lq, C)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 6e-3
    plain = ops.attention_fwd(bf(q).view(1, Lq, C).to(DEV), k.view(1, Lk, C).to(DEV),
                              ops.transpose_pad(v.view(Lk, C).to(DEV))[None], H)
    assert rel_l2(plain, _attn_ref(bf(q), k, v).view(1, Lq, C)) < 6e-3
    with pytest.raises(ValueError, match="unknown flags"):
        from videocof_amd import _lib
        _lib.check(_lib.load().wan_attention_fwd(1, C, 0, 1, C, 0, 1, 64, 0, 1, C, 0, 1, 8, 8, H, 128, 0.1, 6, None, 0, None),
                   "wan_attention_fwd")


def test_rmsnorm_x0_scale_scales_only_x0(rope_dev):
    g = torch.Generator().manual_seed(9)
    rows, C = 112, 256
    x = bf(torch.randn(rows, 2 * C, generator=g))
    w = torch.rand(C, generator=g) + 0.5
    rp = RopeParams(7, 4, 4, 2, 3, 4, 0, rows, 1024)
    a = x.to(DEV).clone(); b = x.to(DEV).clone()
    ops.rmsnorm_rope_(a[:, :C], w.to(DEV), a[:, C:], w.to(DEV), 128, 1e-6, rope_dev, rp)
    c = ops.q_prescale(128)
    ops.rmsnorm_rope_(b[:, :C], w.to(DEV), b[:, C:], w.to(DEV), 128, 1e-6, rope_dev, rp, x0_scale=c)
    assert torch.equal(a[:, C:], b[:, C:])                                      # k untouched
    ang = O.rope_angles(128)
    ref = O.rope_apply(O.rms_norm(x[:, :C].float(), w, 1e-6).view(rows, 2, 128), (7, 4, 4), ang, 3, (3, 4)).reshape(rows, C) * c
    assert rel_l2(b[:, :C], ref) < 4e-3
    assert rel_l2(b[:, :C].float().cpu() / c, a[:, :C].float().cpu()) < 4e-3
    with pytest.raises(ValueError, match="x0_scale"):
        ops.rmsnorm_rope_(b[:, :C], w.to(DEV), None, None, 128, 1e-6, x0_scale=0.0)


def test_attention_is_bitwise_reproducible():
    """Run-to-run equality on identical inputs, both q conventions.  (A hand-placed v_max3 on MFMA results
    once slipped past the hazard recogniser: results stayed within tolerance but differed run to run.)"""
    g = torch.Generator(device=DEV).manual_seed(4)
    for Lq, Lk, H in [(420, 420, 2), (840, 1000, 1), (4096, 8192, 4)]:
        C = H * 128
        q = torch.randn(1, Lq, C, device=DEV, generator=g).bfloat16()
        k = torch.randn(1, Lk, C, device=DEV, generator=g).bfloat16()
        vt = ops.transpose_pad(torch.randn(Lk, C, device=DEV, generator=g).bfloat16())[None]
        for pre in (False, True):
            outs = [ops.attention_fwd(q, k, vt, H, q_prescaled=pre).clone() for _ in range(4)]
            assert all(torch.equal(outs[0], o) for o in outs[1:]), (Lq, Lk, H, pre)


@pytest.mark.parametrize("pre", [False, True])
def test_attention_split_tail_round(pre):
    """Launches whose workgroup count leaves a small remainder over the CU count run their last query blocks
    split over the key range + a merge kernel (scratch from wan_attention_workspace_bytes).  Same function as the plain
    launch (tuning attn_tail = 0) and as the oracle, including the ragged last key tile inside the last split."""
    from videocof_amd import _lib
    Lq, Lk, H = 86 * 256 + 10, 1100, 3                  # 87 x 3 = 261 workgroups = 256 + 5
    C = H * 128
    assert _lib.load().wan_attention_workspace_bytes(1, Lq, Lk, H, 128) > 4096      # flags + tail partials
    assert _lib.load().wan_attention_workspace_bytes(1, 4096, Lk, H, 128) == 256    # fits one round: flags only
    g = torch.Generator(device=DEV).manual_seed(8)
    q = torch.randn(1, Lq, C, device=DEV, generator=g).bfloat16()
    k = torch.randn(1, Lk, C, device=DEV, generator=g).bfloat16()
    v = (torch.randn(Lk, C, device=DEV, generator=g) + torch.arange(C, device=DEV) % 128 * 0.01).bfloat16()
    k[0, 1090] = (q[0, -300:].float().mean(0) * 30).bfloat16()       # a spike inside the last split, seen by the tail rows
    vt = ops.transpose_pad(v)[None]
    qq = (q.float() * ops.q_prescale(128)).bfloat16() if pre else q
    out = ops.attention_fwd(qq, k, vt, H, q_prescaled=pre)
    ops.set_tuning("attn_tail", 0)
    try:
        plain = ops.attention_fwd(qq, k, vt, H, q_prescaled=pre)
    finally:
        ops.set_tuning("attn_tail", 1)
    assert torch.equal(out[:, :85 * 256], plain[:, :85 * 256])       # main launch untouched
    assert not torch.equal(out[:, 85 * 256:], plain[:, 85 * 256:])   # tail rows really took the other path
    assert rel_l2(out[:, 85 * 256:], plain[:, 85 * 256:].cpu()) < 3e-3
    rows = torch.cat([torch.arange(0, 40), torch.arange(Lq - 300, Lq)]).to(DEV)
    qe = (qq[0, rows].float() / ops.q_prescale(128)) if pre else q[0, rows].float()
    for h in range(H):
        p = torch.softmax(qe[:, h * 128:(h + 1) * 128] @ k[0, :, h * 128:(h + 1) * 128].float().t() / 128 ** 0.5, dim=-1)
        ref = p @ v[:, h * 128:(h + 1) * 128].float()
        assert rel_l2(out[0, rows, h * 128:(h + 1) * 128], ref.cpu()) < 6e-3
    again = ops.attention_fwd(qq, k, vt, H, q_prescaled=pre)
    assert torch.equal(again, out)


def _hdr(site):
    return site.buf[:16].view(torch.int32).tolist()        # [sticky switch, workgroups redone, lazy-reference repair events, -]


def test_attention_max_free_attempt_and_lazy_reference_fixup():
    """Pre-scaled q with scratch memory runs the max-free form (p = exp2(S), reference 0) and, right behind it, the lazy-reference
    form on every workgroup whose rows left the checked score window.  Without the attempt (tuning attn_fast = 0, or no scratch)
    ONE launch of the lazy-reference form does everything.  (i) ordinary scores: nothing flagged, no repair, both forms agree to
    bf16 noise and with the fp32 oracle; (ii) a key with log2-domain score ~ +300 in tile 10: exp2 overflows -> the max-free
    attempt flags the workgroup, the lazy form repairs its reference, and the flagged workgroups come out BITWISE equal to the
    one-launch lazy result; (iii) every score of some rows ~ -1000: the attempt's sums vanish -> flagged; the lazy form needs no
    repair (its reference starts at the row max of tile 0); (iv) the sticky switch; (v) it belongs to the call site."""
    Lq, Lk, H = 600, 1300, 2
    C = H * 128
    g = torch.Generator(device=DEV).manual_seed(12)
    c = ops.q_prescale(128)
    q = torch.randn(1, Lq, C, device=DEV, generator=g)
    k = torch.randn(1, Lk, C, device=DEV, generator=g).bfloat16()
    v = torch.randn(Lk, C, device=DEV, generator=g).bfloat16()
    vt = ops.transpose_pad(v)[None]

    site = ops.AttentionWorkspace()        # this call site's scratch (flags + sticky word + statistics)

    def run(qf, fast):
        ops.set_tuning("attn_fast", 2 if fast else 0)               # 2: the attempt even on this short launch (1 = long launches only)
        try:
            if site.buf is not None:
                site.buf[8:12].zero_()                               # repair-event counter
            out = ops.attention_fwd((qf * c).bfloat16(), k, vt, H, q_prescaled=True, workspace=site)
            variant = ops.get_tuning("last_attn_variant") & 15
        finally:
            ops.set_tuning("attn_fast", 1)
        ws = site.buf
        run.hdr = _hdr(site)
        flags = ws[16: 16 + 4 * 3 * H].view(torch.int32).clone()    # 3 query blocks x H workgroups
        ws[:4].zero_()                                              # a test must not switch the attempt off for the next one
        assert variant == (2 if fast else 1)                        # WAN_ATTN_VARIANT_W4_MAXFREE / _W4_LAZY
        return out, flags

    fast, flags = run(q, True)
    assert int(flags.sum()) == 0 and run.hdr[2] == 0
    lazy, _ = run(q, False)
    assert run.hdr[2] == 0                                          # ordinary scores never leave the window of the tile-0 reference
    # two independently bf16-rounded evaluations of one function (different softmax reference point): sqrt(2) x one run's error
    assert rel_l2(fast, lazy.cpu()) < 4.5e-3 and not torch.equal(fast, lazy)
    qe = (q * c).bfloat16().float() / c
    ref = _attn_ref(qe.view(1, Lq, H, 128).cpu(), k.view(1, Lk, H, 128).cpu(), v.view(1, Lk, H, 128).cpu()).view(1, Lq, C)
    assert rel_l2(fast, ref) < 6e-3 and rel_l2(lazy, ref) < 6e-3
    nows = ops.attention_fwd((q * c).bfloat16(), k, vt, H, q_prescaled=True, workspace=None)      # default per-stream scratch
    assert rel_l2(nows, ref) < 6e-3
    # (ii) a key aligned with the queries of block 1, head 0: log2-domain score 40 * 60 * c ~ +306 -> exp2 overflows
    u = torch.ones(128, device=DEV) / 128 ** 0.5
    q2 = q.clone()
    k2 = k.clone()
    k2[0, 700, :128] = (60 * u).bfloat16()
    q2[0, 256:512, :128] = 40 * u + 0.1 * q2[0, 256:512, :128]
    k_saved, k = k, k2                                             # `run` reads k from this scope
    fast2, flags2 = run(q2, True)
    assert run.hdr[2] >= 1                                         # the fix-up launch had to raise its reference
    lazy2, _ = run(q2, False)
    assert run.hdr[2] >= 4                                         # >= one repair per wave of the spiked workgroup
    assert torch.isfinite(fast2).all() and torch.isfinite(lazy2).all()
    flagged = flags2.view(H, 3).bool()                              # wg = qblk + 3 * head
    assert bool(flagged[0, 1]) and int(flags2.ne(0).sum()) >= 1
    for h in range(H):
        for b in range(3):
            blk_fast, blk_lazy = fast2[0, 256 * b:256 * (b + 1), 128 * h:128 * (h + 1)], lazy2[0, 256 * b:256 * (b + 1), 128 * h:128 * (h + 1)]
            if bool(flagged[h, b]):
                assert torch.equal(blk_fast, blk_lazy), (h, b)
            else:
                assert rel_l2(blk_fast, blk_lazy.cpu()) < 4.5e-3, (h, b)
    qe2 = (q2 * c).bfloat16().float() / c
    ref2 = _attn_ref(qe2.view(1, Lq, H, 128).cpu(), k.view(1, Lk, H, 128).cpu(), v.view(1, Lk, H, 128).cpu()).view(1, Lq, C)
    assert rel_l2(lazy2, ref2) < 6e-3
    k = k_saved
    # (iii) every score of some rows far below the window: l underflows in the max-free attempt -> flagged -> exact
    q3 = q.clone()
    kk = k.clone()
    kk[0, :, 128:] = (torch.ones(Lk, 128, device=DEV) * 4).bfloat16()
    q3[0, :40, 128:] = -16.0                                                                 # score = -16*4*128*c ~ -1000
    k = kk
    fast3, flags3 = run(q3, True)
    lazy3, _ = run(q3, False)
    assert run.hdr[2] == 0
    assert torch.isfinite(fast3).all() and bool(flags3.view(H, 3)[1, 0])
    assert torch.equal(fast3[0, :256, 128:], lazy3[0, :256, 128:])
    # (iv) the sticky switch: 1 of 6 workgroups redone (> 1/8) turns the attempt off for later calls on this scratch;
    #      they then run the lazy-reference kernel for every workgroup
    assert _hdr(site)[0] == 0
    ops.set_tuning("attn_fast", 2)
    try:
        out_a = ops.attention_fwd((q3 * c).bfloat16(), k, vt, H, q_prescaled=True, workspace=site)
        assert _hdr(site)[:2] == [1, 1] and torch.equal(out_a, fast3)   # this call itself was still an attempt
        k = k_saved
        out_b = ops.attention_fwd((q * c).bfloat16(), k, vt, H, q_prescaled=True, workspace=site)   # harmless input, switch still on
        assert _hdr(site)[:2] == [1, 6] and torch.equal(out_b, lazy)
        # (v) the switch belongs to the call site: another site (another layer type, another model) still makes the attempt
        other = ops.AttentionWorkspace()
        assert torch.equal(ops.attention_fwd((q * c).bfloat16(), k, vt, H, q_prescaled=True, workspace=other), fast)
        assert _hdr(other)[:2] == [0, 0]
        site.reset()
        assert torch.equal(ops.attention_fwd((q * c).bfloat16(), k, vt, H, q_prescaled=True, workspace=site), fast)
    finally:
        ops.set_tuning("attn_fast", 1)
    # (vi) the default policy: a launch this short (6 workgroups) is not worth a second launch -- one lazy-reference launch
    assert torch.equal(ops.attention_fwd((q * c).bfloat16(), k, vt, H, q_prescaled=True, workspace=site), lazy)
    assert ops.get_tuning("last_attn_variant") & 15 == 1


@pytest.mark.parametrize("pre", [True, False])
def test_attention_checkpoint_like_statistics(pre):
    """Scores shaped like a trained checkpoint's rather than a random initialisation's: RMS-normed q / k with per-channel
    gains log-normal up to 8 and one outlier channel per head that carries a large common offset (the massive-activation pattern)
    -- log2-domain scores of several hundred,
    row maxima that differ by > 100 between rows and grow along the sequence.  No window fits; the lazy reference must repair,
    and the result must still be the oracle's (attention_utils.py:115-146 has no input-dependent behaviour either)."""
    Lq, Lk, H = 1024, 4096, 2
    C = H * 128
    g = torch.Generator(device=DEV).manual_seed(77)
    gq = torch.exp(0.9 * torch.randn(C, device=DEV, generator=g)).clamp(max=8.0)
    gk = torch.exp(0.9 * torch.randn(C, device=DEV, generator=g)).clamp(max=8.0)

    def rms(x):
        x = x.view(-1, H, 128)
        return (x / x.pow(2).mean(-1, keepdim=True).sqrt()).view(-1, C)
    ramp = torch.linspace(0.2, 1.5, Lk, device=DEV)[:, None]                      # later keys score higher: references must rise
    qf = rms(torch.randn(Lq, C, device=DEV, generator=g)) * gq
    kf = rms(torch.randn(Lk, C, device=DEV, generator=g)) * gk * ramp
    for h in range(H):                                                            # outlier channels: same sign for every token
        qf[:, h * 128 + 5] = 40.0 + qf[:, h * 128 + 5].abs()
        kf[:, h * 128 + 5] = (45.0 + kf[:, h * 128 + 5].abs()) * ramp[:, 0]
    c = ops.q_prescale(128)
    k = kf.bfloat16()[None]
    v = (torch.randn(Lk, C, device=DEV, generator=g) + torch.arange(C, device=DEV) % 128 * 0.01).bfloat16()
    vt = ops.transpose_pad(v)[None]
    site = ops.AttentionWorkspace()
    if pre:
        qd = (qf * c).bfloat16()[None]
        qe = qd[0].float() / c
    else:
        qd = qf.bfloat16()[None]
        qe = qd[0].float()
    out = ops.attention_fwd(qd, k, vt, H, q_prescaled=pre, workspace=site)
    hdr = _hdr(site)
    s_max = float((qe[:, :128] @ k[0, :, :128].float().t()).abs().max() * c)
    ref = _attn_ref(qe.view(1, Lq, H, 128).cpu(), k.view(1, Lk, H, 128).cpu(), v.view(1, Lk, H, 128).cpu()).view(1, Lq, C)
    assert torch.isfinite(out).all()
    assert s_max > 300                                                             # really outside every fixed window
    assert hdr[2] >= 1, hdr                                                        # the lazy reference had to move
    assert rel_l2(out, ref) < 6e-3, (rel_l2(out, ref), hdr)
    print(f"checkpoint-like scores (pre={pre}): max |log2 score| {s_max:.0f}, flagged/redone workgroups {hdr[1]}, repair events {hdr[2]}, "
          f"rel_l2 {rel_l2(out, ref):.2e}")


def test_attention_rejects_unbuilt_options():
    z = torch.zeros(1, 8, 1, 128, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):
        attention(z, z, z, causal=True)
    with pytest.raises(NotImplementedError):
        attention(z, z, z, window_size=(4, 4))
    with pytest.raises(NotImplementedError):
        attention(z[..., :64], z[..., :64], z[..., :64])


def test_patchify_unpatchify_fixtures(golden):
    g = golden("dit_g4_norms")
    u = torch.from_numpy(g["u"])[0]
    out = ops.unpatchify(u.to(DEV), (7, 3, 5), (1, 2, 2), 16, torch.float32)
    assert torch.equal(out.cpu(), torch.from_numpy(g["unpatch"]))
    cfg = O.DiTConfig()
    x = torch.randn(16, 3, 8, 12)
    tok, grid = O.patchify(x, cfg)
    assert torch.equal(ops.patchify(x.to(DEV), (1, 2, 2)).cpu(), bf(tok))
    assert torch.equal(ops.patchify(bf(x).to(DEV), (1, 2, 2)).cpu(), bf(tok))
    with pytest.raises(ValueError, match="divisible"):
        ops.patchify(torch.zeros(16, 3, 7, 12, device=DEV), (1, 2, 2))


# ------------------------------------------------------------------ BASELINE-size properties
def test_attention_full_size_properties():
    """L = 67 080 tokens (81f@480p CoF), 2 heads: (i) V == const column => output == const
    (softmax rows sum to 1); (ii) permuting keys/values together leaves the output unchanged."""
    L, H = 67080, 2
    g = torch.Generator(device=DEV).manual_seed(0)
    q = torch.randn(1, L, H * 128, device=DEV, generator=g).bfloat16()
    k = torch.randn(1, L, H * 128, device=DEV, generator=g).bfloat16()
    colv = torch.linspace(-2, 2, H * 128, device=DEV).bfloat16()
    ld = ops.round_up(L, 64)
    vt = torch.zeros(1, H * 128, ld, device=DEV, dtype=torch.bfloat16)
    vt[0, :, :L] = colv[:, None]
    out = ops.attention_fwd(q, k, vt, H)
    assert float((out[0].float() - colv.float()).abs().max()) < 2e-2
    v = torch.randn(L, H * 128, device=DEV, generator=g).bfloat16()
    perm = torch.randperm(L, device=DEV, generator=g)
    qs = q[:, :4096]
    o1 = ops.attention_fwd(qs, k, ops.transpose_pad(v)[None], H)
    o2 = ops.attention_fwd(qs, k[:, perm].contiguous(), ops.transpose_pad(v[perm].contiguous())[None], H)
    # two independently bf16-rounded evaluations of the same function: sqrt(2) x the single-run error
    assert rel_l2(o1, o2.cpu()) < 4.5e-3
    # (iii) sampled queries against an fp32 evaluation of softmax(q k^T / sqrt(d)) v at the full key length
    rows = torch.tensor([0, 1, 31, 32, 255, 256, 1000, 4095], device=DEV)
    for h in range(H):
        qh = qs[0, rows, h * 128:(h + 1) * 128].float()
        kh = k[0, :, h * 128:(h + 1) * 128].float()
        p = torch.softmax(qh @ kh.t() / 128 ** 0.5, dim=-1)
        ref = p @ v[:, h * 128:(h + 1) * 128].float()
        assert rel_l2(o1[0, rows, h * 128:(h + 1) * 128], ref.cpu()) < 6e-3


def test_config4_length_321f_720p_rope_and_attention(rope_dev):
    """BASELINE configs[4] sequence length: VideoCoF layout of a 321-frame 720p clip = (81 + 1 + 81) x 45 x 80 =
    586 800 tokens, one head.  (i) RMSNorm + CoF RoPE on slabs at the start, across the source/ground/target
    boundaries and at the end vs the oracle evaluated at the same global token offsets; (ii) attention over the
    full length: constant V column => constant output, and sampled queries vs an fp32 softmax."""
    F, Hp, Wp = 163, 45, 80
    L, C, hw = F * Hp * Wp, 128, 45 * 80
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(L, 2 * C, device=DEV, generator=g).bfloat16()
    w = (torch.rand(C, device=DEV, generator=g) + 0.5)
    qk = x.clone()
    rp = RopeParams(F, Hp, Wp, 2, 81, 82, 0, L, 1024)
    qs = ops.q_prescale(128)
    ops.rmsnorm_rope_(qk[:, :C], w, qk[:, C:], w, 128, 1e-6, rope_dev, rp, x0_scale=qs)
    ang = O.rope_angles(128)
    for off in (0, 81 * hw - 130, 82 * hw - 130, L - 260):
        xs = x[off:off + 260, C:].float().cpu()
        ref = O.rope_apply(O.rms_norm(xs, w.cpu(), 1e-6).view(260, 1, 128), (F, Hp, Wp), ang, 81, (81, 82),
                           token_offset=off, total_tokens=L).reshape(260, C)
        assert rel_l2(qk[off:off + 260, C:], ref) < 4e-3, off
    # temporal positions really are src 1..81 | ground 0 | tgt 1..81: token of frame 82 (first target) == frame 0's map
    q, k = qk[None, :, :C], qk[None, :, C:]
    colv = torch.linspace(-2, 2, C, device=DEV).bfloat16()
    ld = ops.round_up(L, 64)
    vt = torch.zeros(1, C, ld, device=DEV, dtype=torch.bfloat16)
    vt[0, :, :L] = colv[:, None]
    out = ops.attention_fwd(q, k, vt, 1, q_prescaled=True)
    assert float((out[0].float() - colv.float()).abs().max()) < 2e-2
    v = torch.randn(L, C, device=DEV, generator=g).bfloat16()
    rows = torch.tensor([0, 255, 256, 300000, L - 257, L - 1], device=DEV)
    o = ops.attention_fwd(q[:, rows].contiguous(), k, ops.transpose_pad(v)[None], 1, q_prescaled=True)
    p = torch.softmax((q[0, rows].float() / qs) @ k[0].float().t() / 128 ** 0.5, dim=-1)
    assert rel_l2(o[0], (p @ v.float()).cpu()) < 6e-3


def test_gemm_full_size_vs_fp32_matmul_samples():
    """14B FFN shapes at M = 67 080: sampled rows vs an fp32 matmul of the same bf16 operands."""
    M, C, Fd = 67080, 5120, 13824
    g = torch.Generator(device=DEV).manual_seed(1)
    a = torch.randn(M, C, device=DEV, generator=g).bfloat16()
    w = (torch.randn(Fd, C, device=DEV, generator=g) * 0.02).bfloat16()
    bias = torch.randn(Fd, device=DEV, generator=g)
    out = ops.gemm(a, w, bias, ops.EPI_BF16)
    rows = torch.tensor([0, 1, 127, 128, 4095, 33333, 67071, 67072, 67079], device=DEV)
    ref = a[rows].float() @ w.float().t() + bias
    assert rel_l2(out[rows], ref.cpu()) < 4e-3
    vt = ops.gemm(a, w[:C], bias[:C], ops.EPI_BF16_T)
    assert rel_l2(vt[:, rows].t(), ref[:, :C].cpu()) < 4e-3


def test_lincomb_and_fused_unipc(golden):
    g = torch.Generator().manual_seed(2)
    xs = [torch.randn(3, 1000, generator=g) for _ in range(4)]
    cs = [0.7, -1.3, 0.25, 2.0]
    ref = sum(c * x for c, x in zip(cs, xs))
    out = ops.lincomb([(c, x.to(DEV)) for c, x in zip(cs, xs)], torch.float32)
    assert rel_l2(out, ref) < 1e-6
    outb = ops.lincomb([(c, x.bfloat16().to(DEV)) for c, x in zip(cs[:2], xs[:2])], torch.bfloat16)
    assert outb.dtype == torch.bfloat16 and rel_l2(outb, cs[0] * xs[0].bfloat16().float() + cs[1] * xs[1].bfloat16().float()) < 4e-3
    # the scheduler on device tensors goes through the fused kernel and reproduces the reference trajectory
    from videocof_amd import FlowUniPCMultistepScheduler
    gg = golden("dit_g7_unipc")
    s = FlowUniPCMultistepScheduler(shift=1)
    s.set_timesteps(4, device=DEV, shift=3)
    cur = torch.from_numpy(gg["x"]).to(DEV)
    for i, t in enumerate(s.timesteps):
        cur = s.step(torch.from_numpy(gg["v"][i]).to(DEV), t, cur, return_dict=False)[0]
        assert rel_l2(cur, gg["traj"][i]) < 2e-6


@pytest.mark.parametrize("w4", [0, 1])
@pytest.mark.parametrize("M,N,K", [(1100, 520, 512), (2048, 256, 128), (1500, 1536, 1024)])
def test_gemm_256_tile_kernels_all_epilogues(M, N, K, w4):
    """Both 256^2 kernels (8-wave phased, 4-wave with AGPR accumulators) on ragged M / N with K % 128 == 0: every epilogue
    against an fp64 product of the same bf16 operands."""
    g = torch.Generator().manual_seed(M + N + K + w4)
    a, w = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) * 0.1)
    bias = torch.randn(N, generator=g) * 0.5
    gate = torch.randn(2, N, generator=g)
    resid = torch.randn(M, N, generator=g)
    acc = a.double() @ w.double().t() + bias.double()
    ad, wd, bd = a.to(DEV), w.to(DEV), bias.to(DEV)
    ops.set_tuning("gemm_variant", 2)
    ops.set_tuning("gemm_w4", 3 if w4 else 0)          # 3 = the 4-wave kernel whenever K % 128 == 0 (the default takes it from K = 4096)
    try:
        o_bf = ops.gemm(ad, wd, bd, ops.EPI_BF16)
        o_ge = ops.gemm(ad, wd, bd, ops.EPI_GELU_BF16)
        o_f32 = ops.gemm(ad, wd, bd, ops.EPI_F32)
        o_res = resid.to(DEV).clone()
        rpb = (M + 1) // 2
        ops.gemm(ad, wd, bd, ops.EPI_RESID_F32, out=o_res, gate=gate.to(DEV), rows_per_batch=rpb)
        o_t = ops.gemm(ad, wd, bd, ops.EPI_BF16_T)
    finally:
        ops.set_tuning("gemm_variant", 0)
        ops.set_tuning("gemm_w4", 1)
    assert rel_l2(o_bf, acc) < 4e-3 and rel_l2(o_f32, acc) < 1e-5
    x = acc.float().double()
    assert rel_l2(o_ge, 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))) < 5e-3
    gsel = gate.double()[torch.arange(M) // rpb]
    assert rel_l2(o_res, resid.double() + acc * gsel) < 1e-5
    assert o_t.shape[0] == N and rel_l2(o_t[:, :M].t(), acc) < 4e-3
    assert float(o_t[:, M:].abs().max()) == 0.0 if o_t.shape[1] > M else True


@pytest.mark.parametrize("form", [1, 0])
@pytest.mark.parametrize("M,N,K", [(1100, 520, 512), (2304, 1536, 896), (4100, 2100, 256), (9000, 5120, 640), (3000, 1164, 384)])
def test_persistent_stream_k_gemm_all_epilogues(M, N, K, form):
    """wan_gemm_bf16_ws (gemm_pk_kernel: one resident workgroup per CU, 16x16x32 MFMAs, continuous K-tile stream, tiles by
    per-XCD ticket, stream-K remainder combined in K order) at shapes that are mostly or partly SPLIT tiles (15 .. 720 tiles on 256
    workers), ragged M / N: every epilogue against an fp64 product of the same bf16 operands; bitwise run-to-run (the combine
    order does not depend on who arrives last); garbage in the workspace does not matter.  form 1 (product): the epilogues run from
    row-permuted operand tiles (a lane's accumulators contiguous in the output: 16-byte stores of whole row segments); form 0: the
    round-4 epilogues.  (3000, 1164, 384): N is not a multiple of 8 -- the last 4-column group of a row takes the 8-byte path -- and
    the transposed output's row stride (3008) keeps 16-byte alignment while M does not fill the last 8-token group."""
    g = torch.Generator().manual_seed(M + N + K)
    a, w = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) * 0.1)
    bias = torch.randn(N, generator=g) * 0.5
    gate = torch.randn(2, N, generator=g)
    resid = torch.randn(M, N, generator=g)
    acc = a.double() @ w.double().t() + bias.double()
    ad, wd, bd = a.to(DEV), w.to(DEV), bias.to(DEV)
    rpb = (M + 1) // 2
    ops.set_tuning("gemm_pk", 2)                       # 2 = whenever K % 128 == 0 (the default takes shapes the 4-wave kernel would)
    ops.set_tuning("gemm_pk_form", 31 if form else 0)
    # the stream-K cut at a quarter of a tile's K range, whatever the plan's own choice for the shape (round 6: leftover tiles of short
    # launches go whole where the fix-up would cost more than the idle lanes -- this test is about the cut)
    ops.set_tuning("gemm_pk_min_units", max(1, (K // 128 + 3) // 4))
    try:
        from videocof_amd import _lib
        assert _lib.load().wan_gemm_ws_plan(M, N, K) == 3
        ws = ops.gemm_workspace(ad.device, M, N, K)
        assert ws is not None and ws.numel() == _lib.load().wan_gemm_workspace_bytes(M, N, K)
        runs = []
        for rep in range(2):
            ws.fill_(0xA5 if rep else 0xFF)            # the kernel must not depend on what the workspace holds
            o_res = resid.to(DEV).clone()
            ops.gemm(ad, wd, bd, ops.EPI_RESID_F32, out=o_res, gate=gate.to(DEV), rows_per_batch=rpb)
            runs.append((ops.gemm(ad, wd, bd, ops.EPI_BF16), ops.gemm(ad, wd, bd, ops.EPI_GELU_BF16), ops.gemm(ad, wd, bd, ops.EPI_F32),
                         o_res, ops.gemm(ad, wd, None, ops.EPI_BF16_T)))
        ops.set_tuning("gemm_pk_min_units", 0)         # ... and once with the plan's own choice (whole leftover tiles at some of these shapes)
        own = (ops.gemm(ad, wd, bd, ops.EPI_BF16), ops.gemm(ad, wd, bd, ops.EPI_F32))
    finally:
        ops.set_tuning("gemm_pk", 1)
        ops.set_tuning("gemm_pk_form", 31)
        ops.set_tuning("gemm_pk_min_units", 0)
    assert rel_l2(own[0], acc) < 4e-3 and rel_l2(own[1], acc) < 1e-5
    o_bf, o_ge, o_f32, o_res, o_t = runs[0]
    assert all(torch.equal(x, y) for x, y in zip(runs[0], runs[1]))
    assert rel_l2(o_bf, acc) < 4e-3 and rel_l2(o_f32, acc) < 1e-5
    x = acc.float().double()
    assert rel_l2(o_ge, 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))) < 5e-3
    gsel = gate.double()[torch.arange(M) // rpb]
    assert rel_l2(o_res, resid.double() + acc * gsel) < 1e-5
    assert o_t.shape[0] == N and rel_l2(o_t[:, :M].t(), acc - bias.double()) < 4e-3
    assert float(o_t[:, M:].abs().max()) == 0.0 if o_t.shape[1] > M else True


@pytest.mark.parametrize("aligned", [True, False])
def test_persistent_gemm_row_permuted_epilogues_equal_the_round4_epilogues_bitwise(aligned):
    """gemm_pk_form = 1 changes WHERE a panel row sits in LDS (so that a lane's accumulators are contiguous in the output) and
    nothing about the arithmetic: every output element is the same MFMA chain in the same k order.  All five epilogues, ragged
    M / N, a sample seam inside a wave's rows, and -- `aligned` False -- output row strides that are multiples of 4 but not of 8
    (bf16 rows then start on 8-byte boundaries only: the 16-byte stores fall back to 8-byte ones): bitwise equal to form 0."""
    M, N, K = 2900, 1672, 640
    g = torch.Generator().manual_seed(77)
    a, w = bf(torch.randn(M, K, generator=g)).to(DEV), bf(torch.randn(N, K, generator=g) * 0.1).to(DEV)
    bias, gate = torch.randn(N, generator=g).to(DEV), torch.randn(3, N, generator=g).to(DEV)
    resid = torch.randn(M, N, generator=g).to(DEV)
    pad = 0 if aligned else 4
    ldt = ops.round_up(M, 64) + pad
    res = {}
    ops.set_tuning("gemm_pk", 2)
    try:
        for form in (1, 0):
            ops.set_tuning("gemm_pk_form", 31 if form else 0)
            o_bf = torch.zeros(M, N + pad, device=DEV, dtype=torch.bfloat16)[:, :N]
            o_ge = torch.zeros(M, N + pad, device=DEV, dtype=torch.bfloat16)[:, :N]
            o_f = torch.zeros(M, N + pad, device=DEV, dtype=torch.float32)[:, :N]
            o_r = torch.zeros(M, N + pad, device=DEV, dtype=torch.float32)[:, :N]
            o_r.copy_(resid)
            o_t = torch.zeros(N, ldt, device=DEV, dtype=torch.bfloat16)
            ops.gemm(a, w, bias, ops.EPI_BF16, out=o_bf)
            ops.gemm(a, w, bias, ops.EPI_GELU_BF16, out=o_ge)
            ops.gemm(a, w, None, ops.EPI_F32, out=o_f)
            ops.gemm(a, w, bias, ops.EPI_RESID_F32, out=o_r, gate=gate, rows_per_batch=1000)       # seams at rows 1000 and 2000
            ops.gemm(a, w, bias, ops.EPI_BF16_T, out=o_t)
            res[form] = [t.clone() for t in (o_bf, o_ge, o_f, o_r, o_t)]
    finally:
        ops.set_tuning("gemm_pk", 1)
        ops.set_tuning("gemm_pk_form", 31)
    for name, x, y in zip(("bf16", "gelu", "f32", "resid", "transposed"), res[1], res[0]):
        assert torch.equal(x, y), name
    acc = a.double() @ w.double().t()
    gsel = gate.double()[torch.arange(M, device=DEV) // 1000]
    assert rel_l2(res[1][3], resid.double() + (acc + bias.double()) * gsel) < 1e-5
    assert rel_l2(res[1][4][:, :M].t(), acc + bias.double()) < 4e-3 and float(res[1][4][:, M:].abs().max()) == 0.0


@pytest.mark.parametrize("splits", [0, 2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(2304, 1536, 8960), (515, 196, 4096), (1000, 520, 6144)])
def test_split_k_form_of_the_small_shape_gemm(M, N, K, splits):
    """wan_gemm_bf16_ws on shapes the 128^2 kernel takes and whose tiles do not fill the chip (BASELINE configs[0]: M = 2 304): the
    K range of every tile cut into pieces (by shape: splits = 0 here -> wan_gemm_ws_splits; or forced 2 / 3 / 4), fp32 pieces in the
    caller's workspace, combined IN SPLIT ORDER by the last arriver.  Every epilogue against the fp64 product, ragged M / N,
    bitwise run to run, garbage in the workspace; and really split (workspace requested, plan says so)."""
    from videocof_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    a, w = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) * 0.1)
    bias, gate, resid = torch.randn(N, generator=g) * 0.5, torch.randn(2, N, generator=g), torch.randn(M, N, generator=g)
    acc = a.double() @ w.double().t() + bias.double()
    ad, wd, bd = a.to(DEV), w.to(DEV), bias.to(DEV)
    rpb = (M + 1) // 2
    if splits:
        ops.set_tuning("gemm_splitk", splits)
    try:
        assert lib.wan_gemm_ws_plan(M, N, K) == 0
        n_splits = int(lib.wan_gemm_ws_splits(M, N, K))
        assert n_splits == (splits or n_splits) and n_splits >= 2, (n_splits, splits)
        ws = ops.gemm_workspace(ad.device, M, N, K)
        assert ws is not None and ws.numel() >= lib.wan_gemm_workspace_bytes(M, N, K) > 0
        runs = []
        for rep in range(2):
            ws.fill_(0xA5 if rep else 0xFF)
            o_res = resid.to(DEV).clone()
            ops.gemm(ad, wd, bd, ops.EPI_RESID_F32, out=o_res, gate=gate.to(DEV), rows_per_batch=rpb)
            runs.append((ops.gemm(ad, wd, bd, ops.EPI_BF16), ops.gemm(ad, wd, bd, ops.EPI_GELU_BF16), ops.gemm(ad, wd, bd, ops.EPI_F32),
                         o_res, ops.gemm(ad, wd, None, ops.EPI_BF16_T)))
        ops.set_tuning("gemm_splitk", 0)
        plain = ops.gemm(ad, wd, bd, ops.EPI_F32)            # the unsplit kernel: another summation order, the same product
    finally:
        ops.set_tuning("gemm_splitk", 1)
    o_bf, o_ge, o_f32, o_res, o_t = runs[0]
    assert all(torch.equal(x, y) for x, y in zip(runs[0], runs[1]))
    assert rel_l2(o_bf, acc) < 4e-3 and rel_l2(o_f32, acc) < 1e-5 and rel_l2(o_f32, plain) < 1e-5
    x = acc.float().double()
    assert rel_l2(o_ge, 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))) < 5e-3
    gsel = gate.double()[torch.arange(M) // rpb]
    assert rel_l2(o_res, resid.double() + acc * gsel) < 1e-5
    assert o_t.shape[0] == N and rel_l2(o_t[:, :M].t(), acc - bias.double()) < 4e-3
    assert float(o_t[:, M:].abs().max()) == 0.0 if o_t.shape[1] > M else True


def test_persistent_gemm_workspace_is_per_stream():
    """ops.gemm's workspace (arrival / ticket counters and split-tile partial sums of the persistent kernel) is keyed by
    (device, stream): two streams issuing large Linears of one device CONCURRENTLY never share counters.  Both streams run the
    same mostly-split product many times, interleaved from the host without any synchronisation between them; every result
    must be the bitwise result of the product run alone."""
    M, N, K = 9000, 5120, 640
    g = torch.Generator().manual_seed(5)
    a, w = bf(torch.randn(M, K, generator=g)).to(DEV), bf(torch.randn(N, K, generator=g) * 0.1).to(DEV)
    b2 = bf(torch.randn(M, K, generator=g)).to(DEV)
    ops.set_tuning("gemm_pk", 2)
    try:
        ref_a, ref_b = ops.gemm(a, w, None, ops.EPI_BF16), ops.gemm(b2, w, None, ops.EPI_BF16)
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        with torch.cuda.stream(s1):
            ws1 = ops.gemm_workspace(a.device, M, N, K)
        with torch.cuda.stream(s2):
            ws2 = ops.gemm_workspace(a.device, M, N, K)
        assert ws1.data_ptr() != ws2.data_ptr() and ws1.data_ptr() != ops.gemm_workspace(a.device, M, N, K).data_ptr()
        outs1 = [torch.empty_like(ref_a) for _ in range(6)]
        outs2 = [torch.empty_like(ref_b) for _ in range(6)]
        for o1, o2 in zip(outs1, outs2):
            with torch.cuda.stream(s1):
                ops.gemm(a, w, None, ops.EPI_BF16, out=o1)
            with torch.cuda.stream(s2):
                ops.gemm(b2, w, None, ops.EPI_BF16, out=o2)
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            assert ops.gemm_workspace(a.device, M, N, K).data_ptr() == ws1.data_ptr()       # stable per stream
    finally:
        ops.set_tuning("gemm_pk", 1)
    assert all(torch.equal(o, ref_a) for o in outs1) and all(torch.equal(o, ref_b) for o in outs2)


def test_persistent_gemm_gate_with_short_samples_takes_the_per_tile_kernels():
    """The persistent kernel's read-modify-write epilogue allows ONE sample seam per 128-row wave; a gate whose samples are shorter
    (rows_per_batch < 128) is served by the per-tile kernels behind the same entry point -- same result as the fp64 statement."""
    M, N, K, rpb = 2304, 1536, 896, 100
    g = torch.Generator().manual_seed(21)
    a, w = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) * 0.1)
    bias, gate, resid = torch.randn(N, generator=g), torch.randn((M + rpb - 1) // rpb, N, generator=g), torch.randn(M, N, generator=g)
    ops.set_tuning("gemm_pk", 2)
    try:
        outs = []
        for r in (rpb, 1152):                              # 1152: the persistent kernel (one seam, in the middle of a wave's rows)
            o = resid.to(DEV).clone()
            ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), ops.EPI_RESID_F32, out=o, gate=gate.to(DEV), rows_per_batch=r)
            gsel = gate.double()[torch.arange(M) // r]
            assert rel_l2(o, resid.double() + (a.double() @ w.double().t() + bias.double()) * gsel) < 1e-5, r
            outs.append(o)
    finally:
        ops.set_tuning("gemm_pk", 1)


def test_persistent_gemm_beyond_4gib():
    """gemm_pk_kernel where byte offsets leave 32 bits (BASELINE configs[4] on one device: 586 800 tokens x 5 120 fp32 = 12 GB of
    residual stream, 16 GB of ffn activations): an A operand of 4.4 GB (M = 330 000 rows of K = 6 656) and an fp32 read-modify-write
    output of 4.7 GB (N = 3 584), sampled rows on both sides of the 2^31 / 2^32-byte marks against an fp64 product."""
    from videocof_amd import _lib
    M, N, K = 330000, 3584, 6656
    assert M * K * 2 > 2 ** 32 and M * N * 4 > 2 ** 32
    assert _lib.load().wan_gemm_ws_plan(M, N, K) == 3
    g = torch.Generator(device=DEV).manual_seed(11)
    a = torch.empty(M, K, device=DEV, dtype=torch.bfloat16)
    for r0 in range(0, M, 66000):                      # (filled in slabs: a 4.4 GB randn in fp32 would be 8.8 GB of temporaries)
        a[r0:r0 + 66000] = torch.randn(min(66000, M - r0), K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g)
    rows = torch.tensor([0, 255, 256, 161319, 161320, 299592, 299593, 322638, 322639, 329999], device=DEV)
    want = a[rows].double() @ w.double().t() + bias.double()
    o = ops.gemm(a, w, bias, ops.EPI_BF16)
    assert rel_l2(o[rows], want) < 4e-3
    del o
    x = torch.zeros(M, N, device=DEV)
    x[rows] = 1.0
    ops.gemm(a, w, None, ops.EPI_RESID_F32, out=x)
    assert rel_l2(x[rows], want - bias.double() + 1.0) < 1e-5
    # the rows nobody sampled: no row written twice or skipped -- column sums against a GEMV (one row off moves them by ~2e-3)
    cs = (x.sum(dim=0, dtype=torch.float64) - len(rows)).cpu()
    ref = (a.float().sum(dim=0, dtype=torch.float64) @ w.double().t()).cpu()
    assert rel_l2(cs, ref) < 2e-4


@pytest.mark.parametrize("P,T,B,Cl", [(2, 24, 2, 128), (8, 56, 1, 640), (4, 8, 3, 8)])
def test_sp_wire_layout_kernels(rope_dev, P, T, B, Cl):
    """wan_sp_pack_heads / wan_sp_unpack_heads / wan_sp_unpack_vt move bytes exactly like the torch statements of the
    layouts in videocof_amd.dist, and wan_rmsnorm_rope_sp == wan_rmsnorm_rope followed by the pack."""
    import types
    from videocof_amd.dist import SequenceParallelGroup
    sp = types.SimpleNamespace(world_size=P)
    ref = lambda name, *a: getattr(SequenceParallelGroup, name)(sp, *a)
    g = torch.Generator().manual_seed(P + T)
    C = P * Cl
    x = bf(torch.randn(B * T, C + 16, generator=g)).to(DEV)[:, :C]              # strided rows (ldx = C + 16)
    wire = torch.empty(B * T * C, device=DEV, dtype=torch.bfloat16)
    ops.sp_pack_heads(x, wire, P, T, B)
    want = ref("pack_heads_ref", x.view(B, T, C))
    assert torch.equal(wire.view(P, T, B, Cl), want)
    back = torch.zeros(B * T, C, device=DEV, dtype=torch.bfloat16)
    ops.sp_unpack_heads(wire, back, P, T, B)
    assert torch.equal(back, ref("unpack_heads_ref", wire.view(P, T, B, Cl)).reshape(B * T, C))
    vwire = bf(torch.randn(P, Cl, B, T, generator=g)).to(DEV)
    ld = P * T + 40
    vt = torch.zeros(B, Cl, ld, device=DEV, dtype=torch.bfloat16)
    ops.sp_unpack_vt(vwire.view(-1), vt, P, T)
    assert torch.equal(vt, ref("unpack_vt_ref", vwire, ld))
    if Cl % 128 == 0:                                                          # head_dim 128 rows
        w = (torch.rand(C, generator=g) + 0.5).to(DEV)
        rp = RopeParams(T // 4 if T % 4 == 0 else 1, 2, 2, 2, 1, 2, 16, T, 1024)
        xin = x.contiguous()
        inplace = xin.clone()
        ops.rmsnorm_rope_(inplace, w, None, None, 128, 1e-6, rope_dev, rp, x0_scale=0.37)
        w2 = torch.empty(B * T * C, device=DEV, dtype=torch.bfloat16)
        keep = xin.clone()
        ops.rmsnorm_rope_sp(xin, w, None, None, 128, 1e-6, rope_dev, rp, w2, None, P, B, x0_scale=0.37)
        assert torch.equal(xin, keep)                                           # the input is not modified
        assert torch.equal(w2.view(P, T, B, Cl), ref("pack_heads_ref", inplace.view(B, T, C)))
        if Cl >= 256:                                                           # two head groups (the *_split entry points)
            split = (Cl // 128 // 2) * 128
            w3 = torch.empty_like(w2)
            ops.rmsnorm_rope_sp(xin, w, None, None, 128, 1e-6, rope_dev, rp, w3, None, P, B, x0_scale=0.37, split=split)
            assert torch.equal(w3, SequenceParallelGroup.split_wire_ref(w2.view(P, T, B, Cl), split))
    for split in ([8 * (Cl // 16)] if Cl >= 16 else []):
        flat = SequenceParallelGroup.split_wire_ref(want, split)
        w4 = torch.empty_like(wire)
        ops.sp_pack_heads(x, w4, P, T, B, split=split)
        assert torch.equal(w4, flat)
        assert torch.equal(SequenceParallelGroup.join_wire_ref(flat, P, T, B, Cl, split), want)
        back2 = torch.zeros_like(back)
        ops.sp_unpack_heads(flat, back2, P, T, B, split=split)
        assert torch.equal(back2, back)


def test_row_kernels_rows_per_workgroup_are_bitwise_the_one_row_kernels(rope_dev):
    """LN-modulate and RMSNorm+RoPE with 2 (default) or 4 token rows per workgroup -- the per-column parameters fetched once per group --
    against the one-row kernels: same arithmetic, same summation order, so every output form must agree BITWISE; row counts that
    leave a ragged last group, samples whose boundary falls inside a group, strided rows, the wire and e4m3 output forms."""
    g = torch.Generator().manual_seed(5)
    for dim, rows_per_batch, B in ((5120, 37, 2), (1536, 50, 1), (256, 9, 3)):
        rows = rows_per_batch * B
        x = torch.randn(rows, dim, generator=g).to(DEV)
        sc, sh = torch.randn(B, dim, generator=g).to(DEV), torch.randn(B, dim, generator=g).to(DEV)
        qk = bf(torch.randn(rows, 2 * dim + 16, generator=g)).to(DEV)
        w0, w1 = (torch.rand(dim, generator=g) + 0.5).to(DEV), (torch.rand(dim, generator=g) + 0.5).to(DEV)
        rp = RopeParams(3, 3, 5, 2, 1, 2, 4, rows_per_batch, 1024)            # CoF map, token offset 4, rows past the 45-token grid
        P = 2
        res = {}
        for R in (1, 2, 4):
            ops.set_tuning("row_group", R)
            try:
                ln = ops.ln_modulate(x, sc, sh, True, rows_per_batch, 1e-6)
                ln_plain = ops.ln_modulate(x, None, None, False, rows, 1e-6)
                a = qk.clone()
                ops.rmsnorm_rope_(a[:, :dim], w0, a[:, dim:2 * dim], w1, 128, 1e-6, rope_dev, rp, x0_scale=0.37)
                wire = torch.zeros(rows * dim, device=DEV, dtype=torch.bfloat16)
                ops.rmsnorm_rope_sp(qk[:, :dim], w0, None, None, 128, 1e-6, rope_dev, rp, wire, None, P, B, x0_scale=0.37,
                                    split=(dim // P // 128 // 2) * 128)
                q8, k8 = torch.zeros(rows, dim, device=DEV, dtype=ops.FP8), torch.zeros(rows, dim, device=DEV, dtype=ops.FP8)
                ops.rmsnorm_rope_fp8(qk[:, :dim], w0, qk[:, dim:2 * dim], w1, 128, 1e-6, rope_dev, rp, q8, k8, x0_scale=4.0, x1_scale=2.0)
                nr = qk[:, :dim].clone()
                ops.rmsnorm_rope_(nr, w0, None, None, 128, 1e-6)                  # no rotation (the cross-attention q form)
                res[R] = (ln, ln_plain, a, wire, q8.view(torch.uint8), k8.view(torch.uint8), nr)
            finally:
                ops.set_tuning("row_group", 2)
        for R in (2, 4):
            for i, (u, v) in enumerate(zip(res[1], res[R])):
                assert torch.equal(u, v), (dim, R, i)


def test_attention_debug_check_catches_non_finite_vt_padding():
    """The one caller contract the kernels cannot enforce -- V^T pad columns [Lk, roundup(Lk, 64)) finite -- is checked
    (synchronising) when the `debug_checks` switch is on, and costs nothing when it is off."""
    Lq, Lk, H = 256, 100, 1
    g = torch.Generator(device=DEV).manual_seed(3)
    q = torch.randn(1, Lq, 128, device=DEV, generator=g).bfloat16()
    k = torch.randn(1, Lk, 128, device=DEV, generator=g).bfloat16()
    vt = torch.zeros(1, 128, 128, device=DEV, dtype=torch.bfloat16)
    vt[:, :, :Lk] = torch.randn(1, 128, Lk, device=DEV, generator=g).bfloat16()
    good = ops.attention_fwd(q, k, vt, H, k_len=Lk)
    vt[0, 5, Lk + 3] = float("nan")
    ops.set_tuning("debug_checks", 1)
    try:
        with pytest.raises(ValueError, match="pad columns"):
            ops.attention_fwd(q, k, vt, H, k_len=Lk)
        vt[0, 5, Lk + 3] = 0
        assert torch.equal(ops.attention_fwd(q, k, vt, H, k_len=Lk), good)
    finally:
        ops.set_tuning("debug_checks", 0)
