"""-m gpu: randomised-shape parity of every kernel against a plain PyTorch fp32 evaluation of the
same op on the GPU (fast enough to sweep dozens of shapes), plus shapes beyond 2^31 bytes."""
import math
import random

import pytest
import torch

from videocof_amd import ops
from videocof_amd._lib import RopeParams

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_gemm_random_shapes():
    rnd = random.Random(0)
    g = torch.Generator(device=DEV).manual_seed(0)
    for _ in range(24):
        M = rnd.choice([1, 7, 63, 64, 65, 255, 256, 257, 1000, 1024, 1300, 2049])
        N = rnd.choice([4, 60, 64, 128, 192, 260, 512, 1028])
        K = 64 * rnd.randint(1, 12)
        a = torch.randn(M, K, device=DEV, generator=g).bfloat16()
        w = (torch.randn(N, K, device=DEV, generator=g) * 0.1).bfloat16()
        bias = torch.randn(N, device=DEV, generator=g)
        ref = a.float() @ w.float().t() + bias
        for epi in (ops.EPI_BF16, ops.EPI_F32, ops.EPI_GELU_BF16):
            out = ops.gemm(a, w, bias, epi)
            r = torch.nn.functional.gelu(ref, approximate="tanh") if epi == ops.EPI_GELU_BF16 else ref
            assert rel_l2(out.float(), r) < (1e-5 if epi == ops.EPI_F32 else 4e-3), (M, N, K, epi)
        vt = ops.gemm(a, w, bias, ops.EPI_BF16_T)
        assert rel_l2(vt[:, :M].t().float(), ref) < 4e-3, (M, N, K, "T")
        x = torch.randn(M, N, device=DEV, generator=g)
        gate = torch.randn(1, N, device=DEV, generator=g)
        y = x.clone()
        ops.gemm(a, w, bias, ops.EPI_RESID_F32, out=y, gate=gate, rows_per_batch=M)
        assert rel_l2(y, x + ref * gate) < 1e-5, (M, N, K, "resid")


def test_attention_random_shapes():
    rnd = random.Random(1)
    g = torch.Generator(device=DEV).manual_seed(1)
    for _ in range(16):
        B = rnd.choice([1, 2])
        H = rnd.choice([1, 2, 5])
        Lq = rnd.choice([1, 31, 32, 33, 255, 256, 257, 700])
        Lk = rnd.choice([1, 63, 64, 65, 127, 128, 129, 512, 1000, 1025, 1600])
        C = H * 128
        q = (torch.randn(B, Lq, C, device=DEV, generator=g) * rnd.choice([0.5, 1.0, 3.0])).bfloat16()
        k = torch.randn(B, Lk, C, device=DEV, generator=g).bfloat16()
        v = torch.randn(B, Lk, C, device=DEV, generator=g).bfloat16()
        vt = torch.stack([ops.transpose_pad(v[b]) for b in range(B)])
        out = ops.attention_fwd(q, k, vt, H)
        qf = q.float().view(B, Lq, H, 128).transpose(1, 2)
        kf = k.float().view(B, Lk, H, 128).transpose(1, 2)
        vf = v.float().view(B, Lk, H, 128).transpose(1, 2)
        ref = torch.softmax(qf @ kf.transpose(-1, -2) / math.sqrt(128), dim=-1) @ vf
        ref = ref.transpose(1, 2).reshape(B, Lq, C)
        assert rel_l2(out.float(), ref) < 6e-3, (B, H, Lq, Lk)


def test_row_kernels_random_shapes():
    rnd = random.Random(2)
    g = torch.Generator(device=DEV).manual_seed(2)
    ang = None
    for _ in range(12):
        heads = rnd.choice([1, 2, 3, 12, 40])
        C = heads * 128
        rows = rnd.choice([1, 5, 64, 333])
        x = torch.randn(rows, C, device=DEV, generator=g) * 2 + 0.1
        sc, sh = torch.randn(1, C, device=DEV, generator=g), torch.randn(1, C, device=DEV, generator=g)
        out = ops.ln_modulate(x, sc, sh, True, rows, 1e-6)
        ref = torch.nn.functional.layer_norm(x, (C,), eps=1e-6) * (1 + sc) + sh
        assert rel_l2(out.float(), ref) < 4e-3
        xb = x.bfloat16()
        w = torch.rand(C, device=DEV, generator=g) + 0.5
        y = xb.clone()
        ops.rmsnorm_rope_(y, w, None, None, 128, 1e-6)
        xf = xb.float()
        assert rel_l2(y.float(), xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * w) < 4e-3


def test_beyond_2gib_indexing():
    """720p CoF token count (154 800) x 14B q|k width = 3.2 GB: 64-bit row offsets in the row kernels,
    the GEMM epilogue and the attention loads/stores."""
    L, C = 154800, 5120
    g = torch.Generator(device=DEV).manual_seed(3)
    qk = torch.randn(L, 2 * C, device=DEV, generator=g).bfloat16()
    assert qk.numel() * 2 > 2 ** 31
    w = torch.ones(C, device=DEV)
    rows = torch.tensor([0, 1, 104857, 104858, 154799], device=DEV)      # rows beyond the 2^31-byte mark
    before = qk[rows].float().clone()
    ops.rmsnorm_rope_(qk[:, :C], w, qk[:, C:], w, 128, 1e-6)
    for part in (slice(0, C), slice(C, 2 * C)):
        b = before[:, part]
        assert rel_l2(qk[rows][:, part].float(), b * torch.rsqrt(b.pow(2).mean(-1, keepdim=True) + 1e-6)) < 4e-3
    # attention over the strided q|k buffer, one head, sampled query rows
    H = 1
    v = torch.randn(L, 128, device=DEV, generator=g).bfloat16()
    vt = ops.transpose_pad(v)[None]
    out = ops.attention_fwd(qk[None, :, :128], qk[None, :, C:C + 128], vt, H)
    qs = qk[rows, :128].float()
    p = torch.softmax(qs @ qk[:, C:C + 128].float().t() / math.sqrt(128), dim=-1)
    assert rel_l2(out[0, rows].float(), p @ v.float()) < 6e-3
    # GEMM writing a > 2 GiB output
    a = torch.randn(L, 64, device=DEV, generator=g).bfloat16()
    wt = (torch.randn(8192, 64, device=DEV, generator=g) * 0.1).bfloat16()
    o = ops.gemm(a, wt, None, ops.EPI_BF16)
    assert o.numel() * 2 > 2 ** 31
    assert rel_l2(o[rows].float(), a[rows].float() @ wt.float().t()) < 4e-3


def test_attention_split_tail_random_shapes():
    """Shapes whose workgroup count leaves a remainder over the CU count: whatever the launcher decides (plain
    launch or main launch + split tail + merge), the result equals an fp32 evaluation on sampled query rows and
    agrees with the plain launch (tuning attn_tail = 0) to bf16 noise."""
    from videocof_amd import _lib
    rnd = random.Random(5)
    g = torch.Generator(device=DEV).manual_seed(5)
    split_seen = 0
    for _ in range(10):
        B = rnd.choice([1, 1, 2])
        H = rnd.choice([2, 3, 5, 7])
        nqb = rnd.randint(256 // (B * H) + 1, 3 * 256 // (B * H) + 2)
        Lq = nqb * 256 - rnd.randint(0, 255)
        Lk = rnd.choice([1025, 1100, 1664, 2500, 4096])
        pre = rnd.random() < 0.5
        C = H * 128
        flag_bytes = (16 + (Lq + 255) // 256 * H * B * 4 + 255) // 256 * 256
        split_seen += int(_lib.load().wan_attention_workspace_bytes(B, Lq, Lk, H, 128) > flag_bytes)
        q = torch.randn(B, Lq, C, device=DEV, generator=g).bfloat16()
        k = torch.randn(B, Lk, C, device=DEV, generator=g).bfloat16()
        v = torch.randn(B, Lk, C, device=DEV, generator=g).bfloat16()
        vt = torch.stack([ops.transpose_pad(v[b]) for b in range(B)])
        qq = (q.float() * ops.q_prescale(128)).bfloat16() if pre else q
        out = ops.attention_fwd(qq, k, vt, H, q_prescaled=pre)
        ops.set_tuning("attn_tail", 0)
        try:
            plain = ops.attention_fwd(qq, k, vt, H, q_prescaled=pre)
        finally:
            ops.set_tuning("attn_tail", 1)
        assert rel_l2(out.float(), plain.float()) < 3e-3, (B, H, Lq, Lk, pre)
        rows = torch.cat([torch.arange(0, 16), torch.arange(Lq - 700, Lq, 7)]).to(DEV)
        qe = (qq[:, rows].float() / ops.q_prescale(128)) if pre else q[:, rows].float()
        qf = qe.view(B, -1, H, 128).transpose(1, 2)
        kf = k.float().view(B, Lk, H, 128).transpose(1, 2)
        vf = v.float().view(B, Lk, H, 128).transpose(1, 2)
        ref = (torch.softmax(qf @ kf.transpose(-1, -2) / math.sqrt(128), dim=-1) @ vf).transpose(1, 2).reshape(B, -1, C)
        assert rel_l2(out[:, rows].float(), ref) < 6e-3, (B, H, Lq, Lk, pre)
    assert split_seen >= 3, split_seen
