"""What splitting a rank's self-attention into two head-group launches costs (Ulysses head-group pipelining):
the arrived wire operands of an 8 / 4 / 2-way shard at L = 67 080 (5 / 10 / 20 local heads), one launch against two
(2|3, 5|5, 10|10).  Tools only."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from videocof_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
L = 67080
ld = ops.round_up(L, 64)
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n=4):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for Hl in (5, 10, 12, 20, 40):       # 12: the 1.3B model on one device; 40: the 14B model on one device (same process, same box)
    Cl = Hl * 128
    q = torch.randn(L, 1, Cl, device=dev, generator=g).bfloat16()
    k = torch.randn(L, 1, Cl, device=dev, generator=g).bfloat16()
    vt = torch.randn(1, Cl, ld, device=dev, generator=g).bfloat16()
    o = torch.empty(L, 1, Cl, device=dev, dtype=torch.bfloat16)
    bld = lambda w: w.permute(1, 0, 2)
    h1 = Hl // 2
    c1 = h1 * 128
    one = lambda: ops.attention_fwd(bld(q), bld(k), vt, Hl, k_len=L, out=bld(o), q_prescaled=True)

    def two():
        ops.attention_fwd(bld(q)[..., :c1], bld(k)[..., :c1], vt[:, :c1], h1, k_len=L, out=bld(o)[..., :c1], q_prescaled=True)
        ops.attention_fwd(bld(q)[..., c1:], bld(k)[..., c1:], vt[:, c1:], Hl - h1, k_len=L, out=bld(o)[..., c1:], q_prescaled=True)

    ref = o.clone() if one() is None else None
    one(); a = o.clone(); two(); b = o.clone()
    fl = 4.0 * L * L * Cl
    for rep in range(2):
        t1, t2 = timed(one), timed(two)
        print(f"heads/rank {Hl:2d}: one launch {t1:7.3f} ms ({fl / t1 / 1e9:5.0f} TF/s)   {h1}|{Hl - h1} {t2:7.3f} ms ({fl / t2 / 1e9:5.0f} TF/s)"
              f"   ratio {t2 / t1:.4f}   identical {bool(torch.equal(a, b))}", flush=True)
