"""Does the persistent GEMM's epilogue wait for its bias / gate loads?  The same products with and without the per-column vectors."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from videocof_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = 67080
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for (N, K, epi, name) in ((5120, 5120, ops.EPI_BF16, "bf16"), (13824, 5120, ops.EPI_GELU_BF16, "gelu"), (5120, 5120, ops.EPI_RESID_F32, "resid")):
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev, generator=g)
    gate = torch.randn(1, N, device=dev, generator=g)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == ops.EPI_RESID_F32 else torch.bfloat16)
    fl = 2.0 * M * N * K
    for rep in range(2):
        arms = [("bias", dict(bias=bias)), ("no bias", dict(bias=None))]
        if epi == ops.EPI_RESID_F32:
            arms = [("bias+gate", dict(bias=bias, gate=gate)), ("bias", dict(bias=bias)), ("nothing", dict(bias=None))]
        line = []
        for nm, kw in arms:
            gt = kw.get("gate")
            t = timed(lambda: ops.gemm(a, w, kw["bias"], epi, out=out, gate=gt, rows_per_batch=M if gt is not None else 0))
            line.append(f"{nm} {t * 1e3:.3f} ms {fl / t / 1e12:.0f} TF/s")
        print(f"{name:6s} N={N} K={K}: " + " | ".join(line), flush=True)
