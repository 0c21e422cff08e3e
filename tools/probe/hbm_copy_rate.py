"""What a plain device copy reaches on this chip at the row kernels' sizes -- the yardstick for the HBM-bound kernels.
Read : write = 1 : 1 (copy, the RMSNorm+RoPE kernel's mix), 2 : 1 (fp32 -> bf16 cast, LN-modulate's mix), read only (sum)."""
import time

import torch

dev = torch.device("cuda:0")
L, C = 67080, 5120


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


a = torch.randn(L, 2 * C, device=dev).bfloat16()
b = torch.empty_like(a)
x = torch.randn(L, C, device=dev)
y = torch.empty(L, C, device=dev, dtype=torch.bfloat16)
for rep in range(2):
    t = timed(lambda: b.copy_(a))
    print(f"bf16 copy  {a.numel() * 2 / 1e9:.2f} GB in + out: {t * 1e3:.3f} ms  {2 * a.numel() * 2 / t / 1e12:.2f} TB/s (1:1)")
    t = timed(lambda: y.copy_(x))
    print(f"fp32->bf16 {x.numel() * 4 / 1e9:.2f} GB in, {y.numel() * 2 / 1e9:.2f} out: {t * 1e3:.3f} ms  {x.numel() * 6 / t / 1e12:.2f} TB/s (2:1)")
    t = timed(lambda: torch.sum(x))
    print(f"fp32 sum   {x.numel() * 4 / 1e9:.2f} GB in: {t * 1e3:.3f} ms  {x.numel() * 4 / t / 1e12:.2f} TB/s (read only)")
    t = timed(lambda: y.zero_())
    print(f"bf16 fill  {y.numel() * 2 / 1e9:.2f} GB out: {t * 1e3:.3f} ms  {y.numel() * 2 / t / 1e12:.2f} TB/s (write only)")
