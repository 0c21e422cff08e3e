import sys, time, torch
sys.path.insert(0, "/root/repo")
from videocof_amd import ops
dev = torch.device("cuda:0")
L, C = 67080, 5120
x = torch.randn(L, C, device=dev); sc = torch.randn(1, C, device=dev); sh = torch.randn(1, C, device=dev)
o = torch.empty(L, C, device=dev, dtype=torch.bfloat16)
def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for wave in (4, 2, 1):
    ops.set_tuning("row_group", wave)
    for rep in range(2):
        a = timed(lambda: ops.ln_modulate(x, sc, sh, True, L, 1e-6, out=o))
        b = timed(lambda: ops.ln_modulate(x, None, None, False, L, 1e-6, out=o))
        c = timed(lambda: o.copy_(x))
        print(f"row_group={wave}: ln_modulate with scale/shift {a*1e3:.3f} ms {6*L*C/a/1e12:.2f} TB/s | without {b*1e3:.3f} ms {6*L*C/b/1e12:.2f} TB/s | torch cast {c*1e3:.3f} ms {6*L*C/c/1e12:.2f} TB/s", flush=True)
