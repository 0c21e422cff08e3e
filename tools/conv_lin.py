import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videocof_amd import ops
DEV='cuda:0'
g = torch.Generator(device=DEV).manual_seed(0)
def run(cin, cout, T, H, W, mode):
    ops.set_tuning("conv_patch", mode)
    K = 27*cin; Kp = ops.round_up(K, 64)
    x = torch.randn(T,H,W,cin, device=DEV, generator=g).bfloat16()
    hist = torch.randn(2,H,W,cin, device=DEV, generator=g).bfloat16()
    w = torch.zeros(cout, Kp, device=DEV, dtype=torch.bfloat16); w[:, :K] = (torch.randn(cout,K,device=DEV,generator=g)*0.02).bfloat16()
    b = torch.zeros(cout, device=DEV)
    f = lambda: ops.conv_cl(x, w, b, cout, (3,3,3), pad=(2,1,1), out_thw=(T,H,W), hist=hist)
    f(); torch.cuda.synchronize()
    t0=time.perf_counter(); n=3 if '--profile' in sys.argv else 20
    for _ in range(n): f()
    torch.cuda.synchronize()
    dt=(time.perf_counter()-t0)/n
    return dt, 2.0*T*H*W*cout*K/dt/1e12
if "--profile" in sys.argv:          # one shape, both kernels, few launches (rocprofv3 counter passes)
    for mode in (2, 0):
        run(96, 96, 4, 480, 832, mode)
    sys.exit(0)
for cin in (32, 64, 96, 192, 384):
    for mode in (2, 0):
        dt, tf = run(cin, 96, 4, 480, 832, mode)
        print(f"Cin={cin:3d} Cout=96 4x480x832 mode={mode}: {dt*1e3:7.3f} ms {tf:7.1f} TF")
for T,H,W in ((1,480,832),(2,480,832),(8,480,832)):
    dt, tf = run(96, 96, T, H, W, 2)
    print(f"Cin=96 Cout=96 {T}x{H}x{W} mode=2: {dt*1e3:7.3f} ms {tf:7.1f} TF")
