"""Round 5 experiment: hipGraph capture of a sequence-parallel forward with the library communicator's collectives recorded on the
capturing stream itself (csrc/sp_comm.cpp: sp_runs_inline).  One rank (the development boxes have one GPU)."""
import os
import sys
import faulthandler

faulthandler.enable()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.cuda.set_device(0)
from videocof_amd import GraphedForward, WanTransformer3DModel, ops
from videocof_amd import dist as vdist
from videocof_amd.weights import deterministic_dit_state_dict, det_uniform

heads, layers = 4, 3
cfgd = dict(dim=128 * heads, ffn_dim=1024, num_layers=layers, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
m = WanTransformer3DModel(dim=128 * heads, ffn_dim=1024, num_heads=heads, num_layers=layers, text_dim=64)
m.load_state_dict(deterministic_dit_state_dict(**cfgd), device="cuda:0")
lat = det_uniform("sp.lat", (1, 16, 7, 12, 20), 1.0).cuda()
ctx = [det_uniform("sp.c0", (37, 64), 1.0).cuda()]
t = torch.tensor([749], device="cuda:0")
kw = dict(frame_split_indices=[3], ground_frame_indices=[(3, 4)])
single = m(lat, t, ctx, 420, **kw)
vdist.init_sequence_parallel(backend="library", rank=0, world_size=1)
m.enable_multi_gpus_inference()
m.force_ulysses = True
eager = m(lat, t, ctx, 420, **kw)
print("eager SP == single:", float((eager - single).norm() / single.norm()), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "inline-eager":
    ops.set_tuning("sp_inline", 1)
    e2 = m(lat, t, ctx, 420, **kw)
    print("inline eager identical:", bool(torch.equal(e2, eager)), flush=True)
gf = GraphedForward(m)
for i in range(4):
    out = gf(lat, t, ctx, 420, **kw)
    torch.cuda.synchronize()
    print(f"graph call {i}: identical {bool(torch.equal(out, eager))} replays {gf.replays}", flush=True)
lat2 = det_uniform("sp.lat2", (1, 16, 7, 12, 20), 1.0).cuda()
e3 = m(lat2, t, ctx, 420, **kw)
o3 = gf(lat2, t, ctx, 420, **kw)
torch.cuda.synchronize()
print("other input through the graph identical:", bool(torch.equal(o3, e3)), "replays", gf.replays, flush=True)
if "keep-graph" not in sys.argv:
    gf.reset()
    del gf
    torch.cuda.synchronize()
    print("graphs dropped", flush=True)
vdist.destroy_sequence_parallel()
print("communicator destroyed", flush=True)
print("done", flush=True)
