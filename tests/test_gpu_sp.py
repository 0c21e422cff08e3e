"""-m gpu: the sequence-parallel (Ulysses) forward of the HIP-backed DiT equals the single-device
forward (SURVEY.md section 8a note a21: SP has no runnable reference, its oracle is SP == single).
Two ranks share cuda:0 and exchange through gloo with host staging -- the exchange layer's
RCCL path differs only in the transport; shard arithmetic, RoPE token offsets, padded-key
masking and the final all-gather are exactly the multi-GPU code."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q_out, heads):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from videocof_amd import WanTransformer3DModel
        from videocof_amd import dist as vdist
        from videocof_amd.weights import deterministic_dit_state_dict, det_uniform
        cfgd = dict(dim=128 * heads, ffn_dim=1024, num_layers=2, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
        sd = deterministic_dit_state_dict(**cfgd)
        m = WanTransformer3DModel(dim=128 * heads, ffn_dim=1024, num_heads=heads, num_layers=2, text_dim=64)
        m.load_state_dict(sd, device="cuda:0")
        lat = det_uniform("sp.lat", (2, 16, 7, 12, 20), 1.0).cuda()
        ctx = [det_uniform("sp.c0", (37, 64), 1.0).cuda(), det_uniform("sp.c1", (5, 64), 1.0).cuda()]
        t = torch.tensor([749, 749], device="cuda:0")
        kw = dict(frame_split_indices=[3, 3], ground_frame_indices=[(3, 4), (3, 4)])
        single = m(lat, t, ctx, 420, **kw)
        vdist.init_sequence_parallel()
        m.enable_multi_gpus_inference()
        assert m.sp_world_size == world and m.sp_world_rank == rank
        sharded = m(lat, t, ctx, 420, **kw)              # 420 tokens -> padded to a multiple of 8 * world
        torch.cuda.synchronize()
        rel = float((sharded - single).norm() / single.norm())
        q_out.put((rank, rel, float(single.abs().mean())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,heads", [(2, 4), (4, 4), (8, 8)])
def test_sp_forward_equals_single_device(world, heads):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, heads)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    # same kernels, same bf16 roundings except the attention partition (heads instead of all) and
    # the padded-row bookkeeping: results agree to bf16 noise, identically on every rank
    assert all(r[1] < 2e-3 for r in res), res
    assert len({round(r[1], 9) for r in res}) == 1
