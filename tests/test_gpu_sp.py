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


def _nccl_worker(rank, world, port, q_out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = str(rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from videocof_amd import WanTransformer3DModel
        from videocof_amd import dist as vdist
        from videocof_amd.weights import deterministic_dit_state_dict, det_uniform
        heads = 4
        cfgd = dict(dim=128 * heads, ffn_dim=1024, num_layers=2, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
        m = WanTransformer3DModel(dim=128 * heads, ffn_dim=1024, num_heads=heads, num_layers=2, text_dim=64)
        m.load_state_dict(deterministic_dit_state_dict(**cfgd), device=f"cuda:{rank}")
        lat = det_uniform("sp.lat", (2, 16, 7, 12, 20), 1.0).cuda()
        ctx = [det_uniform("sp.c0", (37, 64), 1.0).cuda(), det_uniform("sp.c1", (5, 64), 1.0).cuda()]
        t = torch.tensor([749, 749], device=f"cuda:{rank}")
        kw = dict(frame_split_indices=[3, 3], ground_frame_indices=[(3, 4), (3, 4)])
        single = m(lat, t, ctx, 420, **kw)
        vdist.init_sequence_parallel()
        m.enable_multi_gpus_inference()
        sharded = m(lat, t, ctx, 420, **kw)              # RCCL all_to_all_single over xGMI, async_op on the group's stream
        again = m(lat, t, ctx, 420, **kw)
        torch.cuda.synchronize()
        q_out.put((rank, float((sharded - single).norm() / single.norm()), bool(torch.equal(again, sharded))))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL transport needs two GPUs (the development boxes have one)")
def test_sp_forward_over_rccl_equals_single_device():
    """The same check through the REAL transport -- backend "nccl" = RCCL, device buffers exchanged directly, async
    collectives on the process group's stream -- whenever two GPUs are visible."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert all(r[1] < 2e-3 and r[2] for r in res), res


def _alloc_worker(rank, world, port, q_out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from videocof_amd import WanTransformer3DModel
        from videocof_amd import dist as vdist
        from videocof_amd.weights import deterministic_dit_state_dict, det_uniform
        counts = []
        for layers in (2, 6):
            cfgd = dict(dim=512, ffn_dim=1024, num_layers=layers, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
            m = WanTransformer3DModel(dim=512, ffn_dim=1024, num_heads=4, num_layers=layers, text_dim=64)
            m.load_state_dict(deterministic_dit_state_dict(**cfgd), device="cuda:0")
            vdist.init_sequence_parallel()
            m.enable_multi_gpus_inference()
            m.cache_context = True
            lat = det_uniform("sp.lat", (1, 16, 7, 12, 20), 1.0).cuda()
            ctx = [det_uniform("sp.c0", (37, 64), 1.0).cuda()]
            t = torch.tensor([749], device="cuda:0")
            m(lat, t, ctx, 420)                                               # allocates the persistent buffers
            torch.cuda.synchronize()
            before = torch.cuda.memory_stats()["allocation.all.allocated"]
            m(lat, t, ctx, 420)
            torch.cuda.synchronize()
            counts.append(torch.cuda.memory_stats()["allocation.all.allocated"] - before)
        q_out.put((rank, counts))
    finally:
        dist.destroy_process_group()


def test_sp_layers_allocate_nothing():
    """The per-layer Ulysses path works in persistent wire buffers: the device allocations of a forward do not grow with
    the layer count (2 vs 6 layers), i.e. no per-layer torch.zeros / .contiguous() re-layout copies."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_alloc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    for rank, (c2, c6) in sorted(q.get(timeout=5) for _ in range(2)):
        assert c6 == c2, (rank, c2, c6)
