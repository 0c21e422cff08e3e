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
        m.sp_head_groups = 1                             # one exchange / one attention launch per layer instead of head groups
        one_group = m(lat, t, ctx, 420, **kw)
        torch.cuda.synchronize()
        rel = float((sharded - single).norm() / single.norm())
        # heads are independent and no split tail at this size: the head-group pipeline must not change a bit
        assert torch.equal(one_group, sharded) or heads // world < 2
        q_out.put((rank, rel, float(single.abs().mean())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,heads", [(2, 4), (4, 4), (8, 8), (2, 3), (4, 6), (8, 12)])
def test_sp_forward_equals_single_device(world, heads):
    """(2, 3), (4, 6), (8, 12): num_heads % world != 0 -- the 12-head 1.3B model on 8 GPUs, where the reference would reach for
    ring_degree (dist/fuser.py:46-49).  Here the heads are padded to a multiple of the degree and dealt round-robin by re-arranged, zero
    -padded weight copies (WanTransformer3DModel._pad_heads_for_ulysses): the same kernels and equal-split exchanges, a dummy head on
    the ranks that are one short."""
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
    # (padded heads: the q / k / v / o products run at another width -- C_pad -- than the single-device forward's, i.e. possibly on
    # another GEMM form with another fp32 summation order, and the RMSNorm gains carry the sqrt(C / C_pad) factor: one more layer of
    # bf16 rounding flips, measured 2.1e-3 at 12 heads over 8 ranks)
    bound = 3e-3 if heads % world else 2e-3
    assert all(r[1] < bound for r in res), res
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


def _rccl_ws1_worker(port, q_out):
    """One rank, backend nccl (= RCCL), force_ulysses: every collective of the Ulysses branch is a real RCCL call on the
    process group's own stream, issued async and waited for on the compute stream."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from videocof_amd import WanTransformer3DModel
        from videocof_amd import dist as vdist
        from videocof_amd.weights import deterministic_dit_state_dict, det_uniform
        heads, layers = 4, 6
        cfgd = dict(dim=128 * heads, ffn_dim=1024, num_layers=layers, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
        m = WanTransformer3DModel(dim=128 * heads, ffn_dim=1024, num_heads=heads, num_layers=layers, text_dim=64)
        m.load_state_dict(deterministic_dit_state_dict(**cfgd), device="cuda:0")
        lat = det_uniform("sp.lat", (2, 16, 7, 12, 20), 1.0).cuda()
        ctx = [det_uniform("sp.c0", (37, 64), 1.0).cuda(), det_uniform("sp.c1", (5, 64), 1.0).cuda()]
        t = torch.tensor([749, 749], device="cuda:0")
        kw = dict(frame_split_indices=[3, 3], ground_frame_indices=[(3, 4), (3, 4)])
        single = m(lat, t, ctx, 420, **kw)
        vdist.init_sequence_parallel()
        m.enable_multi_gpus_inference()
        m.force_ulysses = True
        assert m.sp_world_size == 1 and not vdist.get_sp_group()._host_staged
        comm = m._comm_events = []
        sharded = m(lat, t, ctx, 420, **kw)               # 420 -> 424 rows (multiple of 8): also the padded-key masking
        wb = m._bufs[m._bufs_last]
        assert m._usp and wb.vt is None and wb.kw_s is not None     # really the wire-buffer branch
        again = m(lat, t, ctx, 420, **kw)                 # persistent wire buffers reused: a reuse hazard shows up here
        m.sp_head_groups = 1                              # (4 local heads: the default ran 2 | 2 head groups)
        m._comm_events = None
        one_group = m(lat, t, ctx, 420, **kw)
        assert torch.equal(one_group, sharded), "head-group pipelining changed the result"
        m.sp_head_groups = 2
        m.use_block_composite = False                     # the token-local part of every layer op by op instead of wan_dit_block_tail_forward
        per_op = m(lat, t, ctx, 420, **kw)
        assert torch.equal(per_op, sharded), "the tail composite is not the per-op launch sequence"
        m.use_block_composite = True
        m._comm_events = comm
        lat2 = det_uniform("sp.lat2", (2, 16, 7, 12, 20), 1.0).cuda()
        other = m(lat2, t, ctx, 420, **kw)                # other data through the same buffers ...
        third = m(lat, t, ctx, 420, **kw)                 # ... and back
        torch.cuda.synchronize()
        n_ev = len(comm)
        ms = sum(a.elapsed_time(b) for _, a, b in comm)
        assert {tag for tag, _, _ in comm} == {"q_g0", "o_g1", "all_gather"}
        from videocof_amd import GraphedForward
        try:                                              # this transport's collectives run on the process group's own stream: not captured
            GraphedForward(m)
            raise AssertionError("graph capture over the torch.distributed transport must be refused")
        except NotImplementedError:
            pass
        m.force_ulysses = False
        m._comm_events = None
        plain_again = m(lat, t, ctx, 420, **kw)
        torch.cuda.synchronize()
        q_out.put((float((sharded - single).norm() / single.norm()), bool(torch.equal(sharded, single)),
                   bool(torch.equal(again, sharded)), bool(torch.equal(third, sharded)), bool(torch.equal(other, sharded)),
                   bool(torch.equal(plain_again, single)), n_ev, ms, layers))
    finally:
        dist.destroy_process_group()


def test_sp_async_path_over_rccl_on_one_gpu():
    """The Ulysses branch on its REAL transport with one rank: backend "nccl" = RCCL, device buffers, `all_to_all_single(async_op=True)`
    on the group's stream, every wait_*() ordering against the compute stream, the persistent wire buffers -- the code that
    the 2-GPU test below covers but that a 1-GPU box otherwise never executes.  With P = 1 the exchanges are identities, so
    the result must equal the plain forward (different GEMM split: q and k projected separately) and repeat bitwise."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_ws1_worker, args=(_free_port(), q))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    rel, bitwise_single, again, third, other_equal, plain_ok, n_ev, ms, layers = q.get(timeout=5)
    assert rel < 2e-3, rel
    assert again and third, "the Ulysses forward does not repeat bitwise: a wire buffer is reused before its consumer finished"
    assert not other_equal and plain_ok
    assert n_ev == (2 * layers + 1) * 4                  # 4 forwards x ((waits before attention + inverse exchange) per layer + the all-gather)
    print(f"rccl world_size=1: rel={rel:.2e} bitwise_vs_single={bitwise_single} exposed_comm={ms:.2f} ms over {n_ev} windows")


def _fp8_attn_sp_worker(port, q_out):
    """The fp8 attention options under the Ulysses branch (RCCL, one rank, force_ulysses): the e4m3 q / k (K smoothing over the
    arrived tokens), the MX V^T and both fp8 kernels run on the arrived wire operands."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from videocof_amd import WanTransformer3DModel, ops
        from videocof_amd import dist as vdist
        from videocof_amd.weights import deterministic_dit_state_dict, det_uniform
        heads, layers = 4, 3
        cfgd = dict(dim=128 * heads, ffn_dim=1024, num_layers=layers, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
        m = WanTransformer3DModel(dim=128 * heads, ffn_dim=1024, num_heads=heads, num_layers=layers, text_dim=64)
        m.load_state_dict(deterministic_dit_state_dict(**cfgd), device="cuda:0")
        # the CFG batch of BASELINE configs[3] (two samples, two prompts): the token-major wire [P*Ll][B][Cl] is quantised as one matrix
        lat = det_uniform("sp.lat", (2, 16, 7, 12, 20), 1.0).cuda()
        ctx = [det_uniform("sp.c0", (37, 64), 1.0).cuda(), det_uniform("sp.c1", (5, 64), 1.0).cuda()]
        t = torch.tensor([749, 749], device="cuda:0")
        kw = dict(frame_split_indices=[3, 3], ground_frame_indices=[(3, 4), (3, 4)])
        bf16 = m(lat, t, ctx, 420, **kw)
        vdist.init_sequence_parallel()
        m.enable_multi_gpus_inference()
        res = []
        for layers_ in (("attn",), ("attn", "attn_pv")):
            m.enable_fp8_linear(layers_)                         # (fresh per-layer exponents: calibrated on the next forward)
            m.force_ulysses = False
            single = m(lat, t, ctx, 420, **kw)
            exp_single = [tuple(b.f8["attn_exp"]) for b in m.blocks]
            m.enable_fp8_linear(layers_)                         # forget them: the Ulysses branch calibrates for itself
            m.force_ulysses = True
            assert m.sp_head_groups == 2                         # 4 local heads as 2 | 2: q groups quantised on arrival
            sharded = m(lat, t, ctx, 420, **kw)
            exp_sp = [tuple(b.f8["attn_exp"]) for b in m.blocks]
            variant = ops.get_tuning("last_attn_variant")        # the last attention call of a forward is the bf16 cross-attention
            again = m(lat, t, ctx, 420, **kw)
            m.sp_head_groups = 1
            one_group = m(lat, t, ctx, 420, **kw)
            m.sp_head_groups = 2
            wb = m._bufs[m._bufs_last]
            used = m._usp and wb.vt is None and hasattr(wb, "q8") and (("attn_pv" not in layers_) or hasattr(wb, "v8"))
            torch.cuda.synchronize()
            res.append((float((sharded - single).norm() / single.norm()), float((sharded - bf16).norm() / bf16.norm()),
                        bool(torch.equal(again, sharded)), bool(used), bool(torch.equal(sharded, bf16)), variant,
                        exp_single == exp_sp, float((one_group - sharded).norm() / sharded.norm())))
        q_out.put(res)
    finally:
        dist.destroy_process_group()


def test_fp8_attention_options_under_ulysses():
    """`enable_fp8_linear(("attn",))` / `(("attn", "attn_pv"))` with sequence parallelism on, a CFG batch of two samples, per-layer
    CALIBRATED exponents and head-group pipelining (2 | 2): same result as the single-device fp8 forward up to the different q | k
    projection split, the same calibrated exponents, repeatable bitwise, and really different from the bf16 forward."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_fp8_attn_sp_worker, args=(_free_port(), q))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    for rel_single, rel_bf16, again, used, same_as_bf16, _, same_exponents, rel_groups in q.get(timeout=5):
        assert used and again and not same_as_bf16
        assert same_exponents            # the Ulysses branch calibrates the per-layer exponents a single device measures
        assert rel_single < 5e-3 and 0 < rel_bf16 < 3e-2, (rel_single, rel_bf16)
        assert rel_groups < 2e-3         # head groups: the same operands quantised per group (heads are independent)


def _fp8_attn_gloo_worker(rank, world, port, q_out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from videocof_amd import WanTransformer3DModel
        from videocof_amd import dist as vdist
        from videocof_amd.weights import deterministic_dit_state_dict, det_uniform
        heads = 8
        cfgd = dict(dim=128 * heads, ffn_dim=1024, num_layers=2, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
        m = WanTransformer3DModel(dim=128 * heads, ffn_dim=1024, num_heads=heads, num_layers=2, text_dim=64)
        sd = deterministic_dit_state_dict(**cfgd)
        # head-dependent q gains: the largest |q| sits in ONE rank's heads, so the ranks only agree on the exponents through the reduction
        for i in range(2):
            sd[f"blocks.{i}.self_attn.norm_q.weight"][5 * 128:6 * 128] *= 6.0
        m.load_state_dict(sd, device="cuda:0")
        lat = det_uniform("sp.lat", (2, 16, 7, 12, 20), 1.0).cuda()
        ctx = [det_uniform("sp.c0", (37, 64), 1.0).cuda(), det_uniform("sp.c1", (5, 64), 1.0).cuda()]
        t = torch.tensor([749, 749], device="cuda:0")
        kw = dict(frame_split_indices=[3, 3], ground_frame_indices=[(3, 4), (3, 4)])
        m.enable_fp8_linear(("attn", "attn_pv"))
        single = m(lat, t, ctx, 420, **kw)
        exp_single = [tuple(b.f8["attn_exp"]) for b in m.blocks]
        m.enable_fp8_linear(("attn", "attn_pv"))
        vdist.init_sequence_parallel()
        m.enable_multi_gpus_inference()
        sharded = m(lat, t, ctx, 420, **kw)
        exp_sp = [tuple(b.f8["attn_exp"]) for b in m.blocks]
        torch.cuda.synchronize()
        q_out.put((rank, float((sharded - single).norm() / single.norm()), exp_single == exp_sp, exp_sp != [(5, 2)] * 2))
    finally:
        dist.destroy_process_group()


def test_fp8_attention_sharded_over_two_ranks_equals_single_device():
    """All-fp8 attention with the heads REALLY split (2 ranks, gloo, shared GPU; 8 heads = 4 local as 2 | 2 head groups; a CFG batch of
    two): each rank quantises its arrived heads (K mean over all tokens, MX V^T per group) with per-layer exponents CALIBRATED on the
    first forward and agreed through an all-reduce(max) -- one rank's heads carry a x6 q gain, so without the reduction the ranks
    would differ -- same exponents and same numbers as the single-device fp8 forward up to the projection split."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fp8_attn_gloo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert all(r[1] < 5e-3 for r in res), res
    assert len({round(r[1], 9) for r in res}) == 1
    assert all(r[2] and r[3] for r in res), res          # calibrated (not the static pair), and equal to the single device's


def _library_comm_worker(q_out):
    """The Ulysses branch over the communicator the LIBRARY owns (wan_sp_*: RCCL bound by dlopen, one side stream, two events),
    one rank, no torch.distributed at all."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    from videocof_amd import WanTransformer3DModel
    from videocof_amd import dist as vdist
    from videocof_amd.weights import deterministic_dit_state_dict, det_uniform
    heads, layers = 4, 4
    cfgd = dict(dim=128 * heads, ffn_dim=1024, num_layers=layers, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
    m = WanTransformer3DModel(dim=128 * heads, ffn_dim=1024, num_heads=heads, num_layers=layers, text_dim=64)
    m.load_state_dict(deterministic_dit_state_dict(**cfgd), device="cuda:0")
    lat = det_uniform("sp.lat", (2, 16, 7, 12, 20), 1.0).cuda()
    ctx = [det_uniform("sp.c0", (37, 64), 1.0).cuda(), det_uniform("sp.c1", (5, 64), 1.0).cuda()]
    t = torch.tensor([749, 749], device="cuda:0")
    kw = dict(frame_split_indices=[3, 3], ground_frame_indices=[(3, 4), (3, 4)])
    single = m(lat, t, ctx, 420, **kw)
    comm = vdist.init_sequence_parallel(backend="library", rank=0, world_size=1)
    assert isinstance(comm, vdist.LibraryComm) and comm.world_size == 1 and not torch.distributed.is_initialized()
    m.enable_multi_gpus_inference()
    m.force_ulysses = True
    outs = [m(lat, t, ctx, 420, **kw) for _ in range(3)]
    # the raw calls: an exchange of a known pattern on a side stream behind a compute-stream producer, and the error paths
    a = torch.arange(1 << 20, device="cuda:0", dtype=torch.int32)
    send = a * 3                                             # producer on the compute stream, right before the exchange
    recv = torch.zeros_like(send)
    wait = comm.exchange(recv, send, async_op=True)
    wait()
    ok_raw = bool(torch.equal(recv, a * 3))
    gathered = comm.all_gather_tokens(torch.ones(2, 5, 3, device="cuda:0"))
    try:
        comm.exchange(send, send)
        inplace_error = ""
    except ValueError as e:
        inplace_error = str(e)
    torch.cuda.synchronize()
    vdist.destroy_sequence_parallel()
    q_out.put((float((outs[0] - single).norm() / single.norm()), all(bool(torch.equal(o, outs[0])) for o in outs[1:]), ok_raw,
               tuple(gathered.shape), inplace_error))


def test_sp_over_the_library_owned_communicator():
    """`wan_sp_init`, `wan_sp_a2a_*`, one library-owned side stream + events.  One rank (RCCL refuses two ranks on one
    device), so the exchanges are identities -- what is exercised is the whole path: dlopen of RCCL, communicator set-up, the
    compute-stream -> side-stream -> compute-stream event chain around every exchange of every layer, teardown."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_library_comm_worker, args=(q,))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    rel, repeat, ok_raw, gshape, inplace_error = q.get(timeout=5)
    assert rel < 2e-3 and repeat and ok_raw and gshape == (2, 5, 3)
    assert "in-place" in inplace_error


def _sp_composition_worker(q_out):
    """fp8 Linears under the Ulysses branch (library-owned communicator, one rank)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    from videocof_amd import GraphedForward, WanTransformer3DModel
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
    bf16 = m(lat, t, ctx, 420, **kw)
    m.enable_fp8_linear(("qkv", "ffn", "o", "cross"))
    single8 = m(lat, t, ctx, 420, **kw)
    vdist.init_sequence_parallel(backend="library", rank=0, world_size=1)
    m.enable_multi_gpus_inference()
    m.force_ulysses = True
    sp8 = m(lat, t, ctx, 420, **kw)                           # fp8 projections + exchanges
    again = m(lat, t, ctx, 420, **kw)
    wb = m._bufs[m._bufs_last]
    used = bool(m._usp and wb.vt is None and hasattr(wb, "hq"))
    refused = True
    torch.cuda.synchronize()
    vdist.destroy_sequence_parallel()
    rel = lambda a, b: float((a - b).norm() / b.norm())
    q_out.put((refused, used, rel(sp8, single8), rel(sp8, bf16), bool(torch.equal(again, sp8))))


def _sp_graph_worker(q_out):
    """hipGraph capture of a sequence-parallel forward: the library communicator records its collectives on the capturing stream
    itself (csrc/sp_comm.cpp, sp_runs_inline) -- eager call, capture + replay, replays, another input through the same graph."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    from videocof_amd import GraphedForward, WanTransformer3DModel
    from videocof_amd import dist as vdist
    from videocof_amd.weights import deterministic_dit_state_dict, det_uniform
    heads, layers = 4, 3
    cfgd = dict(dim=128 * heads, ffn_dim=1024, num_layers=layers, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
    m = WanTransformer3DModel(dim=128 * heads, ffn_dim=1024, num_heads=heads, num_layers=layers, text_dim=64)
    m.load_state_dict(deterministic_dit_state_dict(**cfgd), device="cuda:0")
    lat = det_uniform("sp.lat", (2, 16, 7, 12, 20), 1.0).cuda()
    lat2 = det_uniform("sp.lat2", (2, 16, 7, 12, 20), 1.0).cuda()
    ctx = [det_uniform("sp.c0", (37, 64), 1.0).cuda(), det_uniform("sp.c1", (5, 64), 1.0).cuda()]
    t = torch.tensor([749, 749], device="cuda:0")
    kw = dict(frame_split_indices=[3, 3], ground_frame_indices=[(3, 4), (3, 4)])
    vdist.init_sequence_parallel(backend="library", rank=0, world_size=1)
    m.enable_multi_gpus_inference()
    m.force_ulysses = True
    eager, eager2 = m(lat, t, ctx, 420, **kw), m(lat2, t, ctx, 420, **kw)
    gf = GraphedForward(m)
    same = [bool(torch.equal(gf(lat, t, ctx, 420, **kw), eager)) for _ in range(3)]          # eager, capture + replay, replay
    same.append(bool(torch.equal(gf(lat2, t, ctx, 420, **kw), eager2)))
    replays = gf.replays
    gf.reset()                                               # graphs go before the communicator (RCCL's teardown waits for them)
    del gf
    torch.cuda.synchronize()
    vdist.destroy_sequence_parallel()
    q_out.put((same, replays))


def test_graph_capture_of_a_sequence_parallel_forward():
    """Round 4 could not record an SP forward (hipStreamEndCapture segfaulted with RCCL on a forked side stream).  With the
    collectives on the capturing stream itself the whole Ulysses forward -- wire-layout kernels, head-group exchanges, all-gather --
    replays from ONE hipGraph, bit-identical to the eager forward, also for another input.  The torch.distributed transport stays
    refused (its collectives run on the process group's own stream)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_sp_graph_worker, args=(q,))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    same, replays = q.get(timeout=5)
    assert all(same) and replays == 3


def test_fp8_linears_compose_with_ulysses():
    """fp8 projections under sequence parallelism (q, k and V^T projected from the same e4m3 token rows, weight copies split by output
    rows) == the single-device fp8 forward up to the projection split, bitwise repeatable, really different from bf16."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_sp_composition_worker, args=(q,))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    refused, used, rel_single, rel_bf16, again = q.get(timeout=5)
    assert refused and used and again
    assert rel_single < 5e-3 and 1e-4 < rel_bf16 < 3e-2, (rel_single, rel_bf16)


def _shard_shape_worker(rank, world, port, q_out):
    """14B width (40 heads -> 5 per rank at P = 8: no XCD pinning, split-KV tail round on), L = 16 384."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from videocof_amd import WanTransformer3DModel, ops, _lib
        from videocof_amd import dist as vdist
        from videocof_amd.weights import random_dit_state_dict
        dev = torch.device("cuda", 0)
        shapes = dict(dim=5120, ffn_dim=13824, num_layers=1)
        m = WanTransformer3DModel(dim=5120, ffn_dim=13824, num_heads=40, num_layers=1)
        m.load_state_dict(random_dit_state_dict(dev, seed=3, exercise_epilogues=True, **shapes), device=dev)
        g = torch.Generator(device=dev).manual_seed(5)
        lat = torch.randn(1, 16, 16, 64, 64, device=dev, generator=g)                     # grid (16, 32, 32) = 16 384 tokens; fp32 in -> fp32 out
        ctx = [torch.randn(37, 4096, device=dev, generator=g).bfloat16()]
        t = torch.tensor([500], device=dev)
        kw = dict(frame_split_indices=[7], ground_frame_indices=[(7, 8)])
        single = m(lat, t, ctx, 16384, **kw)
        vdist.init_sequence_parallel()
        m.enable_multi_gpus_inference()
        ev = m._attn_events = []
        grouped = m(lat, t, ctx, 16384, **kw)            # default: head groups 2 | 3 (128 + 192 workgroups: no tail round)
        m.sp_head_groups = 1                             # all 5 local heads in one launch: 320 workgroups, the split-KV tail round runs
        sharded = m(lat, t, ctx, 16384, **kw)
        torch.cuda.synchronize()
        variant = int(m._last_attn_variant)
        planned = int(_lib.load().wan_attention_workspace_bytes(1, 16384, 16384, 40 // world, 128))
        rel = lambda a: float((a.float() - single.float()).norm() / single.float().norm())
        q_out.put((rank, max(rel(sharded), rel(grouped)), variant, planned))
    finally:
        dist.destroy_process_group()


def test_sp_equals_single_device_at_the_14b_shard_shape():
    """SP == single device where the shard arithmetic of the headline configuration is live: 40 heads over 8 ranks = 5 local
    heads (nbh % 8 != 0 -> no XCD pinning), 64 query blocks x 5 heads = 320 workgroups = 256 + 64 -> the split-KV tail round
    runs, 14B width (C = 5120, ffn 13 824, K = 5120 / 13 824 GEMM dispatch), non-zero biases and non-unit gains."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_shape_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    from videocof_amd import _lib
    assert all(r[1] < 3e-3 for r in res), res
    assert len({round(r[1], 9) for r in res}) == 1
    assert all(r[2] & _lib.ATTN_VARIANT_SPLIT_TAIL and not (r[2] & _lib.ATTN_VARIANT_XCD_PINNED) for r in res), res
    assert all(r[3] > 4096 for r in res)             # the tail plan asked for scratch


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


def test_bench_line_under_sequence_parallelism_is_self_validating():
    """`bench.py --gpus 2` (two ranks on this GPU, host-staged gloo exchanges: the N > 1 code path of the bench, its numbers
    meaningless): the line must CHECK the sharded forward it timed -- `parity` = the ranks' probes gathered, the last block + head
    + unpatchify against the oracle on rank 0 -- and carry the per-rank wall clocks and the exposed-communication split per exchange."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                        "--workload", "1.3b-small", "--layers", "3", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["config"]["parallelism"] == "ulysses-sp2"
    assert line["config"]["layers_override"] == 3
    p = line["parity"]
    assert p is not None and p.get("ok") is True, p
    assert p["rel_l2"] < 1e-2 and p["cosine"] > 0.9999 and "sequence-parallel forward over 2 ranks" in p["what"]
    rw = line["rank_wall_s"]
    assert len(rw["per_rank"]) == 2 and rw["max_over_min"] >= 1.0
    assert abs(max(rw["per_rank"]) * 1e3 / line["steps"] - line["ms_per_step"]) < 0.5          # the line's time is the slowest rank's
    ec = line["exposed_comm_ms"]
    assert set(ec["per_step_by_exchange"]) == {"q_g0", "o_g1", "all_gather"}
    assert abs(sum(ec["per_step_by_exchange"].values()) - ec["per_step"]) < 0.01
    assert line["box"] is not None and line["box"]["mfma_mix_tflops"] > 100 and line["value_normalised"] > 0


def test_bench_sequence_parallel_line_over_rccl_with_one_rank():
    """`bench.py --force-sp`: the N > 1 code path of the bench on its REAL transport -- a 1-rank RCCL process group, model.force_ulysses:
    every exchange an async RCCL all-to-all on the group's stream, the parity probe gathered through the group, walls through all_gather
    on device tensors, barriers, teardown -- everything a multi-GPU run executes except bytes crossing between devices."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-sp", "--workload", "1.3b-small", "--layers", "3",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and "RCCL" in line["backend"] and "force_ulysses" in line["backend"]
    assert line["config"]["parallelism"] == "ulysses-sp1"
    assert line["parity"]["ok"] is True and line["parity"]["rel_l2"] < 1e-2
    assert set(line["exposed_comm_ms"]["per_step_by_exchange"]) == {"q_g0", "o_g1", "all_gather"}
    assert line["rank_wall_s"]["per_rank"] and line["rank_wall_s"]["max_over_min"] == 1.0
    assert line["roofline"]["heads_local"] == 12 and "e2e" not in line


def test_bench_cfg_and_emulated_rank_lines():
    """`bench.py --cfg S` (a step with classifier-free guidance as WanPipeline runs it: one forward over [uncond, cond], B = 2 -- BASELINE
    configs[3]'s per-step work) and `--emulate-sp P` (the PROJECTION line of one rank of P): the lines say what they are."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}

    def line(*args):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "1.3b-small", "--layers", "3", "--steps", "2", "--warmup", "1",
                            "--no-cpu-baseline", *args], capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    one, two = line(), line("--cfg", "5.0")
    assert (one["config"]["global_batch"], one["config"]["guidance_scale"]) == (1, 1.0)
    assert (two["config"]["global_batch"], two["config"]["guidance_scale"]) == (2, 5.0) and two["parity"]["ok"] is True
    assert two["roofline"]["flop_per_launch"] == 2 * one["roofline"]["flop_per_launch"] and "e2e" not in two
    emu = line("--emulate-sp", "4", "--cfg", "5.0")
    assert emu["metric"].startswith("PROJECTION") and emu["projection"]["of_n_gpus"] == 4 and "skipped" in emu["parity"]
    assert emu["config"]["parallelism"].startswith("EMULATED rank 0 of ulysses-sp4") and emu["roofline"]["heads_local"] == 3 and emu["n_gpus"] == 1
    assert emu["tokens_per_s_per_gpu"] == round(emu["value"] / 4, 1) and emu["cpu_baseline"] is None


def test_emulated_rank_runs_the_shard_shapes_and_degree_one_is_the_single_device():
    """`bench.py --emulate-sp P` (videocof_amd.dist.EmulatedRank): ONE rank's work of a P-way group with device-local copies for the
    exchanges.  With P = 1 the copies ARE the exchanges of a one-rank group, so the forward must equal the single-device one
    (to the tolerance of the SP == single tests); with P = 2 the rank sees its own slabs where its peer's would arrive -- the output
    is not a latent, but it is finite, complete, and the attention ran on H / P heads over the padded full sequence."""
    from videocof_amd import WanTransformer3DModel
    from videocof_amd import dist as vdist
    from videocof_amd.weights import deterministic_dit_state_dict, det_uniform
    heads = 4
    cfgd = dict(dim=128 * heads, ffn_dim=1024, num_layers=2, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
    m = WanTransformer3DModel(dim=128 * heads, ffn_dim=1024, num_heads=heads, num_layers=2, text_dim=64)
    m.load_state_dict(deterministic_dit_state_dict(**cfgd), device="cuda:0")
    lat = det_uniform("sp.lat", (1, 16, 7, 12, 20), 1.0).cuda()
    ctx = [det_uniform("sp.c0", (37, 64), 1.0).cuda()]
    t = torch.tensor([749], device="cuda:0")
    kw = dict(frame_split_indices=[3], ground_frame_indices=[(3, 4)])
    single = m(lat, t, ctx, 420, **kw)
    try:
        vdist.init_sequence_parallel(backend="emulated", rank=0, world_size=1)
        m.enable_multi_gpus_inference()
        m.force_ulysses = True
        one = m(lat, t, ctx, 420, **kw)
        assert float((one - single).norm() / single.norm()) < 2e-3
        vdist.init_sequence_parallel(backend="emulated", rank=0, world_size=2)
        m.enable_multi_gpus_inference()
        m.force_ulysses = False
        ev = m._attn_events = []
        two = m(lat, t, ctx, 420, **kw)
        torch.cuda.synchronize()
        assert two.shape == single.shape and torch.isfinite(two.float()).all()
        assert m.sp_world_size == 2 and m._last_attn_rows == 432 and len(ev) == 2        # 420 -> a multiple of 8 * P; one event pair per layer
        assert float((two - single).norm() / single.norm()) > 1e-2                        # (its own slabs, not the peer's: another function)
    finally:
        m._attn_events = None
        vdist.destroy_sequence_parallel()
