"""Ulysses exchange layer on CPU with gloo, world_size 2 (and 4): the re-layouts are exact
inverses, and 'token-sharded -> head-sharded attention -> token-sharded' equals unsharded
attention (the SP==single-device self-oracle of SURVEY.md section 8a note a21).  The attention
arithmetic inside the test is the CPU oracle's; what is under test is videocof_amd.dist."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import wan_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, L, H, q_out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from videocof_amd import dist as vdist
        dev = vdist.set_multi_gpus_devices(ulysses_degree=world)
        assert vdist.get_sequence_parallel_world_size() == world and vdist.get_sequence_parallel_rank() == rank
        sp = vdist.get_sp_group()
        torch.manual_seed(0)
        B, D = 2, 128
        C = H * D
        Lp = (L + 8 * world - 1) // (8 * world) * (8 * world)
        Ll = Lp // world
        q = torch.randn(B, Lp, C)
        k = torch.randn(B, Lp, C)
        v = torch.randn(B, Lp, C)
        sl = slice(rank * Ll, (rank + 1) * Ll)
        # --- exchange round trips
        qh = sp.scatter_heads(q[:, sl])
        assert qh.shape == (B, Lp, C // world)
        assert torch.equal(qh, q[:, :, rank * (C // world):(rank + 1) * (C // world)])
        # strided input view (q|k packed buffer) through the async path
        qk = torch.cat([q[:, sl], k[:, sl]], dim=2)
        fin = sp.scatter_heads(qk[:, :, C:], async_op=True)
        kh = fin()
        assert torch.equal(kh, k[:, :, rank * (C // world):(rank + 1) * (C // world)])
        vt_local = v[:, sl].transpose(1, 2).contiguous()                    # [B, C, Ll]
        vth = sp.scatter_heads_t(vt_local, ld=Lp + 24)
        assert vth.shape == (B, C // world, Lp + 24)
        assert torch.equal(vth[:, :, :Lp], v[:, :, rank * (C // world):(rank + 1) * (C // world)].transpose(1, 2))
        assert float(vth[:, :, Lp:].abs().max()) == 0.0
        back = sp.gather_heads(qh)
        assert torch.equal(back, q[:, sl])
        # --- sharded attention == unsharded attention (keys >= L masked)
        Hs = H // world
        o_h = torch.stack([O.attention(qh[b].view(Lp, Hs, D), kh[b].view(Lp, Hs, D),
                                       vth[b, :, :Lp].t().reshape(Lp, Hs, D), k_len=L).reshape(Lp, Hs * D)
                           for b in range(B)])
        o_local = sp.gather_heads(o_h)
        o_ref = torch.stack([O.attention(q[b].view(Lp, H, D), k[b].view(Lp, H, D), v[b].view(Lp, H, D),
                                         k_len=L).reshape(Lp, C) for b in range(B)])[:, sl]
        err = float((o_local - o_ref).abs().max())
        # --- the wire layouts the DiT uses (include/wan_hip.h a21): ONE flat all-to-all per tensor, no re-layout of q / k / o
        Cl, Lt = C // world, Lp
        for name, full in (("q", q), ("k", k)):
            send = sp.pack_heads_ref(full[:, sl])                               # [P, Ll, B, Cl] (what wan_rmsnorm_rope_sp writes)
            recv = torch.empty_like(send)
            wait = sp.exchange(recv, send, async_op=True)
            wait()
            # arrived buffer == [P*Ll][B][Cl]: all tokens, my heads, uniform strides
            got = recv.view(Lt, B, Cl).permute(1, 0, 2)
            assert torch.equal(got, full[:, :, rank * Cl:(rank + 1) * Cl]), name
        vsend = sp.pack_vt_ref(vt_local)                                        # [P, Cl, B, Ll] == [C, B, Ll]: GEMM out with ldo = B*Ll
        assert torch.equal(vsend.view(C, B, Ll)[:, 1], vt_local[1])
        vrecv = torch.empty_like(vsend)
        sp.exchange(vrecv, vsend)
        vt_full = sp.unpack_vt_ref(vrecv, Lp + 24)
        assert torch.equal(vt_full, vth)
        # attention on the arrived buffers, output written in wire form, inverse exchange, unpack
        qa = sp.pack_heads_ref(q[:, sl]); ka = sp.pack_heads_ref(k[:, sl])
        qr, kr = torch.empty_like(qa), torch.empty_like(ka)
        sp.exchange(qr, qa); sp.exchange(kr, ka)
        osend = torch.empty(Lt, B, Cl)
        for b in range(B):
            osend[:, b] = O.attention(qr.view(Lt, B, Cl)[:, b].reshape(Lt, Hs, D), kr.view(Lt, B, Cl)[:, b].reshape(Lt, Hs, D),
                                      vt_full[b, :, :Lp].t().reshape(Lt, Hs, D), k_len=L).reshape(Lt, Cl)
        orecv = torch.empty_like(osend)
        sp.exchange(orecv, osend)                                               # slab r of [P*Ll][B][Cl] = rank r's token rows
        o_wire = sp.unpack_heads_ref(orecv.view(world, Ll, B, Cl))
        err = max(err, float((o_wire - o_ref).abs().max()))
        # --- head-group pipelining (wan_*_split): q and o as TWO head groups, each a complete wire buffer with its
        #     own all-to-all and its own attention call (k / V^T of a group are slices of the whole arrived operands)
        if Hs >= 2:
            split = (Hs // 2) * D
            n0 = Lt * B * split
            qs = sp.split_wire_ref(qa, split)                                  # what wan_rmsnorm_rope_sp_split writes
            qg = torch.empty_like(qs)
            w0 = sp.exchange(qg[:n0], qs[:n0], async_op=True)
            w1 = sp.exchange(qg[n0:], qs[n0:], async_op=True)
            w0(); w1()
            mine = q[:, :, rank * Cl:(rank + 1) * Cl]
            assert torch.equal(qg[:n0].view(Lt, B, split).permute(1, 0, 2), mine[..., :split])
            assert torch.equal(qg[n0:].view(Lt, B, Cl - split).permute(1, 0, 2), mine[..., split:])
            og = torch.empty(Lt * B * Cl)
            for lo, hi, view in ((0, split, og[:n0].view(Lt, B, split)), (split, Cl, og[n0:].view(Lt, B, Cl - split))):
                hg = (hi - lo) // D
                qv = (qg[:n0].view(Lt, B, split) if lo == 0 else qg[n0:].view(Lt, B, Cl - split))
                for b in range(B):
                    view[:, b] = O.attention(qv[:, b].reshape(Lt, hg, D), kr.view(Lt, B, Cl)[:, b, lo:hi].reshape(Lt, hg, D),
                                             vt_full[b, lo:hi, :Lp].t().reshape(Lt, hg, D), k_len=L).reshape(Lt, hi - lo)
            ogr = torch.empty_like(og)
            w0 = sp.exchange(ogr[:n0], og[:n0], async_op=True)                 # group 0 travels while group 1 is computed
            sp.exchange(ogr[n0:], og[n0:])
            w0()
            o_grp = sp.unpack_heads_ref(sp.join_wire_ref(ogr, world, Ll, B, Cl, split))
            err = max(err, float((o_grp - o_ref).abs().max()))
            assert torch.equal(sp.join_wire_ref(qs, world, Ll, B, Cl, split), qa)
        with pytest.raises(ValueError):
            sp.exchange(torch.empty(7), torch.empty(7))
        # --- final token all-gather
        y = torch.randn(B, Lp, 64)
        full = sp.all_gather_tokens(y[:, sl])
        assert torch.equal(full, y)
        # --- the reduction the ranks agree their fp8 attention exponents with (round 5): every rank measures ITS heads, the
        # element-wise maximum over the group is what a single device would have measured over all heads
        torch.manual_seed(7)
        per_head = torch.rand(H, 2) * 10                      # (max |q|, max |k - mean|) of every head, the same table on every rank
        mine = per_head[rank * Hs:(rank + 1) * Hs].amax(dim=0)
        agreed = sp.all_reduce_max(mine.clone())
        assert torch.equal(agreed, per_head.amax(dim=0))
        q_out.put((rank, err, str(dev)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 2), (4, 4), (2, 4)])
def test_ulysses_exchange_and_sharded_attention(world, H):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 100, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert [r[0] for r in res] == list(range(world))
    assert all(r[1] < 1e-5 for r in res), res


def test_degree_checks_without_process_group():
    from videocof_amd import dist as vdist
    with pytest.raises(RuntimeError, match="only Ulysses"):
        vdist.set_multi_gpus_devices(1, ring_degree=2)
    assert vdist.get_sequence_parallel_world_size() == 1
    assert str(vdist.set_multi_gpus_devices(1, 1)) == "cuda"


def test_emulated_rank_has_the_exchange_interface_and_no_peers():
    """bench.py --emulate-sp: one rank of a P-way group with device-local copies where the exchanges would be (measurement only)."""
    import torch
    from videocof_amd import dist as vdist
    try:
        sp = vdist.init_sequence_parallel(backend="emulated", rank=0, world_size=8)
        assert (sp.world_size, sp.rank) == (8, 0) and vdist.get_sequence_parallel_world_size() == 8 and vdist.get_sp_group() is sp
        send = torch.arange(64, dtype=torch.float32)
        recv = torch.zeros(64)
        wait = sp.exchange(recv, send, async_op=True)
        wait()
        assert torch.equal(recv, send) and sp.exchange(recv, send) is None
        with pytest.raises(ValueError, match="divisible"):
            sp.exchange(torch.zeros(12), torch.zeros(12))
        y = torch.randn(1, 5, 3)
        g = sp.all_gather_tokens(y)
        assert g.shape == (1, 40, 3) and all(torch.equal(g[:, 5 * r:5 * r + 5], y) for r in range(8))
        y2 = torch.randn(2, 5, 3)
        g2 = sp.all_gather_tokens(y2)
        assert g2.shape == (2, 40, 3) and torch.equal(g2[1, 10:15], y2[1])
        t = torch.tensor([1.0, 2.0])
        assert sp.all_reduce_max(t) is t
        with pytest.raises(ValueError):
            vdist.init_sequence_parallel(backend="emulated", rank=8, world_size=8)
        with pytest.raises(ValueError, match="emulated"):
            vdist.init_sequence_parallel(backend="nope")
    finally:
        vdist.destroy_sequence_parallel()
    assert vdist.get_sp_group() is None
