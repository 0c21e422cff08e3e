"""The persistent stream-K GEMM's work decomposition (videocof_amd/csrc/gemm_bf16_pk.hip), checked on the HOST: the library exports the
very functions its kernel evaluates (``wan_gemm_pk_segment``), so no GPU is needed to know that every (output tile, K tile) pair is
computed exactly once, that the pieces of a split tile chain up in K order under the lanes the combiner will read them from, that no two
pieces share a workspace slot, and that every worker gets the same amount of work."""
import collections
import ctypes

import pytest

from videocof_amd import _lib

SHAPES = [(67080, 5120, 5120), (67080, 10240, 5120), (67080, 13824, 5120), (67080, 5120, 13824),      # the 14B Linears at L = 67 080
          (8392, 5120, 5120), (8392, 10240, 5120), (8392, 13824, 5120), (8392, 5120, 13824),         # ... on an 8-way Ulysses shard
          (16776, 5120, 5120), (1024, 512, 4096), (300, 300, 128), (2304, 1536, 8960), (2304, 8960, 1536), (75600 * 2, 5120, 5120)]


def plan(lib, M, N, K, fp8=False):
    G = lib.wan_gemm_pk_grid(M, N)
    out = (ctypes.c_int * 11)()
    segs = []
    segment = lib.wan_gemm_fp8_pk_segment if fp8 else lib.wan_gemm_pk_segment
    for w in range(G):
        i = 0
        while segment(M, N, K, w, i, out):
            segs.append((w, i) + tuple(out))
            i += 1
    return G, segs


@pytest.mark.parametrize("fp8", [False, True], ids=["bf16", "e4m3"])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_every_k_tile_of_every_output_tile_is_computed_exactly_once(M, N, K, fp8):
    """(e4m3: the same plan over K tiles of 128 elements -- wan_gemm_fp8_pk_segment; K % 256 == 0, two K tiles per unit)"""
    lib = _lib.load()
    if fp8 and K % 256:
        assert plan(lib, M, N, K, True)[1] == []
        return
    G, segs = plan(lib, M, N, K, fp8)
    assert G % 8 == 0 and G >= 8
    nk, tiles_m, tiles_n = K // (128 if fp8 else 64), (M + 255) // 256, (N + 255) // 256
    cover, work, pieces, slots = collections.Counter(), collections.Counter(), collections.defaultdict(list), set()
    for (w, i, tm, tn, kb, ke, partial, slot, cnt, jlo, jhi, me, tau) in segs:
        assert 0 <= tm < tiles_m and 0 <= tn < tiles_n and 0 <= kb < ke <= nk
        assert kb % 2 == 0 and ke % 2 == 0           # a segment starts on LDS buffer 0 and keeps two K tiles in flight
        for k in range(kb, ke, 2):
            cover[(tm, tn, k)] += 1
        work[w] += ke - kb
        assert bool(partial) == (not (kb == 0 and ke == nk))
        if partial:
            assert slot not in slots and slot // 2 == (w % 8) * (G // 8) + w // 8 and me == w // 8
            slots.add(slot)
            assert 0 <= cnt < 512                    # the arrival counters live below the ticket counters in the workspace head
            pieces[(w % 8, cnt)].append((kb, ke, me, jlo, jhi, tm, tn))
    assert len(cover) == tiles_m * tiles_n * nk // 2 and set(cover.values()) == {1}
    for ps in pieces.values():
        ps.sort()
        assert ps[0][0] == 0 and ps[-1][1] == nk and all(a[1] == b[0] for a, b in zip(ps, ps[1:]))      # a K-ordered chain ...
        assert len({(p[5], p[6]) for p in ps}) == 1                                                    # ... of ONE tile ...
        assert [p[2] for p in ps] == list(range(ps[0][3], ps[0][4] + 1)) and len({(p[3], p[4]) for p in ps}) == 1   # ... under lanes j_lo..j_hi
        assert len(ps) >= 2
    if tiles_m * tiles_n * nk // 2 >= 4 * G:         # enough work to balance: no worker waits for a tail round ...
        if pieces:
            assert max(work.values()) - min(work.values()) <= max(2 * 24, nk // 2 + 2)
        else:                                        # ... or, where cutting the leftover tiles costs more than idling (round 6), for ONE whole tile
            assert max(work.values()) - min(work.values()) <= nk
    if fp8:
        return
    # (default dispatch: no workspace, the persistent kernel's, or -- small shapes on the 128^2 kernel -- the split-K form's)
    splits = int(lib.wan_gemm_ws_splits(M, N, K))
    t128 = ((M + 127) // 128) * ((N + 127) // 128)
    split_ws = ((t128 * 4 + 4095) // 4096) * 4096 + t128 * splits * 128 * 128 * 4 if splits > 1 else 0
    assert int(lib.wan_gemm_workspace_bytes(M, N, K)) in (0, 4096 + G * 2 * 256 * 256 * 4, split_ws)


def test_workspace_entry_falls_back_and_validates():
    lib = _lib.load()
    # no workspace, or a shape the persistent kernel does not take: wan_gemm_bf16's own validation answers
    assert lib.wan_gemm_bf16_ws(None, 64, None, 64, None, None, 64, 4, 64, 64, 0, None, 0, None, 0, None) == _lib.WAN_ERR_INVALID
    assert lib.wan_gemm_ws_plan(515, 64, 1024) == lib.wan_gemm_plan(515, 64, 1024) == 0
    assert lib.wan_gemm_workspace_bytes(515, 64, 512) == 0 and lib.wan_gemm_ws_splits(515, 64, 512) == 1       # 8 K tiles: nothing to cut
    # round 5: small shapes whose 128^2 tiles do not fill the chip are cut along K (BASELINE configs[0]: ffn.2 of the 1.3B model at
    # 2 304 tokens is 216 tiles of 140 K steps) -- never shapes that already fill 3/4 of a round, never pieces under 32 K tiles
    # (o / cross-o at K = 1 536: measured slower split)
    assert lib.wan_gemm_ws_splits(2304, 1536, 8960) == 2 and lib.wan_gemm_ws_splits(2304, 1536, 1536) == 1
    assert lib.wan_gemm_ws_splits(2304, 3072, 1536) == 1 and lib.wan_gemm_ws_splits(67080, 64, 5120) == 1
    assert lib.wan_gemm_workspace_bytes(2304, 1536, 8960) == 4096 + 216 * 2 * 128 * 128 * 4
    assert lib.wan_gemm_ws_plan(67080, 5120, 5120) == 3 and _lib.GEMM_VARIANT_KERNELS[3] == "gemm_pk_kernel"
    # round 5: the K = 1536 Linears of the 1.3B model at a video's token count run the persistent kernel too (K % 128 == 0, K >= 1024);
    # without a workspace they stay on the 8-wave per-tile kernel, and shallow K (the VAE attention block's 384) stays there either way
    assert lib.wan_gemm_ws_plan(67080, 5120, 1536) == 3 and lib.wan_gemm_plan(67080, 5120, 1536) == 1
    assert lib.wan_gemm_ws_plan(67080, 1536, 8960) == 3 and lib.wan_gemm_ws_plan(67080, 3072, 1536) == 3
    # round 6: the e4m3 Linear asks the same plan about the bf16 product of the same tile count (K / 2); K % 256 != 0 or no
    # workspace -> wan_gemm_fp8 (whose own validation answers)
    assert lib.wan_gemm_fp8_ws_plan(67080, 5120, 5120) == 3 and lib.wan_gemm_fp8_ws_plan(67080, 5120, 13824) == 3
    assert lib.wan_gemm_fp8_ws_plan(67080, 5120, 1536) == 1 and lib.wan_gemm_fp8_ws_plan(67080, 5120, 5120 + 128) == 1
    assert lib.wan_gemm_fp8_ws_plan(515, 64, 1024) == 1
    # ... and the 8-way Ulysses shard (M = 8 392: 660 tiles of 40 K tiles) runs persistent too: measured 1.09-1.33x the per-tile kernel
    assert lib.wan_gemm_fp8_ws_plan(8392, 5120, 5120) == 3 and lib.wan_gemm_fp8_ws_plan(8392, 5120, 13824) == 3
    assert lib.wan_gemm_fp8_ws_plan(8392, 5120, 2048) == 1 and lib.wan_gemm_ws_plan(8392, 5120, 2560) == 1
    assert lib.wan_gemm_fp8_workspace_bytes(8392, 5120, 5120) == lib.wan_gemm_workspace_bytes(8392, 5120, 5120) > 0
    assert lib.wan_gemm_fp8_workspace_bytes(8392, 5120, 2048) == 0 and lib.wan_gemm_fp8_workspace_bytes(67080, 5120, 5120 + 128) == 0
    assert lib.wan_gemm_fp8_ws(None, 128, None, None, 128, None, None, None, 64, 4, 64, 128, 0, None, 0, None, 0, None) == _lib.WAN_ERR_INVALID
    # round 6: leftover tiles go whole where the stream-K fix-up costs more than the idle lanes (measured: profiles/r06/gemm_split.log)
    def cut(M, N, K):
        return any(seg[6] for seg in plan(lib, M, N, K)[1])
    assert not cut(8392, 5120, 5120) and not cut(8392, 13824, 5120) and not cut(8392, 5120, 13824)        # the 8-way Ulysses shard
    assert not cut(67080, 3072, 1536) and not cut(67080, 1536, 1536)                                         # the 1.3B model's K = 1 536
    assert cut(8392, 10240, 5120) and cut(16770, 5120, 5120) and cut(33540, 5120, 5120) and cut(67080, 1536, 8960)
    # (20+ rounds: the 14B Linears at M = 67 080 keep the stream-K cut)
    assert cut(67080, 10240, 5120) and cut(67080, 13824, 5120) and cut(67080, 5120, 13824) and cut(67080, 5120, 5120)
    assert lib.wan_gemm_ws_plan(6240, 6240, 384) == lib.wan_gemm_plan(6240, 6240, 384) == 1
    assert lib.wan_gemm_ws_plan(2304, 3072, 1536) == lib.wan_gemm_plan(2304, 3072, 1536) == 0              # < 1 tile per 2 CUs: the 128^2 kernel
    assert lib.wan_gemm_ws_plan(2304, 8960, 1536) == lib.wan_gemm_plan(2304, 8960, 1536) == 1              # 315 tiles: shallow K needs >= 4 rounds
    need = int(lib.wan_gemm_workspace_bytes(67080, 5120, 5120))
    st = lib.wan_gemm_bf16_ws(16, 5120, 16, 5120, None, 16, 5120, 67080, 5120, 5120, 0, None, 0, 16, need - 1, None)
    assert st == _lib.WAN_ERR_INVALID and b"workspace" in lib.wan_last_error()


def test_random_shapes_are_covered_exactly_once_property():
    """Property check of the plan over random shapes (hypothesis; host arithmetic only): whatever (M, N, K) -- one tile, a sliver of a
    tile, fewer tiles than workers, K of a single unit -- every (output tile, K unit) is computed exactly once, segments are whole units,
    a split tile's pieces chain in K order, and no two pieces share a workspace slot; bf16 (64-element K tiles) and e4m3 (128)."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    lib = _lib.load()

    @settings(max_examples=40, deadline=None, derandomize=True)
    @given(M=st.integers(1, 9000), N=st.integers(1, 2400).map(lambda n: 4 * n), units=st.integers(1, 40), fp8=st.booleans())
    def check(M, N, units, fp8):
        ktile = 128 if fp8 else 64
        K = units * 2 * ktile
        G, segs = plan(lib, M, N, K, fp8)
        nk, tiles_m, tiles_n = K // ktile, (M + 255) // 256, (N + 255) // 256
        cover, slots, pieces = collections.Counter(), set(), collections.defaultdict(list)
        for (w, i, tm, tn, kb, ke, partial, slot, cnt, jlo, jhi, me, tau) in segs:
            assert 0 <= tm < tiles_m and 0 <= tn < tiles_n and 0 <= kb < ke <= nk and kb % 2 == 0 and ke % 2 == 0
            for k in range(kb, ke, 2):
                cover[(tm, tn, k)] += 1
            assert bool(partial) == (not (kb == 0 and ke == nk))
            if partial:
                assert slot not in slots and 0 <= cnt < 512
                slots.add(slot)
                pieces[(w % 8, cnt)].append((kb, ke, me, jlo, jhi))
        assert len(cover) == tiles_m * tiles_n * nk // 2 and set(cover.values()) == {1}, (M, N, K, fp8)
        for ps in pieces.values():
            ps.sort()
            assert ps[0][0] == 0 and ps[-1][1] == nk and all(a[1] == b[0] for a, b in zip(ps, ps[1:]))
            assert [p[2] for p in ps] == list(range(ps[0][3], ps[0][4] + 1))

    check()
