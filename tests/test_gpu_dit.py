"""-m gpu: the HIP-backed WanTransformer3DModel / WanPipeline against the reference-captured
fixtures (tests/golden) and the CPU oracle.  bf16 kernels vs fp32 reference, per forward:
rel-L2 <= 1e-2 and cosine >= 0.9999; 4-step trajectories <= 2e-2."""
import math

import numpy as np
import pytest
import torch

from oracle import wan_oracle as O
from videocof_amd import FlowUniPCMultistepScheduler, WanPipeline, WanTransformer3DModel
from videocof_amd.weights import deterministic_dit_state_dict, det_uniform

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TINY = dict(dim=256, ffn_dim=512, num_layers=2, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
CFG = O.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)


def rel_l2(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm())


def cosine(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu().flatten(), torch.as_tensor(b).double().cpu().flatten()
    return float(torch.dot(a, b) / (a.norm() * b.norm()))


@pytest.fixture(scope="module")
def model():
    m = WanTransformer3DModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    m.load_state_dict(deterministic_dit_state_dict(**TINY), device=DEV)
    return m


def test_g6_forward_cof(golden, model):
    g = golden("dit_g6_forward")
    lat, ctx = torch.from_numpy(g["lat"]).to(DEV), [torch.from_numpy(g["ctx"]).to(DEV)]
    out = model(lat, torch.tensor([899], device=DEV), ctx, 420, frame_split_indices=[3], ground_frame_indices=[(3, 4)])
    assert out.shape == (1, 16, 7, 12, 20) and out.dtype == torch.float32
    assert rel_l2(out, g["out_cof"]) < 1e-2 and cosine(out, g["out_cof"]) > 0.9999
    # the CoF position map matters: the T2V result must NOT match the CoF fixture
    out_t2v = model(lat, torch.tensor([499], device=DEV), ctx, 420)
    assert rel_l2(out_t2v, g["out_t2v"]) < 1e-2 and cosine(out_t2v, g["out_t2v"]) > 0.9999
    assert rel_l2(out_t2v, g["out_cof"]) > 5e-2


def test_g6_forward_batch2_and_bf16_latents(golden, model):
    g = golden("dit_g6_forward")
    lat2 = torch.from_numpy(g["lat2"]).to(DEV)
    ctx2 = [torch.from_numpy(g["ctx"]).to(DEV), torch.from_numpy(g["ctx2"]).to(DEV)]
    out = model(lat2, torch.tensor([749, 749], device=DEV), ctx2, 420, frame_split_indices=[3, 3],
                ground_frame_indices=[(3, 4), (3, 4)])
    assert rel_l2(out, g["out_b2"]) < 1e-2 and cosine(out, g["out_b2"]) > 0.9999
    out_bf = model(lat2.bfloat16(), torch.tensor([749, 749], device=DEV), ctx2, 420, frame_split_indices=[3, 3],
                   ground_frame_indices=[(3, 4), (3, 4)])
    assert out_bf.dtype == torch.bfloat16 and rel_l2(out_bf.float(), g["out_b2"]) < 1.5e-2
    # list-of-samples input form of the reference (wan_transformer3d.py:838-839)
    out_l = model([lat2[0], lat2[1]], torch.tensor([749, 749], device=DEV), ctx2, 420,
                  frame_split_indices=[3, 3], ground_frame_indices=[(3, 4), (3, 4)])
    assert torch.equal(out_l, out)


def test_forward_random_layouts_vs_oracle(model):
    """A randomised sweep of what a caller can vary at the forward seam (wan_transformer3d.py:818-833) -- latent grid (odd and even
    frame / row / column counts, i.e. clips at their native resolution), how the frames split into source | grounding | target (incl.
    no split, the paired-only form without grounding frames, several grounding frames), batch 1 / 2, seq_len == L and seq_len > L,
    prompt lengths 1 ... 77, timestep -- against the CPU oracle.  seq_len > L: the product follows the reference's flash-attn branch,
    which trims the keys to `k_lens = seq_lens` (wan_transformer3d.py:298, attention_utils.py:95-100), so padding must not change a
    valid row (test_seq_len_padding_and_assert); the oracle restates the SDPA branch, which cannot mask (attention_utils.py:198-210),
    and is therefore evaluated WITHOUT the padding."""
    import random
    rnd = random.Random(7)
    sd = deterministic_dit_state_dict(**TINY)
    worst = 0.0
    for case in range(14):
        F, Hl, Wl = rnd.randint(1, 9), 2 * rnd.randint(1, 7), 2 * rnd.randint(1, 9)
        B = rnd.choice([1, 1, 2])
        L = F * (Hl // 2) * (Wl // 2)
        seq_len = L + rnd.choice([0, 0, 1, 5, 64])
        mode = ("t2v", "paired", "cof", "cof")[case % 4] if F >= 3 else "t2v"
        fsi = gfi = None
        if mode != "t2v":
            fs = rnd.randint(1, F - 2)
            fsi = [fs] * B
            if mode == "cof":
                gfi = [(fs, fs + rnd.randint(1, F - 1 - fs))] * B
        lat = det_uniform(f"fz.lat{case}", (B, 16, F, Hl, Wl), 1.0)
        ctx = [det_uniform(f"fz.ctx{case}.{b}", (rnd.randint(1, 77), 64), 1.0) for b in range(B)]
        t = torch.tensor([rnd.choice([999, 899, 749, 499, 37])] * B)
        ref = O.dit_forward(sd, CFG, lat, t, ctx, L, fsi, gfi)
        out = model(lat.to(DEV), t.to(DEV), [c.to(DEV) for c in ctx], seq_len, frame_split_indices=fsi, ground_frame_indices=gfi)
        assert out.shape == ref.shape
        r, c = rel_l2(out, ref), cosine(out, ref)
        assert r < 1e-2 and c > 0.9999, (case, (B, F, Hl, Wl), seq_len, mode, fsi, gfi, r, c)
        worst = max(worst, r)
    print(f"[fuzz] 14 random layouts: worst rel-L2 {worst:.2e}")


def test_samples_with_different_cof_maps_in_one_call(golden, model):
    """rope_apply_qk takes the CoF position map per sample (wan_transformer3d.py:160-179): a call whose samples split their frames
    differently is served group by group -- against the oracle with per-sample maps, bitwise equal to the samples run alone, output
    order = input order; three samples of which the first and third share a map (one group of two, one of one)."""
    g = golden("dit_g6_forward")
    lat2 = torch.from_numpy(g["lat2"]).to(DEV)
    lat3 = torch.cat([lat2, lat2[:1] * 0.5])
    ctx3 = [torch.from_numpy(g["ctx"]).to(DEV), torch.from_numpy(g["ctx2"]).to(DEV), torch.from_numpy(g["ctx"]).to(DEV)]
    t3 = torch.tensor([749, 500, 749], device=DEV)
    fsi, gfi = [3, 2, 3], [(3, 4), (2, 4), (3, 4)]
    out = model(lat3, t3, ctx3, 420, frame_split_indices=fsi, ground_frame_indices=gfi)
    assert out.shape == (3, 16, 7, 12, 20)
    for b in range(3):
        alone = model(lat3[b:b + 1], t3[b:b + 1], [ctx3[b]], 420, frame_split_indices=[fsi[b]], ground_frame_indices=[gfi[b]])
        assert torch.equal(out[b], alone[0]), b
    ref = O.dit_forward(deterministic_dit_state_dict(**TINY), CFG, lat3.cpu(), t3.cpu(), [c.cpu() for c in ctx3], 420, fsi, gfi)
    assert rel_l2(out, ref) < 1e-2 and cosine(out, ref) > 0.9999
    wrong = model(lat3[1:2], t3[1:2], [ctx3[1]], 420, frame_split_indices=[3], ground_frame_indices=[(3, 4)])
    assert rel_l2(wrong[0], ref[1]) > 2 * rel_l2(out[1], ref[1])   # (the maps matter: sample 1 under its neighbours' map is another result)
    assert not torch.equal(wrong[0], out[1])
    with pytest.raises(ValueError):
        model(lat3, t3, ctx3, 420, frame_split_indices=[3, 2])


def test_g5_single_block_through_model(golden):
    """One WanAttentionBlock: run a 1-layer model whose patch-embed/head are bypassed by
    comparing the residual stream.  Done through the model's own block loop via a probe model."""
    g = golden("dit_g5_block")
    sd = deterministic_dit_state_dict(**TINY)
    # oracle block on the fixture input pins the oracle; here: HIP full model vs oracle full model with
    # L = 420 ragged tokens is covered by g6.  This test pins cross-attention's unmasked padded rows:
    ctx_short = [det_uniform("g5.short", (3, 64), 1.0)]
    lat = det_uniform("g5.lat", (1, 16, 7, 12, 20), 1.0)
    m = WanTransformer3DModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    m.load_state_dict(sd, device=DEV)
    out = m(lat.to(DEV), torch.tensor([999], device=DEV), [c.to(DEV) for c in ctx_short], 420, frame_split_indices=[3],
            ground_frame_indices=[(3, 4)])
    ref = O.dit_forward(sd, CFG, lat, torch.tensor([999]), ctx_short, 420, [3], [(3, 4)])
    assert rel_l2(out, ref) < 1e-2 and cosine(out, ref) > 0.9999
    assert g["out"].shape == (1, 420, 256)


def test_seq_len_padding_and_assert(model):
    lat = det_uniform("pad.lat", (1, 16, 3, 8, 8), 1.0).to(DEV)
    ctx = [det_uniform("pad.ctx", (5, 64), 1.0).to(DEV)]
    with pytest.raises(AssertionError):
        model(lat, torch.tensor([10], device=DEV), ctx, 40)          # L = 48 > seq_len
    a = model(lat, torch.tensor([10], device=DEV), ctx, 48)
    b = model(lat, torch.tensor([10], device=DEV), ctx, 64)          # padded keys are masked (flash k_lens semantics)
    assert rel_l2(b, a.cpu()) < 1e-6


def test_context_cache_is_parity_neutral(model):
    lat = det_uniform("cc.lat", (1, 16, 3, 8, 8), 1.0).to(DEV)
    ctx = [det_uniform("cc.ctx", (9, 64), 1.0).to(DEV)]
    a = model(lat, torch.tensor([500], device=DEV), ctx, 48)
    model.cache_context = True
    try:
        b1 = model(lat, torch.tensor([500], device=DEV), ctx, 48)
        b2 = model(lat, torch.tensor([500], device=DEV), ctx, 48)      # served from the cache
        ctx2 = [det_uniform("cc.ctx2", (9, 64), 1.0).to(DEV)]
        c = model(lat, torch.tensor([500], device=DEV), ctx2, 48)      # new prompt -> cache miss
    finally:
        model.cache_context = False
        model._ctx_cache = None
    assert torch.equal(a, b1) and torch.equal(a, b2)
    assert not torch.equal(a, c)


def test_context_cache_never_outlives_its_prompt(model):
    """The hoisted text K/V are keyed by tensor identity and the entry keeps the prompt alive: a second prompt that the
    caching allocator places at the freed address of the first (same shape, _version 0) must not be served from it."""
    lat = det_uniform("cc2.lat", (1, 16, 3, 8, 8), 1.0).to(DEV)
    t = torch.tensor([500], device=DEV)
    p1, p2 = det_uniform("cc2.p1", (9, 64), 1.0), det_uniform("cc2.p2", (9, 64), 1.0)
    want2 = model(lat, t, [p2.to(DEV)], 48)
    model.cache_context = True
    try:
        c1 = p1.to(DEV)
        a = model(lat, t, [c1], 48)
        ptr = c1.data_ptr()
        del c1
        c2 = torch.empty(9, 64, device=DEV)              # would reuse the block of c1 if the cache did not hold it
        c2.copy_(p2)
        got2 = model(lat, t, [c2], 48)
        assert model._ctx_cache[0][0] is c2 and c2.data_ptr() != ptr
    finally:
        model.cache_context = False
        model.clear_context_cache()
    assert torch.equal(got2, want2) and not torch.equal(a, want2)


def test_pipeline_clears_the_context_cache(golden, model):
    g = golden("dit_g8_cof_loop")
    pipe = WanPipeline(transformer=model, scheduler=FlowUniPCMultistepScheduler(shift=1))
    lat = torch.cat([torch.from_numpy(g["src"]), torch.from_numpy(g["noise"])], dim=2).to(DEV)
    kw = dict(latents=lat, source_frames=9, reasoning_frames=4, num_inference_steps=2, guidance_scale=1.0, shift=3,
              repeat_rope=True, cot=True, output_type="latent", weight_dtype=torch.float32)
    ctx_a = torch.from_numpy(g["ctx"]).to(DEV)
    a = pipe(prompt_embeds=[ctx_a], **kw).latents
    assert model._ctx_cache is None and model.cache_context is False
    ctx_b = (ctx_a * 0.5).contiguous()
    b = pipe(prompt_embeds=[ctx_b], **kw).latents
    model.cache_context = False
    b_ref = pipe(prompt_embeds=[ctx_b], cache_context=False, **kw).latents
    assert torch.equal(b, b_ref) and not torch.equal(a, b)


def test_graph_replay_is_bit_identical(golden, model):
    """hipGraph capture of the denoise step (8f-1): a 4-step CoF loop replayed from the graph gives the SAME BITS as the
    eager loop, for the call that captures (eager step, capture + replay, replays), for a later call with another prompt
    (replays only; text K/V refreshed in place outside the graph) and with CFG (B = 2: its own graph)."""
    from videocof_amd import GraphedForward
    g = golden("dit_g8_cof_loop")
    lat = torch.cat([torch.from_numpy(g["src"]), torch.from_numpy(g["noise"])], dim=2).to(DEV)
    ctx_a = torch.from_numpy(g["ctx"]).to(DEV)
    ctx_b = (ctx_a * 0.7).contiguous()
    kw = dict(latents=lat, source_frames=9, reasoning_frames=4, num_inference_steps=4, shift=3, repeat_rope=True, cot=True,
              output_type="latent", weight_dtype=torch.float32)

    def run(pipe, ctx, graph, scale=1.0):
        return pipe(prompt_embeds=[ctx], negative_prompt_embeds=[ctx_b * 0.5] if scale > 1 else None, guidance_scale=scale,
                    capture_graph=graph, **kw).latents

    eager = WanPipeline(transformer=model, scheduler=FlowUniPCMultistepScheduler(shift=1))
    graphed = WanPipeline(transformer=model, scheduler=FlowUniPCMultistepScheduler(shift=1))
    want_a, want_b, want_cfg = run(eager, ctx_a, False), run(eager, ctx_b, False), run(eager, ctx_a, False, 3.0)
    got_a = run(graphed, ctx_a, True)
    assert isinstance(graphed._graphed, GraphedForward) and graphed._graphed.replays == 3      # 1 eager + 3 replays
    got_b = run(graphed, ctx_b, True)
    assert graphed._graphed.replays == 7                                                      # 4 more, no new capture
    got_cfg = run(graphed, ctx_a, True, 3.0)
    assert torch.equal(got_a, want_a) and torch.equal(got_b, want_b) and torch.equal(got_cfg, want_cfg)
    assert not torch.equal(got_a, got_b)
    assert float(got_a[:, :, :3].sub(lat[:, :, :3]).abs().max()) < 1e-5       # source frames: masked on the device
    assert model.mask_source_frames == 0 and model._ctx_cache is None          # the pipeline restored the model


def test_whole_loop_graph_is_bit_identical(golden, model):
    """capture_graph='loop': every step's forward, the CFG combine, the CoF mask and the UniPC update of the whole denoise loop
    (pipeline_wan.py:694-740) recorded as ONE hipGraph.  Call 1 of a signature is eager, call 2 captures + replays, later calls
    replay -- with another prompt (other length: the embeddings live zero-padded in a static buffer), other latents, and with
    CFG (its own signature).  All of them give the SAME BITS as the eager loop."""
    from videocof_amd import GraphedLoop
    g = golden("dit_g8_cof_loop")
    lat = torch.cat([torch.from_numpy(g["src"]), torch.from_numpy(g["noise"])], dim=2).to(DEV)
    lat2 = lat.clone()
    lat2[:, :, 3:] = lat[:, :, 3:].flip(-1)
    ctx_a = torch.from_numpy(g["ctx"]).to(DEV)
    ctx_b = (ctx_a[:5] * 0.7).contiguous()                       # another prompt, another token count
    kw = dict(source_frames=9, reasoning_frames=4, num_inference_steps=4, shift=3, repeat_rope=True, cot=True,
              output_type="latent", weight_dtype=torch.float32)

    def run(pipe, latents, ctx, graph, scale=1.0):
        return pipe(latents=latents, prompt_embeds=[ctx], negative_prompt_embeds=[ctx_a * 0.5] if scale > 1 else None,
                    guidance_scale=scale, capture_graph=graph, **kw).latents

    eager = WanPipeline(transformer=model, scheduler=FlowUniPCMultistepScheduler(shift=1))
    looped = WanPipeline(transformer=model, scheduler=FlowUniPCMultistepScheduler(shift=1))
    want = [run(eager, lat, ctx_a, False), run(eager, lat, ctx_a, False), run(eager, lat2, ctx_b, False)]
    want_cfg = run(eager, lat, ctx_a, False, 3.0)
    got = [run(looped, lat, ctx_a, "loop"), run(looped, lat, ctx_a, "loop"), run(looped, lat2, ctx_b, "loop")]
    assert isinstance(looped._graphed_loop, GraphedLoop) and looped._graphed_loop.replays == 2      # eager, capture + replay, replay
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert not torch.equal(got[0], got[2])
    got_cfg = [run(looped, lat, ctx_a, "loop", 3.0) for _ in range(3)]
    assert looped._graphed_loop.replays == 4 and all(torch.equal(x, want_cfg) for x in got_cfg)
    assert torch.equal(run(looped, lat, ctx_a, "loop"), want[0])                                    # back to the first signature
    assert model.mask_source_frames == 0 and model._ctx_cache is None
    with pytest.raises(NotImplementedError, match="callback"):
        looped(latents=lat, prompt_embeds=[ctx_a], guidance_scale=1.0, capture_graph="loop", callback_on_step_end=lambda *a: None, **kw)


def test_graph_entries_pin_their_buffers_and_follow_the_model(model):
    """A captured graph replays raw device addresses.  (i) Two call shapes A, B, A: shape B (and an eager call of yet another
    shape) must not free or reuse the workspaces graph A was captured with -- A's replay still equals the eager result, bitwise;
    (ii) fractional timesteps are not truncated by the static timestep buffer; (iii) reloading the weights drops the captured
    graphs instead of replaying stale kernels on freed weight tensors; (iv) release_workspaces() does the same."""
    from videocof_amd import GraphedForward
    ctx = [det_uniform("gp.ctx", (9, 64), 1.0).to(DEV)]
    lat_a, lat_b = det_uniform("gp.a", (1, 16, 5, 8, 8), 1.0).to(DEV), det_uniform("gp.b", (1, 16, 7, 12, 20), 1.0).to(DEV)
    lat_c = det_uniform("gp.c", (1, 16, 3, 16, 16), 1.0).to(DEV)
    ta, tb = torch.tensor([500], device=DEV), torch.tensor([250], device=DEV)
    want_a, want_b = model(lat_a, ta, ctx, 80), model(lat_b, tb, ctx, 420)
    import gc
    gc.collect()                                         # GraphedForwards of earlier tests un-pin their workspaces when they die
    pinned0 = sum(1 for v in model._bufs.values() if v.pinned)
    gf = GraphedForward(model)
    for _ in range(3):                                   # eager, capture + replay, replay
        got_a = gf(lat_a, ta, ctx, 80)
    assert torch.equal(got_a, want_a) and gf.replays == 2
    for _ in range(3):
        got_b = gf(lat_b, tb, ctx, 420)                  # a second shape: its own workspaces, its own graph
    assert torch.equal(got_b, want_b) and gf.replays == 4
    model(lat_c, ta, ctx, 192)                           # an eager call of a third shape evicts only unpinned workspaces
    junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(64)]     # whatever was freed gets overwritten
    assert torch.equal(gf(lat_a, ta, ctx, 80), want_a) and torch.equal(gf(lat_b, tb, ctx, 420), want_b)
    assert gf.replays == 6 and sum(1 for v in model._bufs.values() if v.pinned) == pinned0 + 2
    del junk
    # (ii)
    tf = torch.tensor([500.5], device=DEV)
    want_f = model(lat_a, tf, ctx, 80)
    assert not torch.equal(want_f, want_a)
    for _ in range(2):
        got_f = gf(lat_a, tf, ctx, 80)
    assert torch.equal(got_f, want_f)
    # (iii) other weights: the captured graphs are dropped, the next call is eager again and right
    sd2 = {k: (v * 1.01 if k.endswith("ffn.0.weight") else v) for k, v in deterministic_dit_state_dict(**TINY).items()}
    replays = gf.replays
    model.load_state_dict(sd2, device=DEV)
    try:
        want_a2 = model(lat_a, ta, ctx, 80)
        got_a2 = gf(lat_a, ta, ctx, 80)
        assert gf.replays == replays and torch.equal(got_a2, want_a2) and not torch.equal(want_a2, want_a)
        assert all(e.epoch == model._graph_epoch for e in gf._entries.values())
        gf(lat_a, ta, ctx, 80)                               # captured again
        # (iv)
        model.release_workspaces()
        assert not model._bufs
        assert torch.equal(gf(lat_a, ta, ctx, 80), want_a2) and len(gf._entries) == 1
    finally:
        model.load_state_dict(deterministic_dit_state_dict(**TINY), device=DEV)
    assert torch.equal(model(lat_a, ta, ctx, 80), want_a)


def test_unpatchify_method_is_the_references_einsum(model):
    """``WanTransformer3DModel.unpatchify(x, grid_sizes)`` (wan_transformer3d.py:1108-1131): per sample the first prod(grid) rows,
    `fhwpqrc->cfphqwr`; fp32 and bf16 token lists, a tensor or a list of grids, rows beyond the grid ignored."""
    grids = [(3, 4, 5), (2, 2, 3)]
    xs = [torch.randn(3 * 4 * 5 + 7, 64, device=DEV), torch.randn(2 * 2 * 3, 64, device=DEV)]
    for dt in (torch.float32, torch.bfloat16):
        got = model.unpatchify([x.to(dt) for x in xs], torch.tensor(grids))
        for x, v, o in zip(xs, grids, got):
            u = x.to(dt)[:v[0] * v[1] * v[2]].view(*v, 1, 2, 2, 16)
            want = torch.einsum("fhwpqrc->cfphqwr", u).reshape(16, v[0], v[1] * 2, v[2] * 2)
            assert o.dtype == dt and torch.equal(o, want)
    assert torch.equal(model.unpatchify(xs[:1], [grids[0]])[0], model.unpatchify(xs[:1], torch.tensor(grids[:1]))[0])


def test_unpatchify_zero_frames_is_the_cof_mask(model):
    lat = det_uniform("zf.lat", (1, 16, 5, 8, 8), 1.0).to(DEV)
    ctx = [det_uniform("zf.ctx", (9, 64), 1.0).to(DEV)]
    t = torch.tensor([500], device=DEV)
    plain = model(lat, t, ctx, 80, frame_split_indices=[2], ground_frame_indices=[(2, 3)])
    model.mask_source_frames = 2
    try:
        masked = model(lat, t, ctx, 80, frame_split_indices=[2], ground_frame_indices=[(2, 3)])
    finally:
        model.mask_source_frames = 0
    want = plain.clone()
    want[:, :, :2] = 0
    assert torch.equal(masked, want) and float(plain[:, :, :2].abs().max()) > 0


def test_g14_teacache_sequence(golden, model):
    """TeaCache (opt-in, lossy): the HIP model against the 8-step sequence captured from the reference with enable_teacache --
    same run / skip decisions, outputs within the bf16 forward tolerance; then off again == the plain forward."""
    g = golden("dit_g14_teacache")
    lat0, dl = torch.from_numpy(g["lat0"]).to(DEV), torch.from_numpy(g["dlat"]).to(DEV)
    ctx = [torch.from_numpy(g["ctx"]).to(DEV)]
    kw = dict(frame_split_indices=[3], ground_frame_indices=[(3, 4)])
    for key, skip in (("", 1), ("_skip3", 3)):
        model.enable_teacache(g["coeff"].tolist(), len(g["ts"]), float(g["thresh"]), num_skip_start_steps=skip, offload=False)
        decisions = []
        try:
            for i, t in enumerate(g["ts"]):
                out = model(lat0 + i * dl, torch.tensor([int(t)], device=DEV), ctx, 420, **kw)
                decisions.append(bool(model.should_calc))
                assert rel_l2(out, g["out" + key][i]) < 1.2e-2 and cosine(out, g["out" + key][i]) > 0.9999, (key, i)
            assert model.teacache.cnt == 0                     # reset after num_steps forwards
        finally:
            model.disable_teacache()
        assert decisions == g["calc" + key].tolist()
    plain = model(lat0, torch.tensor([999], device=DEV), ctx, 420, **kw)
    assert rel_l2(plain, g["out"][0]) < 1e-2
    model.enable_teacache(g["coeff"].tolist(), 4, 0.1, num_skip_start_steps=0)
    try:
        with pytest.raises(TypeError, match="num_skip_start_steps"):       # the reference fails at the same point (cache_utils.py:65)
            model(lat0, torch.tensor([999], device=DEV), ctx, 420, **kw)
    finally:
        model.disable_teacache()


def test_block_composite_is_the_python_launch_sequence(golden, model):
    """wan_dit_block_forward (one C call per block) enqueues the same kernels with the same arguments as the per-op Python
    sequence: bit-identical outputs, B = 1 and B = 2, padded sequence; and the raw C ABI rejects inconsistent arguments."""
    g = golden("dit_g6_forward")
    lat2 = torch.from_numpy(g["lat2"]).to(DEV)
    ctx2 = [torch.from_numpy(g["ctx"]).to(DEV), torch.from_numpy(g["ctx2"]).to(DEV)]
    outs = {}
    for flag in (True, False):
        model.use_block_composite = flag
        try:
            outs[flag] = (model(lat2[:1], torch.tensor([899], device=DEV), ctx2[:1], 448, frame_split_indices=[3],
                                ground_frame_indices=[(3, 4)]),
                          model(lat2, torch.tensor([749, 749], device=DEV), ctx2, 420, frame_split_indices=[3, 3],
                                ground_frame_indices=[(3, 4), (3, 4)]))
        finally:
            model.use_block_composite = True
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    assert rel_l2(outs[True][1], g["out_b2"]) < 1e-2
    # the last-block suffix launches (skip_source_frames) have their own attention scratch: a differently sized launch must
    # never re-allocate -- and thereby invalidate -- the scratch whose address the composite hands to the C side, nor leave
    # the sticky "max-free attempt off" word of any call site set
    model.skip_source_frames = 3
    try:
        a = model(lat2[:1], torch.tensor([899], device=DEV), ctx2[:1], 420, frame_split_indices=[3], ground_frame_indices=[(3, 4)])
        b = model(lat2[:1], torch.tensor([899], device=DEV), ctx2[:1], 420, frame_split_indices=[3], ground_frame_indices=[(3, 4)])
    finally:
        model.skip_source_frames = 0
    assert torch.equal(a, b) and torch.equal(a[:, :, 3:], outs[True][0][:, :, 3:] * 0 + a[:, :, 3:])
    for ws in (model._ws_self, model._ws_cross, model._ws_self_sfx, model._ws_cross_sfx):
        assert ws.buf is None or int(ws.buf[:4].view(torch.int32)) == 0
    from videocof_amd import _lib
    import ctypes
    lib = _lib.load()
    bytes6, ldvt = (ctypes.c_int64 * 6)(), ctypes.c_int64()
    assert lib.wan_dit_block_workspace_bytes(256, 512, 2, 420, 420, bytes6, ctypes.byref(ldvt)) == 0
    assert ldvt.value == 448 and list(bytes6) == [2 * 420 * 256 * 2, 2 * 420 * 512 * 2, 2 * 420 * 256 * 2, 2 * 420 * 256 * 2,
                                                  2 * 420 * 512 * 2, 2 * 256 * 448 * 2]
    st = lib.wan_dit_block_forward(None, None, None, None, None, None, None, None, None, 1, 420, 420, None)
    assert st == _lib.WAN_ERR_INVALID


def test_forward_composite_is_the_python_launch_sequence(golden, model):
    """wan_dit_forward (the whole token path as one C call) against the per-block composite and the per-op Python sequence:
    bit-identical for fp32 and bf16 latents, B = 1 padded and B = 2, with the CoF mask folded into the unpatchify; the raw
    entry rejects null arguments."""
    g = golden("dit_g6_forward")
    lat2 = torch.from_numpy(g["lat2"]).to(DEV)
    ctx2 = [torch.from_numpy(g["ctx"]).to(DEV), torch.from_numpy(g["ctx2"]).to(DEV)]
    t1, t2 = torch.tensor([899], device=DEV), torch.tensor([749, 749], device=DEV)
    kw1 = dict(frame_split_indices=[3], ground_frame_indices=[(3, 4)])
    kw2 = dict(frame_split_indices=[3, 3], ground_frame_indices=[(3, 4), (3, 4)])
    outs = {}
    for name, fwd, blk in (("forward", True, True), ("block", False, True), ("ops", False, False)):
        model.use_forward_composite, model.use_block_composite = fwd, blk
        try:
            model.mask_source_frames = 0
            a = model(lat2[:1], t1, ctx2[:1], 448, **kw1)
            b = model(lat2, t2, ctx2, 420, **kw2)
            c = model(lat2.bfloat16(), t2, ctx2, 420, **kw2)
            model.mask_source_frames = 3
            d = model(lat2[:1], t1, ctx2[:1], 420, **kw1)
            outs[name] = (a, b, c, d)
        finally:
            model.use_forward_composite = model.use_block_composite = True
            model.mask_source_frames = 0
    for name in ("block", "ops"):
        for u, v in zip(outs["forward"], outs[name]):
            assert u.dtype == v.dtype and torch.equal(u, v), name
    assert rel_l2(outs["forward"][1], g["out_b2"]) < 1e-2
    assert outs["forward"][2].dtype == torch.bfloat16
    d = outs["forward"][3]
    assert float(d[:, :, :3].abs().max()) == 0 and float(d[:, :, 3:].abs().max()) > 0
    from videocof_amd import _lib
    lib = _lib.load()
    st = lib.wan_dit_forward(None, 0, None, 0, None, None, None, None, None, None, None, None, None, 1, 7, 12, 20, 420, 0, None)
    assert st == _lib.WAN_ERR_INVALID


def test_g7_sched50_on_device(golden):
    g50 = golden("dit_g7_sched50")
    s = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, solver_order=2)
    s.set_timesteps(50, device=DEV, shift=5.0)
    assert s.timesteps.is_cuda and s.timesteps.cpu().tolist() == g50["timesteps"].tolist()
    np.testing.assert_array_equal(s.sigmas.numpy(), g50["sigmas"])


def test_g8_cof_denoise_loop(golden, model):
    g = golden("dit_g8_cof_loop")
    pipe = WanPipeline(transformer=model, scheduler=FlowUniPCMultistepScheduler(shift=1))
    lat = torch.cat([torch.from_numpy(g["src"]), torch.from_numpy(g["noise"])], dim=2).to(DEV)
    seen = []
    out = pipe(latents=lat, prompt_embeds=[torch.from_numpy(g["ctx"]).to(DEV)], source_frames=9, reasoning_frames=4,
               num_inference_steps=4, guidance_scale=1.0, shift=3, repeat_rope=True, cot=True, output_type="latent",
               weight_dtype=torch.float32, callback_on_step_end=lambda p, i, t, kw: seen.append(kw["latents"].clone()) or {})
    for i in range(4):
        assert rel_l2(seen[i], g["steps"][i]) < 2e-2, i
    assert cosine(out.latents, g["steps"][3]) > 0.9998
    # source frames: algebraically fixed; tolerance-based (parity checklist 11)
    assert float((out.latents[:, :, :3].cpu() - torch.from_numpy(g["src"])).abs().max()) < 1e-5


def test_g8b_cfg_loop(golden, model):
    g, gb = golden("dit_g8_cof_loop"), golden("dit_g8b_cfg_loop")
    pipe = WanPipeline(transformer=model, scheduler=FlowUniPCMultistepScheduler(shift=1))
    lat = torch.cat([torch.from_numpy(g["src"]), torch.from_numpy(g["noise"])], dim=2).to(DEV)
    out = pipe(latents=lat, prompt_embeds=[torch.from_numpy(g["ctx"]).to(DEV)],
               negative_prompt_embeds=[torch.from_numpy(gb["neg"]).to(DEV)], source_frames=9, reasoning_frames=4,
               num_inference_steps=3, guidance_scale=5.0, shift=5.0, repeat_rope=True, cot=True,
               output_type="latent", weight_dtype=torch.float32)
    # CFG amplifies (cond - uncond) by 5: looser bound than the single forward
    assert rel_l2(out.latents, gb["steps"][2]) < 5e-2 and cosine(out.latents, gb["steps"][2]) > 0.999


def test_pipeline_bf16_mode_runs_and_is_close(golden, model):
    """The shipped mode: bf16 latents end to end (fast_infer.py:158)."""
    g = golden("dit_g8_cof_loop")
    pipe = WanPipeline(transformer=model, scheduler=FlowUniPCMultistepScheduler(shift=1))
    lat = torch.cat([torch.from_numpy(g["src"]), torch.from_numpy(g["noise"])], dim=2).to(DEV)
    out = pipe(latents=lat, prompt_embeds=[torch.from_numpy(g["ctx"]).to(DEV)], source_frames=9, reasoning_frames=4,
               num_inference_steps=4, guidance_scale=1.0, shift=3, repeat_rope=True, cot=True, output_type="latent",
               weight_dtype=torch.bfloat16)
    assert out.latents.dtype == torch.bfloat16
    assert rel_l2(out.latents.float(), g["steps"][3]) < 4e-2
    # `timesteps=`: with the UniPC scheduler the reference never reads it (pipeline_wan.py:613-615) -- accepted and ignored, same bits
    again = pipe(latents=lat, prompt_embeds=[torch.from_numpy(g["ctx"]).to(DEV)], source_frames=9, reasoning_frames=4,
                 num_inference_steps=4, guidance_scale=1.0, shift=3, repeat_rope=True, cot=True, output_type="latent",
                 weight_dtype=torch.bfloat16, timesteps=[900, 500, 100, 10])
    assert torch.equal(again.latents, out.latents)


def test_wider_model_3_heads_vs_oracle():
    """C=384 / 3 heads / 3 layers: catches head-ordering mistakes the 2-head fixture cannot."""
    cfgd = dict(dim=384, ffn_dim=768, num_layers=3, in_dim=16, out_dim=16, text_dim=128, freq_dim=256)
    sd = deterministic_dit_state_dict(**cfgd)
    m = WanTransformer3DModel(dim=384, ffn_dim=768, num_heads=3, num_layers=3, text_dim=128)
    m.load_state_dict(sd, device=DEV)
    lat = det_uniform("w.lat", (1, 16, 5, 10, 14), 1.0)
    ctx = [det_uniform("w.ctx", (21, 128), 1.0)]
    out = m(lat.to(DEV), torch.tensor([650], device=DEV), [c.to(DEV) for c in ctx], 175,
            frame_split_indices=[2], ground_frame_indices=[(2, 3)])
    cfg = O.DiTConfig(dim=384, ffn_dim=768, num_heads=3, num_layers=3, text_dim=128)
    ref = O.dit_forward(sd, cfg, lat, torch.tensor([650]), ctx, 175, [2], [(2, 3)])
    assert rel_l2(out, ref) < 1e-2 and cosine(out, ref) > 0.9999


def test_baseline_config0_wan_1p3b_single_forward_vs_cpu_oracle():
    """BASELINE.json configs[0]: the real Wan2.1-T2V-1.3B architecture (dim 1536, 12 heads, ffn 8960, 30 layers,
    text_dim 4096, 1.42 B parameters) on a 9 x 32 x 32 random latent (L = 2304), CoF layout 4|1|4, one forward
    against the fp32 CPU oracle on the same weights (random, rounded to bf16 so both sides hold identical values)."""
    from videocof_amd.weights import random_dit_state_dict
    cfgd = dict(dim=1536, ffn_dim=8960, num_layers=30, in_dim=16, out_dim=16, text_dim=4096, freq_dim=256)
    sd = random_dit_state_dict("cpu", dtype=torch.bfloat16, seed=3, **cfgd)
    g = torch.Generator().manual_seed(0)
    for k in sd:                      # random_* zeroes biases / unit norms: perturb them so they are exercised
        if k.endswith(".bias"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.02
        elif k.endswith(("norm_q.weight", "norm_k.weight", "norm3.weight")):
            sd[k] = 1 + torch.randn(sd[k].shape, generator=g) * 0.1
    sd = {k: v.float() for k, v in sd.items()}
    m = WanTransformer3DModel(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30, text_dim=4096)
    m.load_state_dict(sd, device=DEV)
    lat = torch.randn(1, 16, 9, 32, 32, generator=g)
    ctx = [torch.randn(77, 4096, generator=g)]
    out = m(lat.to(DEV), torch.tensor([749], device=DEV), [c.to(DEV) for c in ctx], 2304,
            frame_split_indices=[4], ground_frame_indices=[(4, 5)])
    cfg = O.DiTConfig(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30, text_dim=4096)
    ref = O.dit_forward(sd, cfg, lat, torch.tensor([749]), ctx, 2304, [4], [(4, 5)])
    assert out.shape == ref.shape == (1, 16, 9, 32, 32)
    assert rel_l2(out, ref) < 1.5e-2 and cosine(out, ref) > 0.9999


def test_end_to_end_video_in_video_out():
    """WanPipeline with the HIP VAE and the HIP DiT: source video -> VAE encode -> CoF latents ->
    4-step denoise -> decode ground + edit segments, against the same chain built from the CPU oracles
    (pipeline_wan.py:595-799)."""
    from oracle.vae_oracle import WanVAEOracle
    from videocof_amd import AutoencoderKLWan
    from videocof_amd.weights import deterministic_vae_state_dict
    vsd = deterministic_vae_state_dict()
    vae = AutoencoderKLWan()
    vae.load_state_dict(vsd, device=DEV)
    sd = deterministic_dit_state_dict(**TINY)
    m = WanTransformer3DModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    m.load_state_dict(sd, device=DEV)
    video = det_uniform("e2e.video", (1, 3, 9, 32, 48), 0.8)           # 9 source frames -> 3 latent frames
    ctx = [det_uniform("e2e.ctx", (11, 64), 1.0)]
    gen = torch.Generator(device=DEV).manual_seed(7)
    pipe = WanPipeline(vae=vae, transformer=m, scheduler=FlowUniPCMultistepScheduler(shift=1))
    out = pipe(video=video.to(DEV), prompt_embeds=[c.to(DEV) for c in ctx], height=32, width=48, source_frames=9,
               reasoning_frames=4, num_inference_steps=4, guidance_scale=1.0, shift=3, repeat_rope=True, cot=True,
               generator=gen, weight_dtype=torch.float32, output_type="numpy", return_dict=True)
    assert out.ground_videos.shape == (1, 3, 1, 32, 48)        # 1 grounding latent -> 1 frame
    assert out.edit_videos.shape == (1, 3, 9, 32, 48)          # 3 target latents -> 9 frames
    assert out.videos.shape == (1, 3, 10, 32, 48)
    assert 0.0 <= float(out.videos.min()) and float(out.videos.max()) <= 1.0
    # `videos` = grounding | edit frames (pipeline_wan.py:777), decoded straight into one page-locked clip: the segments are its views
    assert out.videos.dtype == np.float32 and np.array_equal(out.videos[:, :, :1], out.ground_videos)
    assert np.array_equal(out.videos[:, :, 1:], out.edit_videos) and np.shares_memory(out.videos, out.edit_videos)
    # oracle chain from the SAME noise (the pipeline's latents[:, :, 3:] at step 0 are the generator's draw)
    gen2 = torch.Generator(device=DEV).manual_seed(7)
    noise = torch.randn((1, 16, 4, 4, 6), generator=gen2, device=DEV, dtype=torch.float32).cpu()
    orc = WanVAEOracle(vsd)
    src = orc.encode(video[0])[:16][None]
    steps = O.cof_denoise(sd, CFG, src, noise, ctx, 4, 3.0, 3, 1)
    lat = steps[-1]
    ref_ground = (orc.decode(lat[0, :, 3:4]) / 2 + 0.5).clamp(0, 1)
    ref_edit = (orc.decode(lat[0, :, 4:]) / 2 + 0.5).clamp(0, 1)
    assert rel_l2(out.latents, lat) < 4e-2
    assert rel_l2(out.ground_videos[0], ref_ground) < 5e-2
    assert rel_l2(out.edit_videos[0], ref_edit) < 5e-2


def test_from_pretrained_safetensors_and_lora(tmp_path, model):
    """Checkpoint ingest as fast_infer.py does it (wan_transformer3d.py:1157-1299): config.json +
    sharded *.safetensors with the reference's key names; then a LoRA merged on the state dict."""
    import json
    from safetensors.torch import save_file
    from videocof_amd.lora_utils import merge_lora_state_dict
    sd = deterministic_dit_state_dict(**TINY)
    keys = sorted(sd)
    save_file({k: sd[k].contiguous() for k in keys[: len(keys) // 2]}, str(tmp_path / "diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file({k: sd[k].contiguous() for k in keys[len(keys) // 2:]}, str(tmp_path / "diffusion_pytorch_model-00002-of-00002.safetensors"))
    json.dump(dict(model_type="t2v", dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, in_dim=16,
                   out_dim=16, freq_dim=256, eps=1e-6, _class_name="WanTransformer3DModel", unknown_key=1),
              open(tmp_path / "config.json", "w"))
    m = WanTransformer3DModel.from_pretrained(str(tmp_path))
    lat = det_uniform("fp.lat", (1, 16, 3, 8, 8), 1.0).to(DEV)
    ctx = [det_uniform("fp.ctx", (9, 64), 1.0).to(DEV)]
    t = torch.tensor([321], device=DEV)
    assert torch.equal(m(lat, t, ctx, 48), model(lat, t, ctx, 48))
    lora = {"diffusion_model.blocks.0.self_attn.q.lora_down.weight": det_uniform("fp.d", (4, 256), 0.3),
            "diffusion_model.blocks.0.self_attn.q.lora_up.weight": det_uniform("fp.u", (256, 4), 0.3)}
    sd2 = {k: v.clone() for k, v in sd.items()}
    assert merge_lora_state_dict(sd2, lora, 1.0, device=DEV) == 1
    m2 = WanTransformer3DModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    m2.load_state_dict(sd2, device=DEV)
    out = m2(lat, t, ctx, 48)
    ref = O.dit_forward(sd2, CFG, lat.cpu(), t.cpu(), [c.cpu() for c in ctx], 48)
    assert rel_l2(out, ref) < 1e-2 and not torch.equal(out, model(lat, t, ctx, 48))


def test_from_pretrained_mismatched_checkpoint_like_the_reference(tmp_path, model, capsys):
    """A checkpoint that does not fit the config, loaded the way the reference loads it (wan_transformer3d.py:1259-1290):
    torch-pickle file, 20-channel patch embedding truncated to 16, a wrong-size matrix and an unknown key skipped with the
    reference's message, what is then missing REPORTED and filled with a fresh model's initial values -- and the forward of
    the result equals the oracle on the state dict those rules produce."""
    import json
    sd = deterministic_dit_state_dict(**TINY)
    ckpt = {k: v.clone() for k, v in sd.items()}
    wide = det_uniform("mm.pe", (256, 20, 1, 2, 2), 0.05)
    ckpt["patch_embedding.weight"] = wide
    ckpt["blocks.1.ffn.2.weight"] = torch.zeros(256, 1024)          # size mismatch -> skipped -> fresh xavier values
    ckpt["blocks.0.cross_attn.norm_q.weight"] = torch.zeros(128)     # size mismatch -> skipped -> ones
    ckpt["img_emb.proj.0.weight"] = torch.zeros(4)                   # another family's key -> skipped
    del ckpt["blocks.1.self_attn.o.bias"]                            # absent -> zeros
    torch.save(ckpt, str(tmp_path / "diffusion_pytorch_model.bin"))
    json.dump(dict(model_type="t2v", dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, in_dim=16,
                   out_dim=16, freq_dim=256, eps=1e-6, _class_name="WanTransformer3DModel"), open(tmp_path / "config.json", "w"))
    m = WanTransformer3DModel.from_pretrained(str(tmp_path), torch_dtype=torch.bfloat16)
    out = capsys.readouterr().out
    for key in ("blocks.1.ffn.2.weight", "blocks.0.cross_attn.norm_q.weight", "img_emb.proj.0.weight"):
        assert f"{key} Size don't match, skip" in out
    assert "### missing keys: 3;" in out and "blocks.1.self_attn.o.bias" in out
    got = m.state_dict()
    assert torch.equal(got["patch_embedding.weight"].cpu().float(), wide[:, :16].bfloat16().float())
    assert torch.equal(got["blocks.0.cross_attn.norm_q.weight"].cpu(), torch.ones(256))
    assert float(got["blocks.1.self_attn.o.bias"].abs().max()) == 0.0
    w = got["blocks.1.ffn.2.weight"].float()
    assert 0 < float(w.abs().max()) <= (6.0 / (256 + 512)) ** 0.5 + 1e-3
    eff = {k: v.detach().cpu().float() for k, v in got.items()}
    lat = det_uniform("mm.lat", (1, 16, 3, 8, 8), 1.0).to(DEV)
    ctx = [det_uniform("mm.ctx", (9, 64), 1.0).to(DEV)]
    t = torch.tensor([500], device=DEV)
    ref = O.dit_forward(eff, CFG, lat.cpu(), t.cpu(), [c.cpu() for c in ctx], 48)
    assert rel_l2(m(lat, t, ctx, 48), ref) < 1e-2
    # the strict loader refuses the same tensors, naming the key
    m2 = WanTransformer3DModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    with pytest.raises(ValueError, match="size mismatch for patch_embedding.weight"):
        m2.load_state_dict(ckpt, device=DEV)
    with pytest.raises(NotImplementedError, match="bfloat16"):
        WanTransformer3DModel.from_pretrained(str(tmp_path), torch_dtype=torch.float16)


def test_skip_source_prediction_is_parity_neutral(golden, model):
    """The last block may skip the query rows of the source frames whose prediction the pipeline
    zeroes (pipeline_wan.py:736): the denoised latents must not change."""
    g = golden("dit_g8_cof_loop")
    lat = torch.cat([torch.from_numpy(g["src"]), torch.from_numpy(g["noise"])], dim=2).to(DEV)
    kw = dict(latents=lat, prompt_embeds=[torch.from_numpy(g["ctx"]).to(DEV)], source_frames=9, reasoning_frames=4,
              num_inference_steps=4, guidance_scale=1.0, shift=3, repeat_rope=True, cot=True, output_type="latent",
              weight_dtype=torch.float32)
    pipe = WanPipeline(transformer=model, scheduler=FlowUniPCMultistepScheduler(shift=1))
    a = pipe(skip_source_prediction=False, **kw).latents
    b = pipe(skip_source_prediction=True, **kw).latents
    assert model.skip_source_frames == 0          # restored after the call
    assert rel_l2(b, a.cpu()) < 1e-6              # same kernels on the rows that matter
    assert rel_l2(b, g["steps"][3]) < 2e-2
    # a direct forward with the attribute set returns zeros on the skipped frames
    model.skip_source_frames = 3
    try:
        out = model(lat, torch.tensor([749], device=DEV), [torch.from_numpy(g["ctx"]).to(DEV)], 420,
                    frame_split_indices=[3], ground_frame_indices=[(3, 4)])
    finally:
        model.skip_source_frames = 0
    full = model(lat, torch.tensor([749], device=DEV), [torch.from_numpy(g["ctx"]).to(DEV)], 420,
                 frame_split_indices=[3], ground_frame_indices=[(3, 4)])
    assert float(out[:, :, :3].abs().max()) == 0.0
    assert rel_l2(out[:, :, 3:], full[:, :, 3:].cpu()) < 1e-6


def test_unipc_12_step_trajectory_on_device(golden):
    """The scheduler's update on CUDA tensors is one fused wan_lincomb per step; 12 steps at shift 5 against the
    trajectory captured from the reference scheduler (fixture g7b), fp32 and bf16 latents."""
    g = golden("dit_g7b_unipc12")
    for dtype, tol in ((torch.float32, 2e-6), (torch.bfloat16, 3e-2)):
        s = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, solver_order=2)
        s.set_timesteps(12, device=DEV, shift=5.0)
        cur = torch.from_numpy(g["x"]).to(DEV, dtype)
        for i, t in enumerate(s.timesteps):
            cur = s.step(torch.from_numpy(g["v"][i]).to(DEV, dtype), t, cur, return_dict=False)[0]
            assert cur.dtype == dtype and cur.is_cuda
            if dtype == torch.float32:
                assert rel_l2(cur, g["traj"][i]) < tol, i
        assert rel_l2(cur.float(), g["traj"][11]) < tol


def test_unipc_device_path_equals_host_path_over_configurations():
    """The scheduler's CUDA path (one fused wan_lincomb per update) against its own torch path on CPU tensors -- which
    tests/test_oracle_vs_reference.py pins to the reference over 648 configurations -- for every configuration whose trajectory is finite:
    steps x shift x solver order x lower_order_final x disable_corrector x bh1 / bh2, fp32 latents."""
    import itertools
    g = torch.Generator().manual_seed(0)
    compared = 0
    for steps, shift, order, lof, dc, st in itertools.product([1, 2, 4, 7], [1.0, 3.0], [1, 2, 3], [True, False], [[], [0], [1, 2]], ["bh1", "bh2"]):
        kw = dict(num_train_timesteps=1000, shift=1, solver_order=order, lower_order_final=lof, disable_corrector=dc, solver_type=st)
        host, dev = FlowUniPCMultistepScheduler(**kw), FlowUniPCMultistepScheduler(**kw)
        host.set_timesteps(steps, device="cpu", shift=shift)
        dev.set_timesteps(steps, device=DEV, shift=shift)
        a = torch.randn(1, 16, 2, 6, 6, generator=g)
        b = a.to(DEV)
        ok = True
        for th, td in zip(host.timesteps, dev.timesteps):
            v = torch.randn(1, 16, 2, 6, 6, generator=g)
            try:
                a = host.step(v, th, a, return_dict=False)[0]
            except (ZeroDivisionError, np.linalg.LinAlgError):      # (degenerate at the final sigma = 0: the reference is not finite / solvable there either)
                ok = False
                break
            if not torch.isfinite(a).all():
                ok = False
                break
            b = dev.step(v.to(DEV), td, b, return_dict=False)[0]
            assert rel_l2(b, a) < 1e-5, (steps, shift, order, lof, dc, st)
        compared += ok
    assert compared >= 60, compared


def test_unipc_order_3_and_bh1_on_device(golden):
    """solver_order 3 / solver_type bh1 on CUDA tensors (updates of up to five terms: four go through the fused wan_lincomb, the
    five-term corrector through the fp32 torch chain) against the trajectories captured from the reference (fixture g7c)."""
    g = golden("dit_g7c_unipc_orders")
    for tag, order, st in (("o3_bh2", 3, "bh2"), ("o2_bh1", 2, "bh1")):
        s = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, solver_order=order, solver_type=st)
        s.set_timesteps(9, device=DEV, shift=5.0)
        cur = torch.from_numpy(g["x"]).to(DEV)
        for i, t in enumerate(s.timesteps):
            cur = s.step(torch.from_numpy(g["v"][i]).to(DEV), t, cur, return_dict=False)[0]
            if np.isfinite(g[f"traj_{tag}"][i]).all():          # (the reference's own last bh1 step is NaN: fm_solvers_unipc.py:473)
                assert rel_l2(cur, g[f"traj_{tag}"][i]) < 5e-6, (tag, i)
        assert torch.isfinite(cur).all()


def test_merge_lora_in_place_on_the_loaded_model(golden, tmp_path):
    """merge_lora(pipeline, path, multiplier, ...) with the reference's signature (lora_utils.py:371), applied to the
    packed device weights: the merged weights equal fixture g12 (captured from the reference's merge_lora on the same
    LoRA, three key styles), the forward changes, unmerge restores the original output up to bf16 rounding."""
    from types import SimpleNamespace
    from safetensors.torch import save_file
    from videocof_amd.lora_utils import merge_lora, unmerge_lora
    g = golden("dit_g12_lora")
    sd = deterministic_dit_state_dict(**TINY)
    m = WanTransformer3DModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    m.load_state_dict(sd, device=DEV)
    C, r = 256, 4
    lora = {
        "diffusion_model.blocks.0.self_attn.q.lora_down.weight": det_uniform("l.a.down", (r, C), 0.3),
        "diffusion_model.blocks.0.self_attn.q.lora_up.weight": det_uniform("l.a.up", (C, r), 0.3),
        "diffusion_model.blocks.0.self_attn.q.alpha": torch.tensor(2.0),
        "blocks.1.ffn.0.lora_A.default.weight": det_uniform("l.b.down", (r, C), 0.3),
        "blocks.1.ffn.0.lora_B.default.weight": det_uniform("l.b.up", (512, r), 0.3),
        "lora_unet__blocks_1_cross_attn_o.lora_down.weight": det_uniform("l.c.down", (r, C), 0.3),
        "lora_unet__blocks_1_cross_attn_o.lora_up.weight": det_uniform("l.c.up", (C, r), 0.3),
        "lora_unet__blocks_1_cross_attn_o.alpha": torch.tensor(8.0),
        "lora_te_text_model_encoder_layers_0_mlp_fc1.lora_down.weight": torch.zeros(r, 8),
    }
    save_file({k: v.contiguous() for k, v in lora.items()}, str(tmp_path / "lora.safetensors"))
    pipe = SimpleNamespace(transformer=m)
    lat = det_uniform("ml.lat", (1, 16, 3, 8, 8), 1.0).to(DEV)
    ctx = [det_uniform("ml.ctx", (9, 64), 1.0).to(DEV)]
    t = torch.tensor([321], device=DEV)
    before = m(lat, t, ctx, 48)
    assert merge_lora(pipe, str(tmp_path / "lora.safetensors"), 0.75, device=DEV, dtype=torch.bfloat16) is pipe
    w = m.linear_weights()
    for name, key in (("blocks.0.self_attn.q", "q"), ("blocks.1.ffn.0", "ffn0"), ("blocks.1.cross_attn.o", "o"),
                      ("blocks.0.self_attn.k", "untouched")):
        assert rel_l2(w[name], g[key]) < 3e-3, name              # bf16 storage of the fp32 reference merge
    merged = m(lat, t, ctx, 48)
    assert rel_l2(merged, before.cpu()) > 1e-3
    unmerge_lora(pipe, None, 0.75, state_dict=lora)
    assert rel_l2(m(lat, t, ctx, 48), before.cpu()) < 1e-2


def test_module_surface_state_dict_partial_load_and_pipeline_to(model):
    """What fast_infer.py does around the models (:280-362): `m, u = model.load_state_dict(ckpt, strict=False)` with a
    partial fine-tuned checkpoint on top of from_pretrained, `.to()`, `.eval()`, `pipeline.to(device)`, the offload
    switches; plus state_dict() round trip under the reference's key names."""
    sd = deterministic_dit_state_dict(**TINY)
    m = WanTransformer3DModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    assert m.state_dict() == {}
    r = m.load_state_dict(sd, device=DEV)
    assert tuple(r) == ([], []) and r.missing_keys == []
    got = m.state_dict()
    assert set(got) == set(sd)
    for k, v in sd.items():
        assert got[k].shape == v.shape, k
        assert rel_l2(got[k].float(), v) < 4e-3 or float(v.abs().max()) == 0, k
    lat = det_uniform("ms.lat", (1, 16, 3, 8, 8), 1.0).to(DEV)
    ctx = [det_uniform("ms.ctx", (9, 64), 1.0).to(DEV)]
    t = torch.tensor([321], device=DEV)
    base = m(lat, t, ctx, 48)
    assert torch.equal(base, model(lat, t, ctx, 48))
    # partial checkpoint: only two tensors, strict=False -> the rest keeps its values, result is unpackable
    part = {"blocks.1.ffn.0.weight": sd["blocks.1.ffn.0.weight"] * 1.5, "blocks.0.self_attn.q.bias": sd["blocks.0.self_attn.q.bias"] + 0.2,
            "not.a.key": torch.zeros(1)}
    missing, unexpected = m.load_state_dict(part, strict=False)
    assert unexpected == ["not.a.key"] and len(missing) == len(sd) - 2
    sd2 = dict(sd); sd2.update({k: v for k, v in part.items() if k in sd})
    ref = O.dit_forward(sd2, CFG, lat.cpu(), t.cpu(), [c.cpu() for c in ctx], 48)
    out = m(lat, t, ctx, 48)
    assert rel_l2(out, ref) < 1e-2 and rel_l2(out, base.cpu()) > 1e-3
    with pytest.raises(KeyError, match="unexpected"):
        m.load_state_dict(dict(sd, extra=torch.zeros(1)), strict=True)
    assert m.eval() is m and m.to(torch.bfloat16) is m
    pipe = WanPipeline(transformer=m, scheduler=FlowUniPCMultistepScheduler(shift=1))
    assert pipe.to(device=DEV) is pipe and pipe.to("cuda") is pipe
    assert pipe.enable_model_cpu_offload(device=DEV) is None and pipe.enable_sequential_cpu_offload(device=DEV) is None
    with pytest.raises(RuntimeError, match="loaded on"):
        pipe.to("cpu")


def test_ingest_of_a_14b_width_sharded_checkpoint_with_three_rank128_loras():
    """SURVEY 8f-2 at the real WIDTH (the full 40-layer run is tools/bench_ingest.py, profiles/r06/ingest_14b.json): a sharded bf16
    safetensors directory with config.json goes through from_pretrained (wan_transformer3d.py:1157-1299), three rank-128 LoRA files
    with ComfyUI names through merge_lora (lora_utils.py:371-500; the three merges of fast_infer.py:366-386); sampled rows of sampled
    Linears must equal W0 + sum m * alpha / r * up @ down evaluated in fp64 from the files (one bf16 rounding per merge call)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_ingest.py"), "--layers", "2", "--shards", "3"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["check"]["ok"] and d["check"]["sampled_linears"] == 6 and d["lora_layers_merged"] == [20, 20, 20]
    assert d["load_s"] > 0 and d["merge_s"] > 0 and d["checkpoint_bytes"] > 1.4e9
