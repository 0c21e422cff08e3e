"""Live cross-check of the CPU oracles against the REFERENCE itself on fresh random inputs (not the
committed fixtures).  Runs only where /root/reference is mounted (the build container); skipped elsewhere -- the
reference never travels to the GPU box.  fp32 vs fp32: rel-L2 <= 1e-5."""
import pytest
import torch

from oracle.ref_import import load_reference, reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference tree not mounted")


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def ns():
    torch.set_num_threads(8)
    return load_reference()


@torch.no_grad()
def test_dit_forward_random_inputs(ns):
    """Another grid, another source / ground split (two grounding frames), B = 2 with different prompts."""
    from oracle import wan_oracle as O
    from videocof_amd.weights import deterministic_dit_state_dict
    cfgd = dict(dim=384, ffn_dim=640, num_layers=2, in_dim=16, out_dim=16, text_dim=96, freq_dim=256)
    sd = deterministic_dit_state_dict(**cfgd)
    m = ns.transformer.WanTransformer3DModel(model_type="t2v", dim=384, ffn_dim=640, num_heads=3, num_layers=2, text_dim=96,
                                             in_dim=16, out_dim=16, freq_dim=256, cross_attn_norm=True, qk_norm=True)
    m.load_state_dict(sd, strict=True)
    m.eval()
    g = torch.Generator().manual_seed(123)
    lat = torch.randn(2, 16, 8, 6, 10, generator=g)
    ctx = [torch.randn(17, 96, generator=g), torch.randn(3, 96, generator=g)]
    t = torch.tensor([612, 612])
    seq_len = 8 * 3 * 5
    ref = m(lat, t=t, context=ctx, seq_len=seq_len, frame_split_indices=[3, 3], ground_frame_indices=[(3, 5), (3, 5)])
    cfg = O.DiTConfig(dim=384, ffn_dim=640, num_heads=3, num_layers=2, text_dim=96)
    out = O.dit_forward(sd, cfg, lat, t, ctx, seq_len, [3, 3], [(3, 5), (3, 5)])
    assert rel_l2(out, ref) < 1e-5
    ref_pad = m(lat[:1], t=t[:1], context=ctx[:1], seq_len=seq_len + 8)                  # padded sequence, plain T2V
    assert rel_l2(O.dit_forward(sd, cfg, lat[:1], t[:1], ctx[:1], seq_len + 8), ref_pad) < 1e-5
    # the layouts tests/test_gpu_dit.py::test_forward_random_layouts_vs_oracle runs the HIP path on (same generator, same seed): odd
    # grids, t2v / paired (a split without grounding frames) / CoF with one or more grounding frames, batch 1 / 2, prompt lengths
    import random
    rnd = random.Random(7)
    for case in range(14):
        F, Hl, Wl = rnd.randint(1, 9), 2 * rnd.randint(1, 7), 2 * rnd.randint(1, 9)
        B = rnd.choice([1, 1, 2])
        L = F * (Hl // 2) * (Wl // 2)
        rnd.choice([0, 0, 1, 5, 64])                 # (the GPU test's seq_len padding draw: keeps the two streams in step)
        mode = ("t2v", "paired", "cof", "cof")[case % 4] if F >= 3 else "t2v"
        fsi = gfi = None
        if mode != "t2v":
            fs = rnd.randint(1, F - 2)
            fsi = [fs] * B
            if mode == "cof":
                gfi = [(fs, fs + rnd.randint(1, F - 1 - fs))] * B
        lat_c = torch.randn(B, 16, F, Hl, Wl, generator=g)
        ctx_c = [torch.randn(rnd.randint(1, 77), 96, generator=g) for _ in range(B)]
        t_c = torch.tensor([rnd.choice([999, 899, 749, 499, 37])] * B)
        want = m(lat_c, t=t_c, context=ctx_c, seq_len=L, frame_split_indices=fsi, ground_frame_indices=gfi)
        assert rel_l2(O.dit_forward(sd, cfg, lat_c, t_c, ctx_c, L, fsi, gfi), want) < 1e-5, (case, (B, F, Hl, Wl), mode, fsi, gfi)


@torch.no_grad()
def test_unipc_random_trajectory(ns):
    from oracle import wan_oracle as O
    g = torch.Generator().manual_seed(7)
    for steps, shift in ((7, 3.0), (20, 5.0)):
        sch = ns.unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, solver_order=2,
                                                   prediction_type="flow_prediction")
        sch.set_timesteps(steps, device="cpu", shift=shift)
        s = O.UniPCOracle()
        s.set_timesteps(steps, shift)
        assert s.timesteps.tolist() == sch.timesteps.tolist()
        x = torch.randn(1, 16, 2, 4, 4, generator=g)
        a, b = x, x
        for tt in sch.timesteps:
            v = torch.randn(1, 16, 2, 4, 4, generator=g)
            a = sch.step(v, tt, a, return_dict=False)[0]
            b = s.step(v, b)
            assert rel_l2(b, a) < 1e-5


@torch.no_grad()
def test_t5_random_tokens(ns):
    from oracle import t5_oracle as T
    from videocof_amd.weights import deterministic_t5_state_dict
    cfg = dict(vocab=301, dim=192, dim_attn=192, dim_ffn=448, num_heads=3, num_layers=3, num_buckets=32)
    sd = deterministic_t5_state_dict(**cfg)
    m = ns.load_text_encoder().WanT5EncoderModel(shared_pos=False, dropout=0.0, **cfg).eval()
    m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, 301, (2, 150), generator=g)          # spans the logarithmic buckets (offsets > 128)
    mask = torch.ones(2, 150, dtype=torch.long)
    mask[1, 40:] = 0
    ref = m(ids, mask)[0]
    out = T.T5EncoderOracle(sd, 3, 3, 32).forward(ids, mask)
    assert rel_l2(out, ref) < 1e-5


@torch.no_grad()
def test_vae_random_clip(ns):
    from oracle.vae_oracle import WanVAEOracle
    from videocof_amd.weights import deterministic_vae_state_dict
    sd = deterministic_vae_state_dict()
    vae = ns.vae.AutoencoderKLWan()
    vae.load_state_dict(sd, strict=True)
    vae.eval()
    g = torch.Generator().manual_seed(5)
    video = torch.rand(1, 3, 9, 16, 24, generator=g) * 2 - 1
    orc = WanVAEOracle(sd)
    ref_mu = vae.encode(video)[0].mode()
    assert rel_l2(orc.encode(video[0])[:16], ref_mu[0]) < 1e-5
    z = torch.randn(1, 16, 2, 2, 3, generator=g)
    assert rel_l2(orc.decode(z[0]), vae.decode(z).sample[0]) < 1e-5
    # a single image, a clip whose tail does not fill a chunk, 13 frames, odd latent planes (the shapes tests/test_gpu_vae.py runs the HIP path at)
    for T, H, W in ((1, 16, 24), (2, 16, 16), (13, 24, 16), (9, 24, 40)):
        video = torch.rand(1, 3, T, H, W, generator=g) * 2 - 1
        ref_mu = vae.encode(video)[0].mode()
        assert rel_l2(orc.encode(video[0])[:16], ref_mu[0]) < 1e-5, (T, H, W)
        z = torch.randn(1, 16, ref_mu.shape[2], H // 8, W // 8, generator=g)
        assert rel_l2(orc.decode(z[0]), vae.decode(z).sample[0]) < 1e-5, (T, H, W)


@torch.no_grad()
def test_product_host_code_vs_reference(ns):
    """The two pieces of product code that run on the host: the UniPC scheduler (CPU tensors take the torch path, CUDA
    tensors the fused wan_lincomb) and the LoRA merge on state dicts -- against the live reference."""
    import types
    from videocof_amd import FlowUniPCMultistepScheduler
    from videocof_amd.lora_utils import merge_lora_state_dict
    from videocof_amd.weights import deterministic_dit_state_dict
    g = torch.Generator().manual_seed(21)
    ref = ns.unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, solver_order=2, prediction_type="flow_prediction")
    mine = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, solver_order=2)
    for steps, shift in ((9, 3.0), (25, 5.0)):
        ref.set_timesteps(steps, device="cpu", shift=shift)
        mine.set_timesteps(steps, device="cpu", shift=shift)
        assert torch.equal(mine.timesteps, ref.timesteps) and torch.equal(mine.sigmas, ref.sigmas)
        a = b = torch.randn(1, 16, 2, 4, 4, generator=g)
        for tt in ref.timesteps:
            v = torch.randn(1, 16, 2, 4, 4, generator=g)
            a = ref.step(v, tt, a, return_dict=False)[0]
            b = mine.step(v, tt, b, return_dict=False)[0]
            assert rel_l2(b, a) < 1e-5
    # add_noise / index_for_timestep / len (the diffusers scheduler surface around `step`): before a loop (looked up by timestep),
    # and inside one (the current step's sigma)
    x0, nz = torch.randn(2, 16, 2, 4, 4, generator=g), torch.randn(2, 16, 2, 4, 4, generator=g)
    for sch in (ref, mine):
        sch.set_timesteps(9, device="cpu", shift=3.0)
    tsel = ref.timesteps[[2, 5]]
    assert torch.equal(mine.add_noise(x0, nz, tsel), ref.add_noise(x0, nz, tsel))
    assert mine.index_for_timestep(ref.timesteps[4]) == ref.index_for_timestep(ref.timesteps[4]) == 4
    assert mine.index_for_timestep(ref.timesteps[4], ref.timesteps[2:]) == ref.index_for_timestep(ref.timesteps[4], ref.timesteps[2:]) == 2
    assert len(mine) == len(ref) == 1000
    # .config holds every constructor argument, by attribute and by key, as @register_to_config's does; from_config round-trips it
    assert dict(mine.config) == {k: v for k, v in vars(ref.config).items() if not k.startswith("_")}      # (the import shim's config is a namespace)
    assert mine.config.solver_type == mine.config["solver_type"] == "bh2"
    again = FlowUniPCMultistepScheduler.from_config(mine.config, shift=3.0, not_an_argument=1)
    assert again.config.shift == 3.0 and again.config.solver_order == 2
    for sch in (ref, mine):
        sch.set_begin_index(3)
    assert torch.equal(mine.add_noise(x0, nz, tsel), ref.add_noise(x0, nz, tsel))
    cfgd = dict(dim=256, ffn_dim=512, num_layers=2, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
    sd = deterministic_dit_state_dict(**cfgd)
    m = ns.transformer.WanTransformer3DModel(model_type="t2v", dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64,
                                             in_dim=16, out_dim=16, freq_dim=256, cross_attn_norm=True, qk_norm=True)
    m.load_state_dict(sd, strict=True)
    from videocof_amd import WanTransformer3DModel
    mine_m = WanTransformer3DModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    want_cfg = {k: (tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in vars(m.config).items() if not k.startswith("_")}
    assert vars(mine_m.config) == want_cfg                    # every constructor argument, as @register_to_config holds them
    lora = {}
    for name, (o, i) in {"blocks.1.self_attn.v": (256, 256), "blocks.0.ffn.2": (256, 512), "blocks.1.cross_attn.k": (256, 256)}.items():
        lora[f"diffusion_model.{name}.lora_down.weight"] = torch.randn(8, i, generator=g) * 0.1
        lora[f"diffusion_model.{name}.lora_up.weight"] = torch.randn(o, 8, generator=g) * 0.1
        lora[f"diffusion_model.{name}.alpha"] = torch.tensor(4.0)
    ns.load_lora_utils().merge_lora(types.SimpleNamespace(transformer=m), None, 0.6, device="cpu", dtype=torch.float32,
                                    state_dict=dict(lora), transformer_only=True)
    want = m.state_dict()
    got = {k: v.clone() for k, v in sd.items()}
    assert merge_lora_state_dict(got, lora, 0.6) == 3
    for k in want:
        assert rel_l2(got[k], want[k]) < 1e-6, k


@torch.no_grad()
def test_lora_key_conventions_take_part_exactly_as_in_the_reference(ns):
    """WHICH entries of a LoRA file are merged follows from the reference's renaming rules (lora_utils.py:378-395) -- including the
    entries they silently drop: `diffusion_model.`-style names of modules outside the blocks (head, text / time embeddings), `lora_A`
    names without `.default.`.  Every spelling below goes through the reference's merge_lora on the reference's model and through
    merge_lora_state_dict on the same state dict: the same tensors change, by the same amounts."""
    import types
    from videocof_amd.lora_utils import merge_lora_state_dict
    from videocof_amd.weights import deterministic_dit_state_dict
    g = torch.Generator().manual_seed(3)
    cfgd = dict(dim=256, ffn_dim=512, num_layers=2, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
    base = deterministic_dit_state_dict(**cfgd)

    def pair(name, o, i, style, alpha=2.0, r=4):
        stem = {"dm": f"diffusion_model.{name}", "kohya": "lora_unet__" + name.replace(".", "_"), "kohya1": "lora_unet_" + name.replace(".", "_"),
                "peft": name, "peft_dm": f"diffusion_model.{name}", "peft_nodefault": name}[style]
        dn, up = {"peft": (".lora_A.default.weight", ".lora_B.default.weight"), "peft_dm": (".lora_A.default.weight", ".lora_B.default.weight"),
                  "peft_nodefault": (".lora_A.weight", ".lora_B.weight")}.get(style, (".lora_down.weight", ".lora_up.weight"))
        d = {stem + dn: torch.randn(r, i, generator=g) * 0.1, stem + up: torch.randn(o, r, generator=g) * 0.1}
        if alpha is not None and not style.startswith("peft"):
            d[stem + ".alpha"] = torch.tensor(alpha)
        return d
    cases = {
        "dm, blocks": ({**pair("blocks.0.self_attn.q", 256, 256, "dm"), **pair("blocks.1.ffn.0", 512, 256, "dm"), **pair("blocks.1.cross_attn.v", 256, 256, "dm", alpha=None)}, 3),
        "dm, outside the blocks: dropped": ({**pair("text_embedding.0", 256, 64, "dm"), **pair("time_embedding.2", 256, 256, "dm"), **pair("head.head", 64, 256, "dm"),
                                             **pair("time_projection.1", 1536, 256, "dm"), **pair("blocks.0.ffn.2", 256, 512, "dm")}, 1),
        "kohya, blocks and outside": ({**pair("blocks.0.cross_attn.o", 256, 256, "kohya"), **pair("head.head", 64, 256, "kohya"), **pair("text_embedding.2", 256, 256, "kohya"),
                                       **pair("blocks.1.self_attn.k", 256, 256, "kohya1", alpha=None)}, 4),
        "peft with .default.": ({**pair("blocks.0.self_attn.v", 256, 256, "peft"), **pair("blocks.1.ffn.2", 256, 512, "peft")}, 2),
        "peft without .default.: dropped": (pair("blocks.0.self_attn.q", 256, 256, "peft_nodefault"), 0),
        "text-encoder and unknown layers": ({**pair("blocks.0.self_attn.o", 256, 256, "dm"), **pair("blocks.7.self_attn.q", 256, 256, "dm"),
                                             "lora_te_text_model_encoder_layers_0_mlp_fc1.lora_down.weight": torch.zeros(4, 8),
                                             "lora_te_text_model_encoder_layers_0_mlp_fc1.lora_up.weight": torch.zeros(8, 4),
                                             "diffusion_model.blocks.0.self_attn.norm_q.diff": torch.zeros(256)}, 1),
    }
    for what, (lora, n_merged) in cases.items():
        m = ns.transformer.WanTransformer3DModel(model_type="t2v", dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64,
                                                 in_dim=16, out_dim=16, freq_dim=256, cross_attn_norm=True, qk_norm=True)
        m.load_state_dict(base, strict=True)
        ns.load_lora_utils().merge_lora(types.SimpleNamespace(transformer=m), None, 0.7, device="cpu", dtype=torch.float32,
                                        state_dict=dict(lora), transformer_only=True)
        want = m.state_dict()
        got = {k: v.clone() for k, v in base.items()}
        assert merge_lora_state_dict(got, dict(lora), 0.7) == n_merged, what
        assert sum(1 for k in want if not torch.equal(want[k], base[k])) == n_merged, what
        for k in want:
            assert rel_l2(got[k], want[k]) < 1e-6, (what, k)


def test_call_signatures_are_the_references():
    """Drop-in surface: every seam the reference calls has, on the mirror, the reference's
    parameter NAMES in the reference's ORDER with the reference's DEFAULTS -- read from the reference's source with `ast` (no import: its
    pipeline needs the real diffusers), compared with `inspect.signature` of the product classes.  What the mirrors may add: trailing
    parameters WITH defaults (listed below); what may differ in a default: listed below, each a widening (a float where the reference writes
    the same number as an int; None = "the model's device" where the reference says 'cpu'; optional components)."""
    import ast
    import inspect
    import os
    from oracle.ref_import import REFERENCE_ROOT as REF_ROOT
    import videocof_amd as V
    from videocof_amd import attention_utils, cache_utils, dist, fm_solvers_unipc, lora_utils, pipeline_wan, wan_text_encoder, wan_vae

    def ref_sig(path, cls, fn):
        with open(os.path.join(REF_ROOT, "videox_fun", path)) as f:
            body = ast.parse(f.read()).body
        if cls:
            body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls).body
        a = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == fn).args
        names = [x.arg for x in a.posonlyargs + a.args]
        defaults = [None] * (len(names) - len(a.defaults)) + list(a.defaults)
        out = [(n, None if d is None else ast.unparse(d)) for n, d in zip(names, defaults)]
        out += [(x.arg, None if d is None else ast.unparse(d)) for x, d in zip(a.kwonlyargs, a.kw_defaults)]
        return [(n, d) for n, d in out if n != "cls"]

    def my_sig(obj):
        ps = [p for p in inspect.signature(obj).parameters.values() if p.kind not in (p.VAR_POSITIONAL, p.VAR_KEYWORD)]
        return [(p.name, None if p.default is p.empty else repr(p.default)) for p in ps]

    seams = [
        ("models/wan_transformer3d.py", "WanTransformer3DModel", "forward", V.WanTransformer3DModel.forward, (), {}),
        ("models/wan_transformer3d.py", "WanTransformer3DModel", "__init__", V.WanTransformer3DModel.__init__, (), {}),
        ("models/wan_transformer3d.py", "WanTransformer3DModel", "from_pretrained", V.WanTransformer3DModel.from_pretrained, (), {}),
        ("models/wan_transformer3d.py", "WanTransformer3DModel", "enable_teacache", V.WanTransformer3DModel.enable_teacache, (), {}),
        ("models/attention_utils.py", None, "attention", attention_utils.attention, (), {}),
        ("models/attention_utils.py", None, "flash_attention", attention_utils.flash_attention, (), {}),
        ("utils/fm_solvers_unipc.py", "FlowUniPCMultistepScheduler", "__init__", fm_solvers_unipc.FlowUniPCMultistepScheduler.__init__, (), {}),
        ("utils/fm_solvers_unipc.py", "FlowUniPCMultistepScheduler", "set_timesteps", fm_solvers_unipc.FlowUniPCMultistepScheduler.set_timesteps, (), {}),
        ("utils/fm_solvers_unipc.py", "FlowUniPCMultistepScheduler", "step", fm_solvers_unipc.FlowUniPCMultistepScheduler.step, (), {}),
        ("pipeline/pipeline_wan.py", "WanPipeline", "__call__", pipeline_wan.WanPipeline.__call__,
         ("source_latents", "device", "weight_dtype", "cache_context", "skip_source_prediction", "capture_graph"),
         {"guidance_scale": ("6", "6.0"), "shift": ("5", "5.0"), "callback_on_step_end_tensor_inputs": ("['latents']", "('latents',)")}),
        ("pipeline/pipeline_wan.py", "WanPipeline", "__init__", pipeline_wan.WanPipeline.__init__, (),
         {k: (None, "None") for k in ("tokenizer", "text_encoder", "vae", "transformer", "scheduler")}),
        ("pipeline/pipeline_wan.py", "WanPipeline", "prepare_latents", pipeline_wan.WanPipeline.prepare_latents, (), {}),
        ("pipeline/pipeline_wan.py", "WanPipeline", "prepare_video_latents", pipeline_wan.WanPipeline.prepare_video_latents, (), {}),
        ("pipeline/pipeline_wan.py", "WanPipeline", "prepare_video_latents_new", pipeline_wan.WanPipeline.prepare_video_latents_new, ("source_latents",), {}),
        ("pipeline/pipeline_wan.py", "WanPipeline", "prepare_cot_video_latents", pipeline_wan.WanPipeline.prepare_cot_video_latents, ("source_latents",), {}),
        ("pipeline/pipeline_wan.py", "WanPipeline", "prepare_extra_step_kwargs", pipeline_wan.WanPipeline.prepare_extra_step_kwargs, (), {}),
        ("pipeline/pipeline_wan.py", "WanPipeline", "decode_latents", pipeline_wan.WanPipeline.decode_latents, ("out",), {}),
        ("pipeline/pipeline_wan.py", "WanPipeline", "check_inputs", pipeline_wan.WanPipeline.check_inputs, (), {}),
        ("pipeline/pipeline_wan.py", "WanPipeline", "encode_prompt", pipeline_wan.WanPipeline.encode_prompt, (), {}),
        ("pipeline/pipeline_wan.py", "WanPipeline", "_get_t5_prompt_embeds", pipeline_wan.WanPipeline._get_t5_prompt_embeds, (), {}),
        ("models/wan_transformer3d.py", "WanTransformer3DModel", "unpatchify", V.WanTransformer3DModel.unpatchify, (), {}),
        ("models/wan_transformer3d.py", "WanTransformer3DModel", "share_cfg_skip", V.WanTransformer3DModel.share_cfg_skip, (), {}),
        ("models/wan_transformer3d.py", "WanTransformer3DModel", "share_teacache", V.WanTransformer3DModel.share_teacache, (), {}),
        ("models/wan_transformer3d.py", "WanTransformer3DModel", "enable_cfg_skip", V.WanTransformer3DModel.enable_cfg_skip, (), {}),
        ("models/wan_transformer3d.py", "WanTransformer3DModel", "enable_riflex", V.WanTransformer3DModel.enable_riflex, (), {}),
        ("models/cache_utils.py", "TeaCache", "__init__", cache_utils.TeaCache.__init__, (), {}),
        ("utils/fm_solvers_unipc.py", "FlowUniPCMultistepScheduler", "index_for_timestep", fm_solvers_unipc.FlowUniPCMultistepScheduler.index_for_timestep, (), {}),
        ("utils/fm_solvers_unipc.py", "FlowUniPCMultistepScheduler", "add_noise", fm_solvers_unipc.FlowUniPCMultistepScheduler.add_noise, (), {}),
        ("utils/fm_solvers_unipc.py", "FlowUniPCMultistepScheduler", "scale_model_input", fm_solvers_unipc.FlowUniPCMultistepScheduler.scale_model_input, (), {}),
        ("utils/fm_solvers_unipc.py", "FlowUniPCMultistepScheduler", "set_begin_index", fm_solvers_unipc.FlowUniPCMultistepScheduler.set_begin_index, (), {}),
        ("models/wan_vae.py", "AutoencoderKLWan", "encode", wan_vae.AutoencoderKLWan.encode, (), {}),
        ("models/wan_vae.py", "AutoencoderKLWan", "decode", wan_vae.AutoencoderKLWan.decode, (), {}),
        ("models/wan_vae.py", "AutoencoderKLWan", "__init__", wan_vae.AutoencoderKLWan.__init__, (), {}),
        ("models/wan_vae.py", "AutoencoderKLWan", "from_pretrained", wan_vae.AutoencoderKLWan.from_pretrained, (), {}),
        ("models/wan_text_encoder.py", "WanT5EncoderModel", "forward", wan_text_encoder.WanT5EncoderModel.forward, (), {}),
        ("models/wan_text_encoder.py", "WanT5EncoderModel", "__init__", wan_text_encoder.WanT5EncoderModel.__init__, ("text_length",), {}),
        ("models/wan_text_encoder.py", "WanT5EncoderModel", "from_pretrained", wan_text_encoder.WanT5EncoderModel.from_pretrained, ("device",), {}),
        ("utils/lora_utils.py", None, "merge_lora", lora_utils.merge_lora, (), {"device": ("'cpu'", "None")}),
        ("utils/lora_utils.py", None, "unmerge_lora", lora_utils.unmerge_lora, ("state_dict",), {"device": ("'cpu'", "None")}),
        ("dist/fuser.py", None, "set_multi_gpus_devices", dist.set_multi_gpus_devices, (), {"ring_degree": (None, "1")}),
    ]
    for path, cls, fn, obj, extras, widened in seams:
        ref, mine = ref_sig(path, cls, fn), my_sig(obj)
        where = f"{cls + '.' if cls else ''}{fn} ({path})"
        mine_d = dict(mine)
        assert [n for n, _ in mine if n not in extras] == [n for n, _ in ref], (where, ref, mine)          # names and order
        assert [n for n, _ in mine][len(ref):] == list(extras), (where, "additions must trail the reference's parameters", mine)
        assert all(mine_d[n] is not None for n in extras), (where, "additions must be optional")
        for n, d in ref:
            if n in widened:
                assert (d, mine_d[n]) == widened[n], (where, n, d, mine_d[n])
            else:
                assert d == mine_d[n], (where, n, d, mine_d[n])


def test_public_methods_of_the_reference_classes_exist_on_the_mirrors():
    """Every method the reference defines on the classes of the path exists on the mirror, except the ones listed: training hooks,
    the private halves of the schedulers' / VAE's own arithmetic (the mirrors have their own), and the constructor-time initialiser."""
    import ast
    import os
    from oracle.ref_import import REFERENCE_ROOT
    import videocof_amd as V
    from videocof_amd import cache_utils, fm_solvers_unipc, pipeline_wan, wan_text_encoder, wan_vae

    def ref_methods(path, cls):
        with open(os.path.join(REFERENCE_ROOT, "videox_fun", path)) as f:
            c = next(n for n in ast.parse(f.read()).body if isinstance(n, ast.ClassDef) and n.name == cls)
        return [n.name for n in c.body if isinstance(n, ast.FunctionDef)]
    allowed = {
        "WanTransformer3DModel": {"_set_gradient_checkpointing", "init_weights"},       # training; fresh values come from weights.py
        "WanPipeline": set(),
        "AutoencoderKLWan": {"_encode", "_decode"},                                     # private halves of encode / decode
        "WanT5EncoderModel": set(),
        "FlowUniPCMultistepScheduler": {"_threshold_sample", "_sigma_to_t", "_sigma_to_alpha_sigma_t", "time_shift", "convert_model_output",
                                        "multistep_uni_p_bh_update", "multistep_uni_c_bh_update", "_init_step_index"},   # one fused update instead
        "TeaCache": set(),
    }
    for path, cls, obj in (("models/wan_transformer3d.py", "WanTransformer3DModel", V.WanTransformer3DModel),
                           ("pipeline/pipeline_wan.py", "WanPipeline", pipeline_wan.WanPipeline),
                           ("models/wan_vae.py", "AutoencoderKLWan", wan_vae.AutoencoderKLWan),
                           ("models/wan_text_encoder.py", "WanT5EncoderModel", wan_text_encoder.WanT5EncoderModel),
                           ("utils/fm_solvers_unipc.py", "FlowUniPCMultistepScheduler", fm_solvers_unipc.FlowUniPCMultistepScheduler),
                           ("models/cache_utils.py", "TeaCache", cache_utils.TeaCache)):
        missing = {m for m in ref_methods(path, cls) if not hasattr(obj, m)}
        assert missing == allowed[cls], (cls, sorted(missing - allowed[cls]), sorted(allowed[cls] - missing))


@torch.no_grad()
def test_unipc_configuration_sweep(ns):
    """648 scheduler configurations (steps x shift x solver order x lower_order_final x disable_corrector x bh1 / bh2): wherever the
    reference's own trajectory is finite (it is not for bh1 or lower_order_final=False at the final sigma = 0, and its linear solve fails
    for some), the mirror's timesteps / sigmas are equal and its trajectory matches to 1e-5; the mirror never fails where the reference runs."""
    import itertools
    from videocof_amd import FlowUniPCMultistepScheduler as Mine
    g = torch.Generator().manual_seed(0)
    equal = 0
    for steps, shift, order, lof, dc, st in itertools.product([1, 2, 3, 4, 7, 20], [1.0, 3.0, 5.0], [1, 2, 3], [True, False], [[], [0], [1, 2]],
                                                              ["bh1", "bh2"]):
        kw = dict(num_train_timesteps=1000, shift=1, solver_order=order, lower_order_final=lof, disable_corrector=dc, solver_type=st)
        r = ns.unipc.FlowUniPCMultistepScheduler(prediction_type="flow_prediction", **kw)
        m = Mine(**kw)
        r.set_timesteps(steps, device="cpu", shift=shift)
        m.set_timesteps(steps, device="cpu", shift=shift)
        assert torch.equal(m.timesteps, r.timesteps) and torch.equal(m.sigmas, r.sigmas)
        a = b = torch.randn(1, 4, 2, 3, 3, generator=g)
        ok = True
        for tt in r.timesteps:
            v = torch.randn(1, 4, 2, 3, 3, generator=g)
            try:
                a = r.step(v, tt, a, return_dict=False)[0]
            except Exception:
                ok = False
            if not ok or not torch.isfinite(a).all():
                ok = False
                break
            b = m.step(v, tt, b, return_dict=False)[0]
            assert rel_l2(b, a) < 1e-5, (steps, shift, order, lof, dc, st)
        equal += ok
    assert equal >= 200, equal          # (234 here: every bh2 / lower_order_final configuration among them)
