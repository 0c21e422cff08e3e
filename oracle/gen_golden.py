#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

    python oracle/gen_golden.py            # writes tests/golden/dit_*.npz

The reference's Python is imported from /root/reference through
``oracle/ref_import.py`` (diffusers stubs + load-by-path); weights and inputs come
from the integer-hash fill in ``videocof_amd/weights.py`` so the fixtures hold only
inputs and expected outputs.  Golden numbering follows SURVEY.md section 8c.
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_import import load_reference          # noqa: E402
from videocof_amd.weights import deterministic_dit_state_dict, det_uniform  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

TINY = dict(dim=256, ffn_dim=512, num_layers=2, in_dim=16, out_dim=16, text_dim=64, freq_dim=256)
TINY_HEADS = 2


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def build_ref_model(ns, sd):
    m = ns.transformer.WanTransformer3DModel(
        model_type="t2v", dim=TINY["dim"], ffn_dim=TINY["ffn_dim"], num_heads=TINY_HEADS,
        num_layers=TINY["num_layers"], text_dim=TINY["text_dim"], in_dim=16, out_dim=16,
        freq_dim=TINY["freq_dim"], cross_attn_norm=True, qk_norm=True)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return m.eval()


@torch.no_grad()
def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ns = load_reference()
    T = ns.transformer
    sd = deterministic_dit_state_dict(**TINY)
    model = build_ref_model(ns, sd)
    C, H, D = TINY["dim"], TINY_HEADS, TINY["dim"] // TINY_HEADS

    # (1) sinusoidal embedding, fm timesteps of the 4-step schedule
    t = torch.tensor([999, 899, 749, 499])
    save("dit_g1_sinusoid", t=t, out=T.sinusoidal_embedding_1d(256, t))

    # (2) RoPE table rows + checksum
    fr = model.freqs
    rows = [0, 1, 2, 21, 22, 1023]
    save("dit_g2_freqs", rows=np.array(rows), real=fr.real[rows], imag=fr.imag[rows],
         sum_real=fr.real.sum(), sum_imag=fr.imag.sum(), shape=np.array(fr.shape))

    # (3) rope_apply in the three modes; 4 trailing pad rows must pass through
    grid = torch.tensor([[7, 4, 4]])
    L = 7 * 4 * 4
    x = det_uniform("g3.x", (1, L + 4, H, D), 1.0)
    save("dit_g3_rope", x=x, grid=grid,
         default=T.rope_apply(x, grid, fr),
         paired=T.rope_apply(x, grid, fr, frame_split_indices=[3]),
         cof=T.rope_apply(x, grid, fr, frame_split_indices=[3], ground_frame_indices=[(3, 4)]),
         cof_g2=T.rope_apply(x, grid, fr, frame_split_indices=[2], ground_frame_indices=[(2, 4)]))

    # (4) norms / head / unpatchify
    blk = model.blocks[0]
    xin = det_uniform("g4.x", (1, 37, C), 2.0)
    e6 = det_uniform("g4.e", (1, 6, C), 0.5)
    ehead = det_uniform("g4.eh", (1, C), 0.5)
    e = (blk.modulation + e6).chunk(6, dim=1)
    u = det_uniform("g4.u", (1, 7 * 3 * 5, 64), 1.0)
    save("dit_g4_norms", x=xin, e6=e6, ehead=ehead,
         rms_q=blk.self_attn.norm_q(xin),
         ln_mod=blk.norm1(xin) * (1 + e[1]) + e[0],
         ln_affine=blk.norm3(xin),
         head=model.head(xin, ehead),
         u=u, unpatch=model.unpatchify(u, torch.tensor([[7, 3, 5]]))[0])

    # (5) one WanAttentionBlock, ragged L (not a multiple of 64), CoF indices
    g5 = (7, 6, 10)
    L5 = math.prod(g5)
    x5 = det_uniform("g5.x", (1, L5, C), 1.0)
    e5 = det_uniform("g5.e", (1, 6, C), 0.5)
    ctx5 = det_uniform("g5.ctx", (1, 512, C), 1.0)
    y5 = blk(x5, e=e5, seq_lens=torch.tensor([L5]), grid_sizes=torch.tensor([g5]), freqs=fr,
             context=ctx5, context_lens=None, dtype=torch.float32,
             frame_split_indices=[3], ground_frame_indices=[(3, 4)])
    save("dit_g5_block", x=x5, e=e5, ctx=ctx5, grid=np.array(g5), out=y5)

    # (6) tiny full forward with CoF indices (B=1) and a B=2 call with different prompts
    lat = det_uniform("g6.lat", (1, 16, 7, 12, 20), 1.0)
    ctx = [det_uniform("g6.ctx", (37, TINY["text_dim"]), 1.0)]
    seq_len = 7 * 6 * 10
    out6 = model(lat, t=torch.tensor([899]), context=ctx, seq_len=seq_len,
                 frame_split_indices=[3], ground_frame_indices=[(3, 4)])
    out6_t2v = model(lat, t=torch.tensor([499]), context=ctx, seq_len=seq_len)
    lat2 = torch.cat([lat, det_uniform("g6.lat2", (1, 16, 7, 12, 20), 1.0)])
    ctx2 = [ctx[0], det_uniform("g6.ctx2", (11, TINY["text_dim"]), 1.0)]
    out6_b2 = model(lat2, t=torch.tensor([749, 749]), context=ctx2, seq_len=seq_len,
                    frame_split_indices=[3, 3], ground_frame_indices=[(3, 4), (3, 4)])
    save("dit_g6_forward", lat=lat, ctx=ctx[0], lat2=lat2, ctx2=ctx2[1],
         out_cof=out6, out_t2v=out6_t2v, out_b2=out6_b2)

    # (7) UniPC 4-step trajectory on fixed model outputs
    sch = ns.unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, solver_order=2,
                                               prediction_type="flow_prediction")
    sch.set_timesteps(4, device="cpu", shift=3)
    xs = det_uniform("g7.x", (1, 16, 5, 6, 6), 1.0)
    vs = [det_uniform(f"g7.v{i}", (1, 16, 5, 6, 6), 1.0) for i in range(4)]
    traj, orders = [], []
    cur = xs
    for i, tt in enumerate(sch.timesteps):
        cur = sch.step(vs[i], tt, cur, return_dict=False)[0]
        orders.append(sch.this_order)
        traj.append(cur)
    save("dit_g7_unipc", x=xs, v=torch.stack(vs), traj=torch.stack(traj),
         timesteps=sch.timesteps, sigmas=sch.sigmas, orders=np.array(orders))
    # 50-step schedule (inference.py default shift 5) -- schedule only
    sch.set_timesteps(50, device="cpu", shift=5.0)
    save("dit_g7_sched50", timesteps=sch.timesteps, sigmas=sch.sigmas)

    # (8) 4-step CoF denoise loop = reference DiT + reference UniPC + the glue of
    #     pipeline_wan.py:694-740 (source_frames 9 -> cc 3, reasoning_frames 4 -> G 1)
    src = det_uniform("g8.src", (1, 16, 3, 12, 20), 1.0)
    noise = det_uniform("g8.noise", (1, 16, 4, 12, 20), 1.7)
    latents = torch.cat([src, noise], dim=2)
    cc, G = 3, 1
    sch = ns.unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, solver_order=2,
                                               prediction_type="flow_prediction")
    sch.set_timesteps(4, device="cpu", shift=3)
    steps = []
    for tt in sch.timesteps:
        v = model(x=latents, context=ctx, t=tt.expand(1), seq_len=seq_len,
                  frame_split_indices=[cc], ground_frame_indices=[(cc, cc + G)])
        v[:, :, :cc] = 0
        latents = sch.step(v, tt, latents, return_dict=False)[0]
        steps.append(latents)
    save("dit_g8_cof_loop", src=src, noise=noise, ctx=ctx[0], steps=torch.stack(steps))

    # (8b) same loop with classifier-free guidance (inference.py path, guidance 5 -> B=2)
    latents = torch.cat([src, noise], dim=2)
    neg = [det_uniform("g8.neg", (9, TINY["text_dim"]), 1.0)]
    sch.set_timesteps(3, device="cpu", shift=5.0)
    steps = []
    for tt in sch.timesteps:
        inp = torch.cat([latents] * 2)
        v = model(x=inp, context=neg + ctx, t=tt.expand(2), seq_len=seq_len,
                  frame_split_indices=[cc] * 2, ground_frame_indices=[(cc, cc + G)] * 2)
        vu, vt = v.chunk(2)
        v = vu + 5.0 * (vt - vu)
        v[:, :, :cc] = 0
        latents = sch.step(v, tt, latents, return_dict=False)[0]
        steps.append(latents)
    save("dit_g8b_cfg_loop", neg=neg[0], steps=torch.stack(steps))

    # (12) merge_lora of the reference (lora_utils.py:371-500) in the three key styles it accepts
    lu = ns.load_lora_utils()
    r = 4
    lora = {
        "diffusion_model.blocks.0.self_attn.q.lora_down.weight": det_uniform("l.a.down", (r, C), 0.3),
        "diffusion_model.blocks.0.self_attn.q.lora_up.weight": det_uniform("l.a.up", (C, r), 0.3),
        "diffusion_model.blocks.0.self_attn.q.alpha": torch.tensor(2.0),
        "blocks.1.ffn.0.lora_A.default.weight": det_uniform("l.b.down", (r, C), 0.3),
        "blocks.1.ffn.0.lora_B.default.weight": det_uniform("l.b.up", (TINY["ffn_dim"], r), 0.3),
        "lora_unet__blocks_1_cross_attn_o.lora_down.weight": det_uniform("l.c.down", (r, C), 0.3),
        "lora_unet__blocks_1_cross_attn_o.lora_up.weight": det_uniform("l.c.up", (C, r), 0.3),
        "lora_unet__blocks_1_cross_attn_o.alpha": torch.tensor(8.0),
        "lora_te_text_model_encoder_layers_0_mlp_fc1.lora_down.weight": torch.zeros(r, 8),   # must be ignored
    }
    model12 = build_ref_model(ns, sd)
    import types
    pipe = types.SimpleNamespace(transformer=model12)
    lu.merge_lora(pipe, None, 0.75, device="cpu", dtype=torch.float32, state_dict=dict(lora), transformer_only=True)
    msd = model12.state_dict()
    save("dit_g12_lora", multiplier=0.75,
         q=msd["blocks.0.self_attn.q.weight"], ffn0=msd["blocks.1.ffn.0.weight"], o=msd["blocks.1.cross_attn.o.weight"],
         untouched=msd["blocks.0.self_attn.k.weight"])

    # (11) sequence-parallel RoPE slice of the reference (non-CoF), rank r of 2
    xs_ = det_uniform("g11.x", (1, L // 2, H, D), 1.0)
    outs = []
    for r in range(2):
        nsr = load_reference(sp_rank=r, sp_size=2)
        outs.append(nsr.wan_xfuser.rope_apply(xs_, grid, fr))
    save("dit_g11_sp_rope", x=xs_, grid=grid, rank0=outs[0], rank1=outs[1])


if __name__ == "__main__":
    main()
