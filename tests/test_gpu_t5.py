"""-m gpu: the umT5 text encoder through the C ABI, against the CPU oracle
(oracle/t5_oracle.py) and the fixture captured from the reference's WanT5EncoderModel.

Tolerances: row kernels rel-L2 4e-3 (one bf16 rounding); fp32 outputs 1e-5; the whole encoder (bf16
weights and activations, fp32 residual stream) rel-L2 1.5e-2 / cosine 0.9998 against the fp32 reference.
"""
import pytest
import torch

from oracle import t5_oracle as T
from oracle.gen_golden_t5 import TINY
from videocof_amd import ops
from videocof_amd.wan_text_encoder import WanT5EncoderModel, relative_position_buckets
from videocof_amd.weights import deterministic_t5_state_dict, random_t5_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def cosine(a, b):
    a, b = a.detach().double().cpu().flatten(), torch.as_tensor(b).double().cpu().flatten()
    return float(a @ b / (a.norm() * b.norm()))


def bf(x):
    return x.to(torch.bfloat16)


def test_embedding_rows_and_range_check():
    g = torch.Generator().manual_seed(0)
    table = bf(torch.randn(97, 256, generator=g))
    ids = torch.randint(0, 97, (50,), generator=g)
    out = ops.embedding_rows(ids.to(DEV), table.to(DEV))
    assert out.dtype == torch.float32 and torch.equal(out.cpu(), table[ids].float())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.embedding_rows(ids, table)


@pytest.mark.parametrize("rows,dim", [(37, 256), (5, 4096), (3, 8192), (64, 1000)])
def test_rmsnorm_rows_vs_oracle(rows, dim):
    g = torch.Generator().manual_seed(rows + dim)
    x = torch.randn(rows, dim, generator=g) * 3
    w = torch.rand(dim, generator=g) + 0.5
    ref = T.t5_layer_norm(x, w)
    out = ops.rmsnorm_rows(x.to(DEV), w.to(DEV), 1e-6)
    assert out.dtype == torch.bfloat16 and rel_l2(out, ref) < 4e-3
    out32 = ops.rmsnorm_rows(x.to(DEV), w.to(DEV), 1e-6, out_dtype=torch.float32)
    assert rel_l2(out32, ref) < 1e-5


def test_mul_bf16():
    g = torch.Generator().manual_seed(2)
    a, b = bf(torch.randn(77, 320, generator=g)), bf(torch.randn(77, 320, generator=g))
    out = ops.mul_bf16(a.to(DEV), b.to(DEV))
    assert torch.equal(out.cpu(), bf(a.float() * b.float()))
    with pytest.raises(ValueError, match="multiple of 8"):
        ops.mul_bf16(a[:1, :4].contiguous().to(DEV), b[:1, :4].contiguous().to(DEV))


@pytest.mark.parametrize("H,L,D,K2", [(2, 72, 64, 128), (64, 512, 64, 512), (3, 132, 64, 192)])
def test_gemm_batched_heads(H, L, D, K2):
    """Both uses inside T5 attention: S_h = q_h k_h^T (operands interleaved by head inside packed rows) and
    o_h = P_h v_h (W = rows of V^T)."""
    g = torch.Generator().manual_seed(H * L)
    A = H * D
    qk = bf(torch.randn(L, 2 * A, generator=g)).to(DEV)
    S = torch.empty(H, L, L, device=DEV)
    ops.gemm_batched(qk[:, :D], D, qk[:, A:A + D], D, S[0], L * L, L, L, D, H, ops.EPI_F32)
    q = qk[:, :A].float().view(L, H, D)
    k = qk[:, A:].float().view(L, H, D)
    ref = torch.einsum("inc,jnc->nij", q, k)
    assert rel_l2(S, ref) < 1e-5
    Lp = ops.round_up(L, 64)
    P = torch.zeros(H, L, Lp, device=DEV, dtype=torch.bfloat16)
    P[:, :, :L] = bf(torch.rand(H, L, L, generator=g)).to(DEV)
    vt = torch.zeros(A, Lp, device=DEV, dtype=torch.bfloat16)
    vt[:, :L] = bf(torch.randn(A, L, generator=g)).to(DEV)
    o = torch.zeros(L, A, device=DEV, dtype=torch.bfloat16)
    ops.gemm_batched(P[0], L * Lp, vt[:D], D * Lp, o[:, :D], D, L, D, Lp, H, ops.EPI_BF16)
    ref_o = torch.einsum("nij,ncj->inc", P[:, :, :L].float(), vt[:, :L].float().view(H, D, L)).reshape(L, A)
    assert rel_l2(o, ref_o) < 4e-3
    with pytest.raises(ValueError, match="past the end"):
        ops.gemm_batched(P[0], L * Lp, vt[:D], D * Lp, o[:, :D], D, L, D, Lp, H + 1, ops.EPI_BF16)
    with pytest.raises(RuntimeError, match="epilogue"):
        ops.gemm_batched(P[0], L * Lp, vt[:D], D * Lp, o[:, :D], D, L, D, Lp, H, ops.EPI_GELU_BF16)


@pytest.mark.parametrize("H,L,klen", [(2, 72, 37), (2, 72, 72), (64, 512, 19), (5, 600, 333)])
def test_t5_softmax_bias_vs_oracle(H, L, klen):
    g = torch.Generator().manual_seed(L + klen)
    S = torch.randn(H, L, L, generator=g) * 3
    table = torch.randn(32, H, generator=g)
    lut = relative_position_buckets(L, 32)
    Lp = ops.round_up(L, 64)
    P = ops.t5_softmax_bias(S.to(DEV), table.to(DEV), lut.to(DEV), H, klen, Lp)
    rel = torch.arange(L)[None, :] - torch.arange(L)[:, None]
    bias = table[T.relative_position_bucket(rel, 32)].permute(2, 0, 1)
    bias = bias.masked_fill((torch.arange(L) >= klen)[None, None, :], torch.finfo(torch.float32).min)
    ref = torch.softmax(S + bias, dim=-1)
    assert P.shape == (H, L, Lp) and float(P[:, :, klen:].abs().max() if klen < Lp else 0) == 0.0
    assert rel_l2(P[:, :, :L], ref) < 4e-3
    assert float((P[:, :, :L].float().sum(-1).cpu() - 1).abs().max()) < 1e-2


@pytest.fixture(scope="module")
def tiny_model():
    m = WanT5EncoderModel(shared_pos=False, dropout=0.0, **TINY)
    m.load_state_dict(deterministic_t5_state_dict(**TINY), device=DEV)
    return m


def test_g13_encoder_vs_reference_fixture(golden, tiny_model):
    g = golden("t5_g13_encoder")
    ids, mask = torch.from_numpy(g["ids"]).to(DEV), torch.from_numpy(g["mask"]).to(DEV)
    out = tiny_model(ids, mask)[0]
    assert out.shape == (3, 72, TINY["dim"]) and out.dtype == torch.bfloat16
    lens = [37, 11, 72]
    for b, n in enumerate(lens):      # the rows the pipeline keeps (pipeline_wan.py:181)
        assert rel_l2(out[b, :n], g["out"][b, :n]) < 1.5e-2 and cosine(out[b, :n], g["out"][b, :n]) > 0.9998
    # padded query rows carry what the reference computes there (they see the valid keys only)
    assert rel_l2(out, g["out"]) < 1.5e-2
    out_nm = tiny_model(ids[:1], None)[0]
    assert rel_l2(out_nm, g["out_nomask"]) < 1.5e-2
    assert rel_l2(out_nm[0, :37], g["out"][0, :37]) > 1e-3          # the mask matters


def test_encoder_input_validation(tiny_model):
    ids = torch.zeros(1, 16, dtype=torch.long, device=DEV)
    with pytest.raises(IndexError):
        tiny_model(ids + 1000)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tiny_model(ids.cpu())
    m = torch.ones(1, 16, dtype=torch.long, device=DEV)
    m[0, 3] = 0
    with pytest.raises(NotImplementedError, match="prefix"):
        tiny_model(ids, m)
    with pytest.raises(NotImplementedError, match="shared_pos"):
        WanT5EncoderModel(shared_pos=True, **TINY)
    with pytest.raises(KeyError, match="missing"):
        WanT5EncoderModel(shared_pos=False, **TINY).load_state_dict({}, device=DEV)


def test_encoder_mid_size_vs_oracle_and_batch_independence():
    """dim 512 / 8 heads / 512-token padding as the pipeline uses it: vs the fp32 oracle evaluated on the same
    bf16-rounded weights; a sample's valid rows must not depend on its batch neighbours."""
    cfg = dict(vocab=1000, dim=512, dim_attn=512, dim_ffn=1280, num_heads=8, num_layers=3, num_buckets=32)
    sd = random_t5_state_dict("cpu", seed=1, **cfg)
    m = WanT5EncoderModel(shared_pos=False, **cfg)
    m.load_state_dict(sd, device=DEV)
    g = torch.Generator().manual_seed(0)
    L, lens = 512, [300, 41]
    ids = torch.randint(1, 1000, (2, L), generator=g)
    mask = torch.zeros(2, L, dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
        ids[b, n:] = 0
    out = m(ids.to(DEV), mask.to(DEV))[0]
    ref = T.T5EncoderOracle(sd, 8, 3, 32).forward(ids, mask)
    for b, n in enumerate(lens):
        assert rel_l2(out[b, :n], ref[b, :n]) < 1.5e-2 and cosine(out[b, :n], ref[b, :n]) > 0.9998
    solo = m(ids[1:].to(DEV), mask[1:].to(DEV))[0]
    assert torch.equal(solo[0, :41], out[1, :41])
    # a length that is not a multiple of 4 (internally padded with masked positions), no mask given
    short = m(ids[:1, :30].to(DEV))[0]
    assert short.shape == (1, 30, 512)
    assert rel_l2(short, T.T5EncoderOracle(sd, 8, 3, 32).forward(ids[:1, :30])) < 1.5e-2


class _ToyTokenizer:
    """Same call contract as the HuggingFace tokenizer the reference uses (pipeline_wan.py:154-163):
    padding to max_length with id 0, attention_mask, an end-of-sequence token."""
    def __init__(self, vocab):
        self.vocab = vocab

    def __call__(self, prompt, padding=None, max_length=512, truncation=True, add_special_tokens=True, return_tensors="pt"):
        assert padding == "max_length" and return_tensors == "pt"
        ids = torch.zeros(len(prompt), max_length, dtype=torch.long)
        mask = torch.zeros(len(prompt), max_length, dtype=torch.long)
        for b, p in enumerate(prompt):
            toks = [2 + (sum(map(ord, w)) % (self.vocab - 2)) for w in p.split()][: max_length - 1] + [1]
            ids[b, :len(toks)] = torch.tensor(toks)
            mask[b, :len(toks)] = 1
        from types import SimpleNamespace
        return SimpleNamespace(input_ids=ids, attention_mask=mask)


def test_pipeline_encodes_prompt_strings_with_the_hip_encoder():
    """WanPipeline(tokenizer, text_encoder, ...): prompt strings -> umT5 on the GPU -> DiT context
    (pipeline_wan.py:140-181, 595-608) == the same call fed with the oracle's embeddings of the same tokens."""
    from oracle import wan_oracle as O
    from videocof_amd import FlowUniPCMultistepScheduler, WanPipeline, WanTransformer3DModel
    from videocof_amd.weights import deterministic_dit_state_dict, det_uniform
    tcfg = dict(vocab=211, dim=64, dim_attn=64, dim_ffn=128, num_heads=1, num_layers=2, num_buckets=32)
    tsd = deterministic_t5_state_dict(**tcfg)
    t5 = WanT5EncoderModel(shared_pos=False, **tcfg)
    t5.load_state_dict(tsd, device=DEV)
    tok = _ToyTokenizer(211)
    dit = WanTransformer3DModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    dit.load_state_dict(deterministic_dit_state_dict(dim=256, ffn_dim=512, num_layers=2, text_dim=64), device=DEV)
    pipe = WanPipeline(tokenizer=tok, text_encoder=t5, transformer=dit, scheduler=FlowUniPCMultistepScheduler(shift=1))
    src = det_uniform("t5p.src", (1, 16, 3, 12, 20), 1.0).to(DEV)
    lat = torch.cat([src, det_uniform("t5p.noise", (1, 16, 4, 12, 20), 1.7).to(DEV)], dim=2)    # src | ground | tgt
    prompt, neg = "remove the red cup from the wooden table", "blurry, low quality"
    kw = dict(latents=lat, height=96, width=160, source_frames=9, reasoning_frames=4,
              num_inference_steps=3, guidance_scale=5.0, shift=5.0, repeat_rope=True, cot=True,
              weight_dtype=torch.float32, output_type="latent")
    got = pipe(prompt=prompt, negative_prompt=neg, **kw).latents
    enc = tok([prompt, neg], padding="max_length", max_length=512)
    ref_emb = T.T5EncoderOracle(tsd, 1, 2, 32).forward(enc.input_ids, enc.attention_mask)
    n = enc.attention_mask.sum(1).tolist()
    assert n == [9, 4]
    embeds = pipe.encode_prompt(prompt, neg, True, device=torch.device(DEV))
    assert [tuple(e.shape) for e in embeds[0]] == [(9, 64)] and [tuple(e.shape) for e in embeds[1]] == [(4, 64)]
    assert rel_l2(embeds[0][0], ref_emb[0, :9]) < 1.5e-2 and rel_l2(embeds[1][0], ref_emb[1, :4]) < 1.5e-2
    want = pipe(prompt_embeds=[ref_emb[0, :9].to(DEV)], negative_prompt_embeds=[ref_emb[1, :4].to(DEV)], **kw).latents
    assert rel_l2(got, want) < 2e-2


def test_from_pretrained_files_vae_and_t5(tmp_path):
    """The reference's checkpoint conventions: Wan2.1_VAE.pth holds un-prefixed keys (wan_vae.py:699-702 adds
    "model."), models_t5_umt5-xxl-enc-bf16.pth is a plain state dict (wan_text_encoder.py:383-390); both loaders also
    take .safetensors."""
    from safetensors.torch import save_file
    from videocof_amd import AutoencoderKLWan
    from videocof_amd.weights import deterministic_vae_state_dict
    vsd = deterministic_vae_state_dict()
    bare = {k[len("model."):]: v.contiguous() for k, v in vsd.items()}
    save_file(bare, str(tmp_path / "vae.safetensors"))
    torch.save(bare, str(tmp_path / "Wan2.1_VAE.pth"))
    direct = AutoencoderKLWan()
    direct.load_state_dict(vsd, device=DEV)
    video = (torch.rand(1, 3, 5, 32, 48, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(DEV)
    want = direct.encode(video)[0].mode()
    for name in ("vae.safetensors", "Wan2.1_VAE.pth"):
        vae = AutoencoderKLWan.from_pretrained(str(tmp_path / name))
        assert torch.equal(vae.encode(video)[0].mode(), want), name

    tsd = deterministic_t5_state_dict(**TINY)
    save_file({k: v.contiguous() for k, v in tsd.items()}, str(tmp_path / "t5.safetensors"))
    torch.save(tsd, str(tmp_path / "t5.pth"))
    ids = torch.randint(1, TINY["vocab"], (1, 24), generator=torch.Generator().manual_seed(1)).to(DEV)
    ref = WanT5EncoderModel(shared_pos=False, **TINY)
    ref.load_state_dict(tsd, device=DEV)
    want = ref(ids)[0]
    kwargs = dict(TINY, shared_pos=False, dropout=0.0, text_length=512, tokenizer_subpath="ignored")
    for name in ("t5.safetensors", "t5.pth"):
        m = WanT5EncoderModel.from_pretrained(str(tmp_path / name), additional_kwargs=kwargs)
        assert torch.equal(m(ids)[0], want), name
    with pytest.raises(FileNotFoundError):
        WanT5EncoderModel.from_pretrained(str(tmp_path / "missing.pth"), additional_kwargs=kwargs)
