"""umT5 text encoder on the HIP kernels.

Host-side mirror of ``videox_fun/models/wan_text_encoder.py``: same class name, constructor arguments,
``state_dict`` keys, ``from_pretrained`` and ``forward(input_ids, attention_mask) -> (hidden,)`` contract as
``WanT5EncoderModel`` (:264-296, 298-394), encoder only, ``shared_pos=False`` (per-block relative position
tables, the umT5 configuration of ``config/wan2.1/wan_civitai.yaml:14-26``).

The whole padded sequence is processed exactly as the reference does (queries at padded positions
included -- their rows are what the reference returns there as well; ``WanPipeline._get_t5_prompt_embeds``
trims them, pipeline_wan.py:173-181).  Per block (``T5SelfAttention.forward`` :159-164):

    h   = wan_rmsnorm_rows(x)                          T5LayerNorm, fp32 stream -> bf16
    q|k = wan_gemm_bf16(h, [Wq;Wk])                    one GEMM, N = 2*dim_attn
    v^T = wan_gemm_bf16(h, Wv, WAN_EPI_BF16_T)         per sample, [dim_attn, L]
    S   = wan_gemm_bf16_batched(q_h, k_h)              one launch for all heads, fp32 [H, L, L]
    P   = wan_t5_softmax_bias(S, table, lut, k_len)    + relative position bias, key mask, bf16
    o   = wan_gemm_bf16_batched(P_h, v^T_h)            [L, dim_attn]
    x  += wan_gemm_bf16(o, Wo, WAN_EPI_RESID_F32)
    h   = wan_rmsnorm_rows(x)
    u   = wan_mul_bf16(gemm(h, Wfc1), gemm(h, Wgate, WAN_EPI_GELU_BF16))
    x  += wan_gemm_bf16(u, Wfc2, WAN_EPI_RESID_F32)

The residual stream is fp32 (the reference keeps it in the module dtype, bf16); weights are bf16, every
product accumulates in fp32.  There is no CPU / eager path: CPU tensors raise.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch

from . import ops

__all__ = ["WanT5EncoderModel", "relative_position_buckets"]


def relative_position_buckets(L: int, num_buckets: int, max_dist: int = 128) -> torch.Tensor:
    """Bucket id of every key-minus-query offset in [-(L-1), L-1] (index offset + L - 1), int32.
    Bidirectional ``T5RelativeEmbedding._relative_position_bucket`` (wan_text_encoder.py:245-264):
    half of the buckets per sign; exact up to nb/4, then logarithmic up to max_dist, then clipped."""
    nb = num_buckets // 2
    max_exact = nb // 2
    lut = torch.empty(2 * L - 1, dtype=torch.int32)
    for idx, rel in enumerate(range(-(L - 1), L)):
        n = abs(rel)
        if n < max_exact:
            b = n
        else:
            # float32 arithmetic in the reference's operation order, truncation toward zero
            r = torch.log(torch.tensor(n, dtype=torch.float32) / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)
            b = min(max_exact + int(r.long()), nb - 1)
        lut[idx] = b + (nb if rel > 0 else 0)
    return lut


class _Block:
    __slots__ = ("n1", "n2", "w_qk", "w_v", "w_o", "w_gate", "w_fc1", "w_fc2", "table")


class WanT5EncoderModel:
    def __init__(self, vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets, shared_pos=True,
                 dropout=0.1, text_length: int = 512):
        if not isinstance(vocab, int):
            raise TypeError("vocab must be the vocabulary size (a shared nn.Embedding is not supported)")
        if shared_pos:
            raise NotImplementedError("shared_pos=True (T5 v1.0 style single position table) is not built; "
                                      "umT5 / Wan2.1 uses shared_pos=False (wan_civitai.yaml:25)")
        if dim_attn % num_heads or dim_attn // num_heads != 64:
            raise NotImplementedError(f"head_dim={dim_attn / num_heads:g}: only 64 (umT5) is built")
        if dim % 64 or dim_ffn % 64 or dim_attn % 64:
            raise ValueError("dim, dim_attn and dim_ffn must be multiples of 64")
        self.vocab, self.dim, self.dim_attn, self.dim_ffn = vocab, dim, dim_attn, dim_ffn
        self.num_heads, self.num_layers, self.num_buckets = num_heads, num_layers, num_buckets
        self.shared_pos, self.eps = False, 1e-6
        self.config = dict(vocab=vocab, dim=dim, dim_attn=dim_attn, dim_ffn=dim_ffn, num_heads=num_heads,
                           num_layers=num_layers, num_buckets=num_buckets, shared_pos=False, dropout=dropout)
        self.dtype = torch.bfloat16
        self._device: Optional[torch.device] = None
        self._emb: Optional[torch.Tensor] = None
        self._norm: Optional[torch.Tensor] = None
        self._blocks: List[_Block] = []
        self._lut: Dict[int, torch.Tensor] = {}

    # ------------------------------------------------------------------ parameters
    @property
    def device(self):
        return self._device

    def eval(self):
        return self

    def to(self, *args, **kwargs):
        if self._emb is None:
            return self
        raise NotImplementedError("weights are packed on the device given to load_state_dict")

    def state_dict_keys(self) -> List[str]:
        from .weights import t5_param_shapes
        cfg = {k: v for k, v in self.config.items() if k not in ("shared_pos", "dropout")}
        return list(t5_param_shapes(**cfg).keys())

    @torch.no_grad()
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True, device="cuda"):
        """Reference keys (``token_embedding.weight``, ``blocks.i.attn.q.weight`` ...).  q and k are packed
        into one [2*dim_attn, dim] bf16 operand; norms and position tables are kept in fp32."""
        dev = torch.device(device)
        want = set(self.state_dict_keys())
        missing = sorted(want - set(sd))
        unexpected = sorted(set(sd) - want)
        if missing or (strict and unexpected):
            raise KeyError(f"WanT5EncoderModel.load_state_dict: missing={missing[:4]}{'...' if len(missing) > 4 else ''} "
                           f"unexpected={unexpected[:4]}")

        def bf(name):
            return sd[name].to(device=dev, dtype=torch.bfloat16).contiguous()

        def f32(name):
            return sd[name].to(device=dev, dtype=torch.float32).contiguous()

        if tuple(sd["token_embedding.weight"].shape) != (self.vocab, self.dim):
            raise ValueError("token_embedding.weight has the wrong shape")
        self._emb, self._norm = bf("token_embedding.weight"), f32("norm.weight")
        self._blocks = []
        for i in range(self.num_layers):
            p, b = f"blocks.{i}.", _Block()
            b.n1, b.n2 = f32(p + "norm1.weight"), f32(p + "norm2.weight")
            b.w_qk = torch.cat([bf(p + "attn.q.weight"), bf(p + "attn.k.weight")]).contiguous()
            b.w_v, b.w_o = bf(p + "attn.v.weight"), bf(p + "attn.o.weight")
            b.w_gate, b.w_fc1, b.w_fc2 = bf(p + "ffn.gate.0.weight"), bf(p + "ffn.fc1.weight"), bf(p + "ffn.fc2.weight")
            b.table = f32(p + "pos_embedding.embedding.weight")
            if tuple(b.table.shape) != (self.num_buckets, self.num_heads):
                raise ValueError(p + "pos_embedding.embedding.weight has the wrong shape")
            self._blocks.append(b)
        self._device = dev
        from .wan_transformer3d import IncompatibleKeys
        return IncompatibleKeys([], unexpected)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, additional_kwargs={}, low_cpu_mem_usage=False,
                        torch_dtype=torch.bfloat16, device="cuda"):
        """``models_t5_umt5-xxl-enc-bf16.pth`` / ``.safetensors`` loader (wan_text_encoder.py:298-394)."""
        import inspect
        ok = set(inspect.signature(cls.__init__).parameters) - {"self"}
        model = cls(**{k: v for k, v in dict(additional_kwargs).items() if k in ok})
        if not os.path.isfile(pretrained_model_path):
            raise FileNotFoundError(pretrained_model_path)
        if pretrained_model_path.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(pretrained_model_path)
        else:
            sd = torch.load(pretrained_model_path, map_location="cpu", weights_only=True)
        model.load_state_dict(sd, strict=False, device=device)
        return model

    # ------------------------------------------------------------------ forward
    def _bucket_lut(self, L: int) -> torch.Tensor:
        if L not in self._lut:
            self._lut[L] = relative_position_buckets(L, self.num_buckets).to(self._device)
        return self._lut[L]

    @torch.no_grad()
    def forward(self, input_ids: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None):
        """input_ids int64 [B, L]; attention_mask [B, L] (1 = token, 0 = padding, a prefix of ones as the
        tokenizer produces it) or None.  Returns ``(hidden,)`` with hidden bf16 [B, L, dim]."""
        if self._emb is None:
            raise RuntimeError("WanT5EncoderModel: load_state_dict first")
        if input_ids is None or input_ids.dim() != 2:
            raise ValueError("input_ids must be [B, L]")
        if not input_ids.is_cuda:
            raise RuntimeError("WanT5EncoderModel runs on the HIP kernels only (no CPU fallback): move input_ids to the GPU")
        B, L = input_ids.shape
        if L > 2048:
            raise NotImplementedError("sequences longer than 2048 tokens are not built")
        ids = input_ids.to(torch.int64).contiguous()
        lo, hi = int(ids.min()), int(ids.max())
        if lo < 0 or hi >= self.vocab:
            raise IndexError(f"input_ids outside [0, {self.vocab}): min={lo} max={hi}")
        if attention_mask is None:
            k_lens = [L] * B
        else:
            m = attention_mask.to(device=ids.device).reshape(B, L) != 0
            k_lens = [int(v) for v in m.sum(dim=1).tolist()]
            prefix = torch.arange(L, device=ids.device)[None, :] < torch.tensor(k_lens, device=ids.device)[:, None]
            if not torch.equal(m, prefix):
                raise NotImplementedError("attention_mask must be a prefix of ones per sample (right padding)")
            if min(k_lens) < 1:
                raise ValueError("attention_mask has a sample without any token")
        H, D, C, A, Fd = self.num_heads, 64, self.dim, self.dim_attn, self.dim_ffn
        L_in = L
        if L % 4:       # the score product needs L % 4 == 0: append masked positions, trimmed from the result
            L = ops.round_up(L, 4)
            ids = torch.cat([ids, ids.new_zeros(B, L - L_in)], dim=1).contiguous()
        Lp = ops.round_up(L, 64)
        M = B * L
        dev = ids.device
        lut = self._bucket_lut(L)

        x = ops.embedding_rows(ids.view(-1), self._emb)                        # fp32 [M, C]
        h = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
        qk = torch.empty(M, 2 * A, device=dev, dtype=torch.bfloat16)
        vt = torch.zeros(A, Lp, device=dev, dtype=torch.bfloat16)               # pad columns stay zero
        S = torch.empty(H, L, L, device=dev, dtype=torch.float32)
        P = torch.empty(H, L, Lp, device=dev, dtype=torch.bfloat16)
        o = torch.empty(M, A, device=dev, dtype=torch.bfloat16)
        g = torch.empty(M, Fd, device=dev, dtype=torch.bfloat16)
        f = torch.empty(M, Fd, device=dev, dtype=torch.bfloat16)
        for blk in self._blocks:
            ops.rmsnorm_rows(x, blk.n1, self.eps, out=h)
            ops.gemm(h, blk.w_qk, None, ops.EPI_BF16, out=qk)
            for b in range(B):
                r0 = b * L
                ops.gemm(h[r0:r0 + L], blk.w_v, None, ops.EPI_BF16_T, out=vt)
                qb = qk[r0:r0 + L]                                              # q | k rows of this sample
                # S[hd] = q_hd k_hd^T: heads are 64 columns apart in both operands
                ops.gemm_batched(qb[:, :D], D, qb[:, A:A + D], D, S[0], L * L, L, L, D, H, ops.EPI_F32)
                ops.t5_softmax_bias(S, blk.table, lut, H, k_lens[b], Lp, out=P)
                # o[:, hd] = P[hd] v_hd: W rows are the 64 d-rows of head hd in v^T
                ops.gemm_batched(P[0], L * Lp, vt[:D], D * Lp, o[r0:r0 + L, :D], D, L, D, Lp, H, ops.EPI_BF16)
            ops.gemm(o, blk.w_o, None, ops.EPI_RESID_F32, out=x)
            ops.rmsnorm_rows(x, blk.n2, self.eps, out=h)
            ops.gemm(h, blk.w_gate, None, ops.EPI_GELU_BF16, out=g)
            ops.gemm(h, blk.w_fc1, None, ops.EPI_BF16, out=f)
            ops.mul_bf16(f, g, out=g)
            ops.gemm(g, blk.w_fc2, None, ops.EPI_RESID_F32, out=x)
        out = ops.rmsnorm_rows(x, self._norm, self.eps, out_dtype=torch.bfloat16)
        return (out.view(B, L, C)[:, :L_in],)

    __call__ = forward
