"""CPU restatement of the reference's umT5 encoder (TEST INFRASTRUCTURE ONLY).

Follows ``videox_fun/models/wan_text_encoder.py`` of the reference, fp32 on the CPU:

* ``relative_position_bucket``  -- ``T5RelativeEmbedding._relative_position_bucket`` (:245-264), bidirectional
* ``t5_layer_norm``             -- ``T5LayerNorm.forward`` (:48-60)
* ``t5_attention``              -- ``T5Attention.forward`` (:75-113): no 1/sqrt(d) scaling, additive position bias,
                                   masked keys get ``finfo.min`` in place of the bias, fp32 softmax
* ``t5_feed_forward``           -- ``T5FeedForward.forward`` (:125-130) with the tanh GELU of :38-41
* ``T5EncoderOracle.forward``   -- ``WanT5EncoderModel.forward`` (:281-296) with per-block position
                                   embeddings (``shared_pos=False``, ``T5SelfAttention.forward`` :159-164)

Pinned by ``tests/golden/t5_*.npz`` which ``oracle/gen_golden_t5.py`` captured from the reference itself.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def relative_position_bucket(rel_pos: torch.Tensor, num_buckets: int, max_dist: int = 128) -> torch.Tensor:
    """rel_pos = key index - query index (int64) -> bucket id, bidirectional variant."""
    nb = num_buckets // 2
    buckets = (rel_pos > 0).long() * nb
    n = rel_pos.abs()
    max_exact = nb // 2
    # log of 0 is -inf -> .long() of -inf is INT64_MIN; torch.where picks the exact branch there
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return buckets + torch.where(n < max_exact, n, large)


def bucket_lut(L: int, num_buckets: int) -> torch.Tensor:
    """Bucket of every offset j - i in [-(L-1), L-1], index j - i + L - 1 (what wan_t5_softmax_bias takes)."""
    return relative_position_bucket(torch.arange(-(L - 1), L), num_buckets).to(torch.int32)


def t5_layer_norm(x: torch.Tensor, w: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    return w * (x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps))


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def t5_attention(x, wq, wk, wv, wo, num_heads, mask, pos_bias):
    """x [B, L, C]; mask [B, L] (1 = keep); pos_bias [1, N, L, L]."""
    B, L, _ = x.shape
    q = (x @ wq.t()).view(B, L, num_heads, -1)
    k = (x @ wk.t()).view(B, L, num_heads, -1)
    v = (x @ wv.t()).view(B, L, num_heads, -1)
    bias = pos_bias.expand(B, -1, -1, -1).clone()
    if mask is not None:
        bias.masked_fill_(mask.view(B, 1, 1, L) == 0, torch.finfo(x.dtype).min)
    attn = torch.einsum("binc,bjnc->bnij", q, k) + bias
    attn = torch.softmax(attn.float(), dim=-1)
    out = torch.einsum("bnij,bjnc->binc", attn, v).reshape(B, L, -1)
    return out @ wo.t()


def t5_feed_forward(x, w_gate, w_fc1, w_fc2):
    return ((x @ w_fc1.t()) * gelu_tanh(x @ w_gate.t())) @ w_fc2.t()


class T5EncoderOracle:
    def __init__(self, sd: Dict[str, torch.Tensor], num_heads: int, num_layers: int, num_buckets: int):
        self.sd = {k: v.float() for k, v in sd.items()}
        self.num_heads, self.num_layers, self.num_buckets = num_heads, num_layers, num_buckets

    def pos_bias(self, i: int, L: int) -> torch.Tensor:
        rel = torch.arange(L)[None, :] - torch.arange(L)[:, None]
        emb = self.sd[f"blocks.{i}.pos_embedding.embedding.weight"][relative_position_bucket(rel, self.num_buckets)]
        return emb.permute(2, 0, 1)[None].contiguous()            # [1, N, L, L]

    def block(self, i: int, x: torch.Tensor, mask) -> torch.Tensor:
        p, sd = f"blocks.{i}.", self.sd
        h = t5_layer_norm(x, sd[p + "norm1.weight"])
        x = x + t5_attention(h, sd[p + "attn.q.weight"], sd[p + "attn.k.weight"], sd[p + "attn.v.weight"],
                             sd[p + "attn.o.weight"], self.num_heads, mask, self.pos_bias(i, x.shape[1]))
        h = t5_layer_norm(x, sd[p + "norm2.weight"])
        return x + t5_feed_forward(h, sd[p + "ffn.gate.0.weight"], sd[p + "ffn.fc1.weight"], sd[p + "ffn.fc2.weight"])

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask=None) -> torch.Tensor:
        x = self.sd["token_embedding.weight"][input_ids]
        for i in range(self.num_layers):
            x = self.block(i, x, attention_mask)
        return t5_layer_norm(x, self.sd["norm.weight"])
