"""Drop-in for the reference's attention backend dispatcher.

Mirrors ``attention()`` / ``flash_attention()`` of
``videox_fun/models/attention_utils.py:43-211`` (same argument names, same
``[B, L, N, D]`` layout, same return) with the gfx950 flash kernel as the only
backend.  The reference selects FA3/FA2/Sage/SDPA through the env var
``VIDEOX_ATTENTION_TYPE`` (:169); here every value maps to the HIP kernel and
options the kernel does not implement raise, as flash-attn itself would.
"""
from __future__ import annotations

import warnings

import torch

from . import ops

__all__ = ["attention", "flash_attention"]


def flash_attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None,
                    causal=False, window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16,
                    version=None):
    """q [B,Lq,N,128], k/v [B,Lk,N,128] -> [B,Lq,N,128] in q.dtype.

    Semantics of the flash-attn branch (attention_utils.py:85-149): inputs that
    are not half precision are cast to ``dtype``; keys beyond ``k_lens`` are not
    attended; query rows beyond ``q_lens`` come back as zeros.
    """
    if dtype not in (torch.bfloat16,):
        raise NotImplementedError("the gfx950 attention kernel computes in bfloat16 only")
    if causal or tuple(window_size) != (-1, -1) or dropout_p != 0.:
        raise NotImplementedError("causal / windowed / dropout attention is not on the Wan2.1 path "
                                  "(wan_transformer3d.py:294-299 passes none of them)")
    if q.size(-1) != 128:
        raise NotImplementedError(f"head_dim {q.size(-1)} (Wan2.1 uses 128)")
    B, Lq, N, D = q.shape
    Lk = k.shape[1]
    out_dtype = q.dtype
    if q_scale is not None:
        q = q * q_scale
    qb = q.to(torch.bfloat16).reshape(B, Lq, N * D)
    kb = k.to(torch.bfloat16).reshape(B, Lk, N * D)
    vb = v.to(torch.bfloat16).reshape(B, Lk, N * D)
    if not qb.is_contiguous():
        qb = qb.contiguous()
    if not kb.is_contiguous():
        kb = kb.contiguous()
    klen = None
    if k_lens is not None:
        kl = [int(x) for x in (k_lens.tolist() if torch.is_tensor(k_lens) else k_lens)]
        if len(set(kl)) != 1:
            # ragged batches: run sample by sample (the reference builds cu_seqlens, :95-100)
            outs = [flash_attention(q[i:i + 1], k[i:i + 1], v[i:i + 1],
                                    None if q_lens is None else q_lens[i:i + 1], [kl[i]],
                                    softmax_scale=softmax_scale, dtype=dtype) for i in range(B)]
            return torch.cat(outs)
        klen = kl[0]
    ldvt = ops.round_up(Lk if klen is None else klen, 64)
    vt = torch.empty(B, N * D, ldvt, device=q.device, dtype=torch.bfloat16)
    for b in range(B):
        ops.transpose_pad(vb[b][: (klen or Lk)], ldvt, out=vt[b])
    out = ops.attention_fwd(qb, kb, vt, N, k_len=klen, softmax_scale=softmax_scale)
    out = out.view(B, Lq, N, D)
    if q_lens is not None:
        ql = [int(x) for x in (q_lens.tolist() if torch.is_tensor(q_lens) else q_lens)]
        for b in range(B):
            out[b, ql[b]:] = 0
    return out.type(out_dtype)


def attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None,
              causal=False, window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16,
              fa_version=None, attention_type=None, attn_mask=None):
    """Same signature as attention_utils.py:152-168.  ``attention_type`` / ``fa_version`` are
    accepted for compatibility; there is one backend."""
    if attn_mask is not None:
        raise NotImplementedError("attn_mask is only honoured by the reference's SDPA fallback "
                                  "(attention_utils.py:207-208); the Wan path never passes one")
    if attention_type not in (None, "FLASH_ATTENTION", "SAGE_ATTENTION", "SDPA"):
        warnings.warn(f"unknown attention_type {attention_type!r}; using the HIP flash kernel")
    return flash_attention(q=q, k=k, v=v, q_lens=q_lens, k_lens=k_lens, dropout_p=dropout_p,
                           softmax_scale=softmax_scale, q_scale=q_scale, causal=causal,
                           window_size=window_size, deterministic=deterministic, dtype=dtype,
                           version=fa_version)
