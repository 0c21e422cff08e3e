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
    attended; query rows beyond ``q_lens`` come back as zeros.  A ragged ``k_lens`` runs as ONE launch
    (``wan_attention_fwd_varlen``: the kernel reads the per-sample key counts from device memory); lengths given as a device
    tensor are never read by the host.
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
    klen, kl_dev = None, None
    if k_lens is not None:
        if torch.is_tensor(k_lens) and k_lens.is_cuda:
            # Lengths that live on the device are never read by the host (no sync), so they cannot be validated here the way a
            # host list is (ValueError below): the kernel CLAMPS -- a length above Lk attends Lk keys, a length <= 0 gives a zero
            # row block -- and lengths are taken as int32 (Lk < 2^31 always).  What can be checked without a sync is checked:
            if k_lens.device != q.device:
                raise ValueError(f"k_lens lives on {k_lens.device}, q on {q.device}")
            if k_lens.dtype not in (torch.int32, torch.int64) or k_lens.dim() != 1:
                raise ValueError(f"k_lens must be a 1-D int32 / int64 tensor, got {k_lens.dtype} with shape {tuple(k_lens.shape)}")
            kl_dev = k_lens.clamp(min=-1, max=Lk).to(torch.int32).contiguous()      # clamp BEFORE the narrowing cast: no int64 wrap-around
        else:
            kl = [int(x) for x in (k_lens.tolist() if torch.is_tensor(k_lens) else k_lens)]
            if len(kl) != B or min(kl) < 0 or max(kl) > Lk:
                raise ValueError(f"k_lens={kl} for a batch of {B} with {Lk} keys")
            if len(set(kl)) == 1 and kl[0] > 0:
                klen = kl[0]
            else:
                kl_dev = torch.tensor(kl, dtype=torch.int32).to(q.device, non_blocking=True)
    if kl_dev is not None:
        # ragged batch: ONE launch, every workgroup reads its sample's key count (the reference packs the samples behind
        # cu_seqlens, :95-146).  V rows past a sample's length are zeroed so that the V^T pad columns the last tile touches are finite
        if kl_dev.shape != (B,):
            raise ValueError(f"k_lens must hold {B} lengths")
        keep = torch.arange(Lk, device=q.device)[None, :] < kl_dev[:, None]
        vb = torch.where(keep[..., None], vb, torch.zeros((), device=q.device, dtype=vb.dtype))
        ldvt = ops.round_up(Lk, 64)
        vt = torch.empty(B, N * D, ldvt, device=q.device, dtype=torch.bfloat16)
        for b in range(B):
            ops.transpose_pad(vb[b], ldvt, out=vt[b])
        out = ops.attention_fwd(qb, kb, vt, N, softmax_scale=softmax_scale, k_lens=kl_dev)
        out = out.view(B, Lq, N, D)
        out.masked_fill_((kl_dev <= 0)[:, None, None, None], 0)      # flash-attn: a sample without keys gives zero rows
    else:
        ldvt = ops.round_up(Lk if klen is None else klen, 64)
        vt = torch.empty(B, N * D, ldvt, device=q.device, dtype=torch.bfloat16)
        for b in range(B):
            ops.transpose_pad(vb[b][: (klen or Lk)], ldvt, out=vt[b])
        out = ops.attention_fwd(qb, kb, vt, N, k_len=klen, softmax_scale=softmax_scale)
        out = out.view(B, Lq, N, D)
    if q_lens is not None:
        ql = q_lens.to(device=q.device, dtype=torch.int64) if torch.is_tensor(q_lens) else \
            torch.tensor([int(x) for x in q_lens], dtype=torch.int64).to(q.device, non_blocking=True)
        out.masked_fill_((torch.arange(Lq, device=q.device)[None, :] >= ql[:, None])[..., None, None], 0)
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
