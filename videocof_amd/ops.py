"""Torch-tensor front end of the C ABI (``include/wan_hip.h``).

Every function takes CUDA(HIP) tensors, validates dtype/layout, and enqueues the
kernel on torch's current stream.  Nothing here computes on the host and nothing
falls back to eager PyTorch: a CPU tensor raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes
import functools
import math
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import RopeParams

EPI_BF16, EPI_GELU_BF16, EPI_F32, EPI_RESID_F32, EPI_BF16_T = (
    _lib.EPI_BF16, _lib.EPI_GELU_BF16, _lib.EPI_F32, _lib.EPI_RESID_F32, _lib.EPI_BF16_T)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _on_tensor_device(fn):
    """Launch on the device that holds the operands: the kernels take raw pointers, so HIP's current device (and
    torch's current stream, which is per device) must be the tensors' device.  Entering ``torch.cuda.device`` only when
    it differs keeps the common single-device case free of extra calls."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        t = next((a for a in args if torch.is_tensor(a)), None)
        if t is None and args and isinstance(args[0], (list, tuple)):       # lincomb: list of (coefficient, tensor)
            t = next((x[1] for x in args[0] if isinstance(x, tuple) and torch.is_tensor(x[1])), None)
        if t is not None and t.is_cuda and t.device.index != torch.cuda.current_device():
            with torch.cuda.device(t.device):
                return fn(*args, **kwargs)
        return fn(*args, **kwargs)
    return wrapper


def set_tuning(key: str, value: int) -> None:
    """Developer switch of the library (``wan_set_tuning``, include/wan_hip.h)."""
    _lib.check(_lib.load().wan_set_tuning(key.encode(), int(value)), "wan_set_tuning")


def get_tuning(key: str) -> int:
    return int(_lib.load().wan_get_tuning(key.encode()))


def box_probe(device=None, target_ms: int = 300) -> dict:
    """``wan_box_probe`` (include/wan_hip.h): the fixed calibration workload of the library -- what the matrix pipes of this box hold
    at the power limit under a flash-attention instruction mix, and its plain copy rate.  SYNCHRONISES; a measurement for benchmarks
    (bench.py's `box` object), never part of the product path.  The 512 MiB scratch is allocated here and freed on return."""
    import ctypes
    dev = torch.device(device if device is not None else torch.cuda.current_device())
    if dev.type != "cuda":
        raise RuntimeError("box_probe needs a GPU")
    lib = _lib.load()
    with torch.cuda.device(dev):
        n = int(lib.wan_box_probe_scratch_bytes())
        scratch = torch.empty(n, device=dev, dtype=torch.uint8)
        res = _lib.BoxProbeResult()
        _lib.check(lib.wan_box_probe(ctypes.byref(res), ctypes.c_void_p(scratch.data_ptr()), n, int(target_ms), _stream()), "wan_box_probe")
        del scratch
    return {"mfma_mix_tflops": round(float(res.mfma_mix_tflops), 1), "copy_tbps": round(float(res.copy_tbps), 3),
            "mfma_ms": round(float(res.mfma_ms), 2), "copy_ms": round(float(res.copy_ms), 2)}


def _need(t: torch.Tensor, dtype, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.dim() >= 1 and t.stride(-1) != 1:
        raise ValueError(f"{name}: last dimension must be contiguous")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ---------------------------------------------------------------------------------------------
@_on_tensor_device
def ln_modulate(x: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor],
                add_one: bool, rows_per_batch: int, eps: float,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x fp32 [rows, dim]; scale/shift fp32 [nbatch, dim] or None -> bf16 [rows, dim]."""
    _need(x, torch.float32, "ln_modulate.x")
    if not x.is_contiguous() or x.dim() != 2:
        raise ValueError("ln_modulate.x must be a contiguous [rows, dim] tensor")
    rows, dim = x.shape
    for nm, t in (("scale", scale), ("shift", shift)):
        if t is not None:
            _need(t, torch.float32, "ln_modulate." + nm)
            if not t.is_contiguous() or t.shape[-1] != dim or t.numel() * rows_per_batch < rows * dim:
                raise ValueError(f"ln_modulate.{nm}: shape {tuple(t.shape)} does not cover {rows} rows of {dim}")
    if out is None:
        out = torch.empty(rows, dim, device=x.device, dtype=torch.bfloat16)
    _need(out, torch.bfloat16, "ln_modulate.out")
    lib = _lib.load()
    _lib.check(lib.wan_ln_modulate(_p(x), _p(scale), _p(shift), int(bool(add_one)), _p(out), rows, dim,
                                   int(rows_per_batch), float(eps), _stream()), "wan_ln_modulate")
    return out


@_on_tensor_device
def rmsnorm_rope_(x0: torch.Tensor, w0: torch.Tensor, x1: Optional[torch.Tensor], w1: Optional[torch.Tensor],
                  head_dim: int, eps: float, rope: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                  rope_params: Optional[RopeParams] = None, x0_scale: float = 1.0) -> None:
    """In-place RMSNorm(+RoPE) of one or two bf16 [rows, dim] views sharing a row stride.
    ``x0_scale`` multiplies the x0 result before its bf16 rounding (``q_prescale(softmax_scale)`` to
    feed ``attention_fwd(..., q_prescaled=True)``)."""
    _need(x0, torch.bfloat16, "rmsnorm_rope.x0")
    _need(w0, torch.float32, "rmsnorm_rope.w0")
    rows, dim = x0.shape
    ld = x0.stride(0)
    if x1 is not None:
        _need(x1, torch.bfloat16, "rmsnorm_rope.x1")
        _need(w1, torch.float32, "rmsnorm_rope.w1")
        if x1.shape != x0.shape or x1.stride(0) != ld:
            raise ValueError("rmsnorm_rope: x0 and x1 must have the same shape and row stride")
    cos = sin = None
    if rope is not None:
        cos, sin = rope
        _need(cos, torch.float32, "rmsnorm_rope.cos")
        _need(sin, torch.float32, "rmsnorm_rope.sin")
        if cos.shape != sin.shape or cos.shape[1] != head_dim // 2 or not (cos.is_contiguous() and sin.is_contiguous()):
            raise ValueError("rmsnorm_rope: cos/sin must be contiguous [max_pos, head_dim/2]")
        if rope_params is None:
            raise ValueError("rmsnorm_rope: rope tables given without rope_params")
    lib = _lib.load()
    _lib.check(lib.wan_rmsnorm_rope(_p(x0), _p(w0), _p(x1), _p(w1), ld, rows, dim, head_dim, float(eps),
                                    _p(cos), _p(sin),
                                    ctypes.byref(rope_params) if rope_params is not None else None,
                                    float(x0_scale), _stream()), "wan_rmsnorm_rope")


@_on_tensor_device
def rmsnorm_rope_sp(x0: torch.Tensor, w0: torch.Tensor, x1: Optional[torch.Tensor], w1: Optional[torch.Tensor], head_dim: int,
                    eps: float, rope: Tuple[torch.Tensor, torch.Tensor], rope_params: RopeParams, wire0: torch.Tensor,
                    wire1: Optional[torch.Tensor], slabs: int, batch: int, x0_scale: float = 1.0, split: int = 0) -> None:
    """``rmsnorm_rope_`` that leaves x0 / x1 untouched and writes the results into Ulysses token-major wire buffers
    ``[slabs][rows_per_batch][batch][dim / slabs]`` (include/wan_hip.h, a21).  ``split`` > 0: two head groups, channels [0, split) and
    [split, dim / slabs) of every slab, each a complete wire buffer, one behind the other (``wan_rmsnorm_rope_sp_split``)."""
    _need(x0, torch.bfloat16, "rmsnorm_rope_sp.x0")
    _need(w0, torch.float32, "rmsnorm_rope_sp.w0")
    rows, dim = x0.shape
    ld = x0.stride(0)
    for nm, t in (("wire0", wire0), ("wire1", wire1)):
        if t is not None:
            _need(t, torch.bfloat16, "rmsnorm_rope_sp." + nm)
            if not t.is_contiguous() or t.numel() < rows * dim:
                raise ValueError(f"rmsnorm_rope_sp.{nm} must be contiguous with >= rows * dim elements")
    if x1 is not None:
        _need(x1, torch.bfloat16, "rmsnorm_rope_sp.x1")
        _need(w1, torch.float32, "rmsnorm_rope_sp.w1")
        if x1.shape != x0.shape or x1.stride(0) != ld or wire1 is None:
            raise ValueError("rmsnorm_rope_sp: x0 / x1 must share shape and row stride, and x1 needs wire1")
    cos, sin = rope
    lib = _lib.load()
    if split:
        _lib.check(lib.wan_rmsnorm_rope_sp_split(_p(x0), _p(w0), _p(x1), _p(w1), ld, rows, dim, head_dim, float(eps), _p(cos), _p(sin),
                                                 ctypes.byref(rope_params), float(x0_scale), _p(wire0), _p(wire1), int(slabs), int(batch),
                                                 int(split), _stream()), "wan_rmsnorm_rope_sp_split")
        return
    _lib.check(lib.wan_rmsnorm_rope_sp(_p(x0), _p(w0), _p(x1), _p(w1), ld, rows, dim, head_dim, float(eps), _p(cos), _p(sin),
                                       ctypes.byref(rope_params), float(x0_scale), _p(wire0), _p(wire1), int(slabs), int(batch),
                                       _stream()), "wan_rmsnorm_rope_sp")


@_on_tensor_device
def sp_pack_heads(x: torch.Tensor, wire: torch.Tensor, P: int, T: int, B: int, split: int = 0) -> torch.Tensor:
    """x bf16 [B*T, ldx >= C] (C = P * Cl) -> token-major wire [P][T][B][Cl] (``split`` > 0: two head-group wires, see ``rmsnorm_rope_sp``)."""
    _need(x, torch.bfloat16, "sp_pack_heads.x")
    _need(wire, torch.bfloat16, "sp_pack_heads.wire")
    C = x.shape[1]
    if x.shape[0] != B * T or C % P or not wire.is_contiguous() or wire.numel() < B * T * C:
        raise ValueError("sp_pack_heads: shapes disagree")
    lib = _lib.load()
    if split:
        _lib.check(lib.wan_sp_pack_heads_split(_p(x), x.stride(0), _p(wire), P, T, B, C // P, int(split), _stream()), "wan_sp_pack_heads_split")
    else:
        _lib.check(lib.wan_sp_pack_heads(_p(x), x.stride(0), _p(wire), P, T, B, C // P, _stream()), "wan_sp_pack_heads")
    return wire


@_on_tensor_device
def sp_unpack_heads(wire: torch.Tensor, x: torch.Tensor, P: int, T: int, B: int, split: int = 0) -> torch.Tensor:
    """token-major wire [P][T][B][Cl] -> x bf16 [B*T, ldx >= P*Cl] (column s * Cl + c); ``split`` > 0: from two head-group wires."""
    _need(x, torch.bfloat16, "sp_unpack_heads.x")
    _need(wire, torch.bfloat16, "sp_unpack_heads.wire")
    C = x.shape[1]
    if x.shape[0] != B * T or C % P or not wire.is_contiguous() or wire.numel() < B * T * C:
        raise ValueError("sp_unpack_heads: shapes disagree")
    lib = _lib.load()
    if split:
        _lib.check(lib.wan_sp_unpack_heads_split(_p(wire), _p(x), x.stride(0), P, T, B, C // P, int(split), _stream()),
                   "wan_sp_unpack_heads_split")
    else:
        _lib.check(lib.wan_sp_unpack_heads(_p(wire), _p(x), x.stride(0), P, T, B, C // P, _stream()), "wan_sp_unpack_heads")
    return x


@_on_tensor_device
def sp_unpack_vt(wire: torch.Tensor, vt: torch.Tensor, P: int, T: int) -> torch.Tensor:
    """arrived channel-major wire [P][Cl][B][T] -> vt bf16 [B, Cl, ldvt >= P*T], column s * T + t."""
    _need(vt, torch.bfloat16, "sp_unpack_vt.vt")
    _need(wire, torch.bfloat16, "sp_unpack_vt.wire")
    B, Cl, ldvt = vt.shape
    if not wire.is_contiguous() or wire.numel() < P * Cl * B * T or not vt.is_contiguous():
        raise ValueError("sp_unpack_vt: shapes disagree")
    lib = _lib.load()
    _lib.check(lib.wan_sp_unpack_vt(_p(wire), _p(vt), ldvt, P, B, Cl, T, _stream()), "wan_sp_unpack_vt")
    return vt


def q_prescale(head_dim: int, softmax_scale: Optional[float] = None) -> float:
    """The factor q must carry for ``attention_fwd(q_prescaled=True)``: softmax_scale * log2(e)."""
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(head_dim)
    return float(scale) * _lib.LOG2E


@_on_tensor_device
def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], epilogue: int,
         out: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None,
         rows_per_batch: int = 0, ldo_t: int = 0) -> torch.Tensor:
    """acc = a @ w.T (+bias) with one of the fused epilogues.  a bf16 [M,K] (row stride free),
    w bf16 [N,K] (nn.Linear layout).  For EPI_RESID_F32 `out` is updated in place.  For
    EPI_BF16_T the result is [N, ldo_t] with out[n, m] (ldo_t >= M)."""
    _need(a, torch.bfloat16, "gemm.a")
    _need(w, torch.bfloat16, "gemm.w")
    M, K = a.shape
    N, Kw = w.shape
    if K != Kw:
        raise ValueError(f"gemm: a is [{M},{K}] but w is [{N},{Kw}]")
    if bias is not None:
        _need(bias, torch.float32, "gemm.bias")
        if bias.numel() != N:
            raise ValueError("gemm: bias length != N")
    dev = a.device
    if epilogue in (EPI_BF16, EPI_GELU_BF16):
        if out is None:
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        _need(out, torch.bfloat16, "gemm.out")
    elif epilogue == EPI_F32:
        if out is None:
            out = torch.empty(M, N, device=dev, dtype=torch.float32)
        _need(out, torch.float32, "gemm.out")
    elif epilogue == EPI_RESID_F32:
        if out is None:
            raise ValueError("gemm: EPI_RESID_F32 needs the residual stream as `out`")
        _need(out, torch.float32, "gemm.out")
        if gate is not None:
            _need(gate, torch.float32, "gemm.gate")
            if not gate.is_contiguous() or gate.shape[-1] != N:
                raise ValueError("gemm.gate must be contiguous [nbatch, N]")
    elif epilogue == EPI_BF16_T:
        if out is None:
            ldo_t = ldo_t or round_up(M, 64)
            out = torch.zeros(N, ldo_t, device=dev, dtype=torch.bfloat16)
        _need(out, torch.bfloat16, "gemm.out")
        if out.shape[0] != N or out.shape[1] < M:
            raise ValueError(f"gemm: transposed out must be [N={N}, >= M={M}], got {tuple(out.shape)}")
    else:
        raise ValueError(f"gemm: unknown epilogue {epilogue}")
    if epilogue != EPI_BF16_T and tuple(out.shape) != (M, N):
        raise ValueError(f"gemm: out shape {tuple(out.shape)} != ({M},{N})")
    lib = _lib.load()
    ws = gemm_workspace(dev, M, N, K)
    _lib.check(lib.wan_gemm_bf16_ws(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0),
                                    M, N, K, epilogue, _p(gate), int(rows_per_batch), _p(ws), ws.numel() if ws is not None else 0,
                                    _stream()), "wan_gemm_bf16_ws")
    return out


_GEMM_WS = {}


def gemm_workspace(dev: torch.device, M: int, N: int, K: int, fp8: bool = False) -> Optional[torch.Tensor]:
    """The caller-owned workspace of ``wan_gemm_bf16_ws`` (the persistent stream-K GEMM's partial tiles, arrival and
    ticket counters): ONE buffer per (device, STREAM), grown to the largest request and then kept for the life of the
    process.  The kernel keeps its counters and split-tile partial sums in it, so two launches that share a workspace must
    be stream-ordered: keyed by the current stream, GEMMs issued on different streams of one device (two pipelines, a
    caller's side stream) can never meet in one buffer.  Launches recorded under stream capture share one per-device
    "capture" workspace (the capturing stream is a pool stream that differs from capture to capture; a graph bakes the
    address in and the entry stays alive with this table); like the model's activation buffers it is owned by the graphs
    of that device, whose replays must not overlap each other; when a capture asks for more than the capture workspace holds, a new
    buffer is allocated and the old one stays alive for the launches already recorded.  The buffer is NOT cleared here -- the launch clears the
    4 KiB of counters it uses with a memset node of its own.  None when the shape does not use a workspace
    (``wan_gemm_workspace_bytes`` == 0).  ``fp8``: the request of the e4m3 Linear (``wan_gemm_fp8_workspace_bytes``; same buffer)."""
    need = int(_lib.load().wan_gemm_fp8_workspace_bytes(M, N, K) if fp8 else _lib.load().wan_gemm_workspace_bytes(M, N, K))
    if need <= 0:
        return None
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    capturing = torch.cuda.is_current_stream_capturing()
    key = (index, "capture" if capturing else int(torch.cuda.current_stream(index).cuda_stream))
    ws = _GEMM_WS.get(key)
    if ws is None or ws.numel() < need:
        if capturing:
            # a graph bakes addresses in: launches already recorded keep the buffer they were given (a workspace only has to live
            # through its own launch), so a grown request gets a NEW buffer and the old one is kept alive for the recorded nodes.
            # The first capture buffer of a device is sized like the largest eager workspace (the eager warm-up call of
            # GraphedForward / GraphedLoop has run every shape), so this branch is normally taken once.
            if ws is not None:
                _GEMM_WS_RETIRED.append(ws)
            need = max([need] + [t.numel() for (i, _), t in _GEMM_WS.items() if i == index])
        ws = torch.empty(need, device=torch.device("cuda", index), dtype=torch.uint8)
        _GEMM_WS[key] = ws
    return ws


_GEMM_WS_RETIRED = []


def release_gemm_workspaces(include_capture: bool = False) -> int:
    """Drop the cached persistent-GEMM workspaces (~128 MiB per (device, stream) that ever issued a Linear): the eager entries
    always -- each is re-allocated by the next GEMM of its stream, and a buffer goes back to the caching allocator on the stream
    whose launches used it, so nothing in flight loses it -- and, with ``include_capture``, the per-device capture workspaces and
    the retired ones as well -- ONLY once every hipGraph of the process that recorded launches on them is gone (a graph holds raw
    addresses, not references; ``WanTransformer3DModel.release_workspaces()`` therefore never passes True).  Returns the bytes released."""
    freed = 0
    for key in [k for k in _GEMM_WS if include_capture or k[1] != "capture"]:
        freed += _GEMM_WS.pop(key).numel()
    if include_capture:
        freed += sum(t.numel() for t in _GEMM_WS_RETIRED)
        _GEMM_WS_RETIRED.clear()
    return freed


class AttentionWorkspace:
    """Scratch of one ``wan_attention_fwd`` CALL SITE (the flags of the max-free attempt, its sticky "fast path off"
    word, the partial results of the split tail round).  One object per call site and stream: the sticky word then
    describes the inputs of THAT site only (a self-attention layer with out-of-window scores does not switch the
    kernel of cross-attention or of another model), and two streams never share flags.  Launches that use one
    workspace must be stream-ordered."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None

    def get(self, device: torch.device, nbytes: int) -> torch.Tensor:
        if self.buf is None or self.buf.device != device or self.buf.numel() < nbytes:
            self.buf = torch.zeros(nbytes, device=device, dtype=torch.uint8)       # zeroed header (sticky word off)
        return self.buf

    def reset(self) -> None:
        """Forget the sticky decision (e.g. after loading other weights)."""
        if self.buf is not None:
            self.buf[:16].zero_()


# ---------------------------------------------------------------------------------------------
# FP8 (OCP e4m3) projections -- SURVEY.md 8f-4; an explicit, lossy option (WanTransformer3DModel.enable_fp8_linear)
# ---------------------------------------------------------------------------------------------
FP8 = torch.float8_e4m3fn
FP8_MAX = 448.0


@_on_tensor_device
def ln_modulate_fp8(x: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor], add_one: bool,
                    rows_per_batch: int, eps: float, out: Optional[torch.Tensor] = None,
                    out_scale: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """``ln_modulate`` with per-token-row e4m3 quantisation fused in: returns (q [rows, dim] float8_e4m3fn, s fp32 [rows])
    with y ~= q * s[:, None]."""
    _need(x, torch.float32, "ln_modulate_fp8.x")
    if not x.is_contiguous() or x.dim() != 2:
        raise ValueError("ln_modulate_fp8.x must be a contiguous [rows, dim] tensor")
    rows, dim = x.shape
    for nm, t in (("scale", scale), ("shift", shift)):
        if t is not None:
            _need(t, torch.float32, "ln_modulate_fp8." + nm)
            if not t.is_contiguous() or t.shape[-1] != dim or t.numel() * rows_per_batch < rows * dim:
                raise ValueError(f"ln_modulate_fp8.{nm}: shape {tuple(t.shape)} does not cover {rows} rows of {dim}")
    if out is None:
        out = torch.empty(rows, dim, device=x.device, dtype=FP8)
    if out_scale is None:
        out_scale = torch.empty(rows, device=x.device, dtype=torch.float32)
    _need(out, FP8, "ln_modulate_fp8.out")
    _need(out_scale, torch.float32, "ln_modulate_fp8.out_scale")
    if not out.is_contiguous() or tuple(out.shape) != (rows, dim) or out_scale.numel() < rows:
        raise ValueError("ln_modulate_fp8: out must be contiguous [rows, dim] and out_scale hold one value per row")
    lib = _lib.load()
    _lib.check(lib.wan_ln_modulate_fp8(_p(x), _p(scale), _p(shift), int(bool(add_one)), _p(out), _p(out_scale), rows, dim,
                                       int(rows_per_batch), float(eps), _stream()), "wan_ln_modulate_fp8")
    return out, out_scale


@_on_tensor_device
def quantize_rows_fp8(x: torch.Tensor, out: Optional[torch.Tensor] = None,
                      out_scale: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """bf16 [rows, cols] -> (e4m3 [rows, cols], fp32 row scales = max|x| / 448)."""
    _need(x, torch.bfloat16, "quantize_rows_fp8.x")
    if x.dim() != 2:
        raise ValueError("quantize_rows_fp8.x must be 2-D")
    rows, cols = x.shape
    if out is None:
        out = torch.empty(rows, cols, device=x.device, dtype=FP8)
    if out_scale is None:
        out_scale = torch.empty(rows, device=x.device, dtype=torch.float32)
    _need(out, FP8, "quantize_rows_fp8.out")
    _need(out_scale, torch.float32, "quantize_rows_fp8.out_scale")
    lib = _lib.load()
    _lib.check(lib.wan_quantize_rows_fp8(_p(x), x.stride(0), _p(out), out.stride(0), _p(out_scale), rows, cols, _stream()),
               "wan_quantize_rows_fp8")
    return out, out_scale


def quantize_weight_fp8(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """One-off weight preparation (torch): [N, K] -> (e4m3 [N, K], fp32 per-output-channel scales = max|w[n,:]| / 448)."""
    wf = w.float()
    s = wf.abs().amax(dim=1).clamp_min(1e-12) / FP8_MAX
    return (wf / s[:, None]).clamp(-FP8_MAX, FP8_MAX).to(FP8).contiguous(), s.contiguous()


@_on_tensor_device
def gemm_fp8(a: torch.Tensor, a_scale: torch.Tensor, w: torch.Tensor, w_scale: torch.Tensor, bias: Optional[torch.Tensor],
             epilogue: int, out: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None,
             rows_per_batch: int = 0, ldo_t: int = 0) -> torch.Tensor:
    """acc = (a_q @ w_q.T) * a_scale[:, None] * w_scale[None, :] (+ bias) with the epilogues of ``gemm``.
    a e4m3 [M, K] (row stride free), w e4m3 [N, K]; K % 128 == 0."""
    _need(a, FP8, "gemm_fp8.a")
    _need(w, FP8, "gemm_fp8.w")
    _need(a_scale, torch.float32, "gemm_fp8.a_scale")
    _need(w_scale, torch.float32, "gemm_fp8.w_scale")
    M, K = a.shape
    N, Kw = w.shape
    if K != Kw or a_scale.numel() < M or w_scale.numel() != N:
        raise ValueError(f"gemm_fp8: a is [{M},{K}], w is [{N},{Kw}], scales {a_scale.numel()} / {w_scale.numel()}")
    if bias is not None:
        _need(bias, torch.float32, "gemm_fp8.bias")
        if bias.numel() != N:
            raise ValueError("gemm_fp8: bias length != N")
    dev = a.device
    if epilogue in (EPI_BF16, EPI_GELU_BF16):
        if out is None:
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        _need(out, torch.bfloat16, "gemm_fp8.out")
    elif epilogue == EPI_F32:
        if out is None:
            out = torch.empty(M, N, device=dev, dtype=torch.float32)
        _need(out, torch.float32, "gemm_fp8.out")
    elif epilogue == EPI_RESID_F32:
        if out is None:
            raise ValueError("gemm_fp8: EPI_RESID_F32 needs the residual stream as `out`")
        _need(out, torch.float32, "gemm_fp8.out")
        if gate is not None:
            _need(gate, torch.float32, "gemm_fp8.gate")
    elif epilogue == EPI_BF16_T:
        if out is None:
            out = torch.zeros(N, ldo_t or round_up(M, 64), device=dev, dtype=torch.bfloat16)
        _need(out, torch.bfloat16, "gemm_fp8.out")
        if out.shape[0] != N or out.shape[1] < M:
            raise ValueError(f"gemm_fp8: transposed out must be [N={N}, >= M={M}], got {tuple(out.shape)}")
    else:
        raise ValueError(f"gemm_fp8: unknown epilogue {epilogue}")
    if epilogue != EPI_BF16_T and tuple(out.shape) != (M, N):
        raise ValueError(f"gemm_fp8: out shape {tuple(out.shape)} != ({M},{N})")
    lib = _lib.load()
    # the persistent stream-K kernel's e4m3 instantiation where the plan says so (its workspace is the bf16 GEMM's: one per stream)
    ws = gemm_workspace(dev, M, N, K, fp8=True)
    _lib.check(lib.wan_gemm_fp8_ws(_p(a), a.stride(0), _p(a_scale), _p(w), w.stride(0), _p(w_scale), _p(bias), _p(out),
                                   out.stride(0), M, N, K, epilogue, _p(gate), int(rows_per_batch), _p(ws),
                                   ws.numel() if ws is not None else 0, _stream()), "wan_gemm_fp8_ws")
    return out


_ATTN_WS = {}          # ad-hoc callers without their own workspace: one per (device, stream)


@_on_tensor_device
def attention_fwd(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, num_heads: int, k_len: Optional[int] = None,
                  softmax_scale: Optional[float] = None, out: Optional[torch.Tensor] = None,
                  q_prescaled: bool = False, workspace: Optional[AttentionWorkspace] = None,
                  k_lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q bf16 [B,Lq,H*128], k bf16 [B,Lk,H*128], vt bf16 [B,H*128,ldvt] (V transposed, ldvt >=
    roundup(k_len,64), finite padding) -> bf16 [B,Lq,H*128].  Keys >= k_len are masked.
    ``q_prescaled``: q already carries softmax_scale*log2(e) (see ``rmsnorm_rope_``'s x0_scale).
    ``workspace``: the call site's ``AttentionWorkspace`` (default: one per device and stream).
    ``k_lens``: int32 [B] ON THE DEVICE -- a ragged batch in one launch (``wan_attention_fwd_varlen``): sample b attends keys
    [0, k_lens[b]); the kernel reads the counts, the host never does."""
    for nm, t in (("q", q), ("k", k), ("vt", vt)):
        _need(t, torch.bfloat16, "attention." + nm)
        if t.dim() != 3:
            raise ValueError(f"attention.{nm} must be 3-D [B, rows, cols]")
    B, Lq, C = q.shape
    Lk = k.shape[1] if k_len is None else int(k_len)
    if Lk > k.shape[1] or Lk <= 0:
        raise ValueError(f"attention: k_len={Lk} outside (0, {k.shape[1]}]")
    head_dim = C // num_heads
    if out is None:
        out = torch.empty(B, Lq, C, device=q.device, dtype=torch.bfloat16)
    _need(out, torch.bfloat16, "attention.out")
    if vt.shape[1] != C or k.shape[2] != C or k.shape[0] != B or vt.shape[0] != B:
        raise ValueError("attention: q/k/vt shapes disagree")
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(head_dim)
    lib = _lib.load()
    ws, ws_bytes = None, int(lib.wan_attention_workspace_bytes(B, Lq, Lk, num_heads, head_dim))
    if ws_bytes > 0:
        if workspace is None:
            key = (q.device, _stream())
            workspace = _ATTN_WS.get(key)
            if workspace is None:
                workspace = _ATTN_WS[key] = AttentionWorkspace()
        ws = workspace.get(q.device, ws_bytes)
    if k_lens is not None:
        _need(k_lens, torch.int32, "attention.k_lens")
        if k_lens.shape != (B,) or not k_lens.is_contiguous():
            raise ValueError(f"attention.k_lens must be a contiguous int32 [{B}]")
        _lib.check(lib.wan_attention_fwd_varlen(_p(q), q.stride(1), q.stride(0), _p(k), k.stride(1), k.stride(0),
                                                _p(vt), vt.stride(1), vt.stride(0), _p(out), out.stride(1), out.stride(0),
                                                B, Lq, Lk, _p(k_lens), num_heads, head_dim, float(scale),
                                                _lib.ATTN_Q_PRESCALED if q_prescaled else 0, _p(ws),
                                                ws_bytes if ws is not None else 0, _stream()), "wan_attention_fwd_varlen")
        return out
    _lib.check(lib.wan_attention_fwd(_p(q), q.stride(1), q.stride(0), _p(k), k.stride(1), k.stride(0),
                                     _p(vt), vt.stride(1), vt.stride(0), _p(out), out.stride(1), out.stride(0),
                                     B, Lq, Lk, num_heads, head_dim, float(scale),
                                     _lib.ATTN_Q_PRESCALED if q_prescaled else 0, _p(ws), ws_bytes if ws is not None else 0,
                                     _stream()), "wan_attention_fwd")
    return out


@_on_tensor_device
def rmsnorm_rope_fp8(x0: torch.Tensor, w0: torch.Tensor, x1: Optional[torch.Tensor], w1: Optional[torch.Tensor], head_dim: int,
                     eps: float, rope: Optional[Tuple[torch.Tensor, torch.Tensor]], rope_params: Optional[RopeParams],
                     out0: torch.Tensor, out1: Optional[torch.Tensor], x0_scale: float = 1.0, x1_scale: float = 1.0) -> None:
    """``rmsnorm_rope_`` that leaves x0 / x1 untouched and writes OCP e4m3 copies of the results (dense ``FP8`` [rows, dim]):
    ``out = e4m3(bf16(result * scale))`` -- the operands of ``attention_fwd_qk8`` (include/wan_hip.h, a9')."""
    _need(x0, torch.bfloat16, "rmsnorm_rope_fp8.x0")
    _need(w0, torch.float32, "rmsnorm_rope_fp8.w0")
    rows, dim = x0.shape
    ld = x0.stride(0)
    for nm, t in (("out0", out0), ("out1", out1)):
        if t is not None:
            _need(t, FP8, "rmsnorm_rope_fp8." + nm)
            if not t.is_contiguous() or t.numel() < rows * dim:
                raise ValueError(f"rmsnorm_rope_fp8.{nm} must be contiguous with >= rows * dim bytes")
    if x1 is not None:
        _need(x1, torch.bfloat16, "rmsnorm_rope_fp8.x1")
        _need(w1, torch.float32, "rmsnorm_rope_fp8.w1")
        if x1.shape != x0.shape or x1.stride(0) != ld or out1 is None:
            raise ValueError("rmsnorm_rope_fp8: x0 / x1 must share shape and row stride, and x1 needs out1")
    cos = sin = None
    if rope is not None:
        cos, sin = rope
        if rope_params is None:
            raise ValueError("rmsnorm_rope_fp8: rope tables given without rope_params")
    lib = _lib.load()
    _lib.check(lib.wan_rmsnorm_rope_fp8(_p(x0), _p(w0), _p(x1), _p(w1), ld, rows, dim, head_dim, float(eps), _p(cos), _p(sin),
                                        ctypes.byref(rope_params) if rope_params is not None else None, float(x0_scale),
                                        float(x1_scale), _p(out0), _p(out1), _stream()), "wan_rmsnorm_rope_fp8")


@_on_tensor_device
def vt_quantize_mx(vt: torch.Tensor, num_heads: int, k_len: int, v8: Optional[torch.Tensor] = None,
                   scales: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """V^T bf16 [B, H*128, ldvt] -> (MX e4m3 [B, H*128, ldvt] with the keys of every 64-tile in the attention kernel's register order,
    uint8 E8M0 scales, one per channel row and 32 consecutive keys): the V operand of ``attention_fwd_f8`` (include/wan_hip.h a9'')."""
    _need(vt, torch.bfloat16, "vt_quantize_mx.vt")
    if vt.dim() != 3:
        raise ValueError("vt_quantize_mx.vt must be [B, H*128, ldvt]")
    B, C, ld = vt.shape
    lib = _lib.load()
    nsc = int(lib.wan_vt_mx_scale_bytes(B, num_heads, int(k_len)))
    if v8 is None:
        v8 = torch.empty(B, C, ld, device=vt.device, dtype=FP8)
    if scales is None:
        scales = torch.empty(nsc, device=vt.device, dtype=torch.uint8)
    _need(v8, FP8, "vt_quantize_mx.v8")
    _need(scales, torch.uint8, "vt_quantize_mx.scales")
    if v8.shape != vt.shape or scales.numel() < nsc or not scales.is_contiguous():
        raise ValueError("vt_quantize_mx: v8 must have vt's shape and scales wan_vt_mx_scale_bytes() contiguous bytes")
    _lib.check(lib.wan_vt_quantize_mx(_p(vt), vt.stride(1), vt.stride(0), B, int(num_heads), int(k_len), _p(v8), v8.stride(1), v8.stride(0),
                                      _p(scales), _stream()), "wan_vt_quantize_mx")
    return v8, scales


@_on_tensor_device
def attention_fwd_f8(q8: torch.Tensor, k8: torch.Tensor, v8: torch.Tensor, v8_scales: torch.Tensor, vt: torch.Tensor, num_heads: int,
                     q_exp: int, k_exp: int, k_len: Optional[int] = None, out: Optional[torch.Tensor] = None,
                     workspace: Optional[AttentionWorkspace] = None) -> torch.Tensor:
    """Self-attention with BOTH products on the fp8 matrix pipe (LOSSY, opt-in; include/wan_hip.h a9'').  q8 / k8 as
    ``attention_fwd_qk8``; v8 / v8_scales from ``vt_quantize_mx(vt, ...)``; the bf16 ``vt`` serves the workgroups whose max-free
    softmax check fails (and the split-KV tail round)."""
    for nm, t, dt in (("q8", q8, FP8), ("k8", k8, FP8), ("v8", v8, FP8), ("vt", vt, torch.bfloat16)):
        _need(t, dt, "attention_f8." + nm)
        if t.dim() != 3:
            raise ValueError(f"attention_f8.{nm} must be 3-D [B, rows, cols]")
    _need(v8_scales, torch.uint8, "attention_f8.v8_scales")
    B, Lq, C = q8.shape
    Lk = k8.shape[1] if k_len is None else int(k_len)
    if Lk > k8.shape[1] or Lk <= 0:
        raise ValueError(f"attention_f8: k_len={Lk} outside (0, {k8.shape[1]}]")
    head_dim = C // num_heads
    if out is None:
        out = torch.empty(B, Lq, C, device=q8.device, dtype=torch.bfloat16)
    _need(out, torch.bfloat16, "attention_f8.out")
    if vt.shape[1] != C or v8.shape != vt.shape or k8.shape[2] != C or k8.shape[0] != B or vt.shape[0] != B:
        raise ValueError("attention_f8: q8/k8/v8/vt shapes disagree")
    lib = _lib.load()
    ws_bytes = max(int(lib.wan_attention_workspace_bytes(B, Lq, Lk, num_heads, head_dim)), 16)
    if workspace is None:
        key = (q8.device, _stream())
        workspace = _ATTN_WS.get(key)
        if workspace is None:
            workspace = _ATTN_WS[key] = AttentionWorkspace()
    ws = workspace.get(q8.device, ws_bytes)
    _lib.check(lib.wan_attention_fwd_f8(_p(q8), q8.stride(1), q8.stride(0), int(q_exp), _p(k8), k8.stride(1), k8.stride(0), int(k_exp),
                                        _p(v8), v8.stride(1), v8.stride(0), _p(v8_scales), _p(vt), vt.stride(1), vt.stride(0),
                                        _p(out), out.stride(1), out.stride(0), B, Lq, Lk, num_heads, head_dim, _p(ws), ws_bytes, _stream()),
               "wan_attention_fwd_f8")
    return out


@_on_tensor_device
def col_mean(x: torch.Tensor, rows_per_batch: int, valid_rows: int, batch: int, out: Optional[torch.Tensor] = None,
             workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 [batch, dim]: mean over the first ``valid_rows`` rows of every sample of bf16 x [batch * rows_per_batch, dim] (row
    stride free).  Two-stage, fixed summation order.  ``workspace``: fp32, >= wan_col_mean_workspace_bytes / 4 elements."""
    _need(x, torch.bfloat16, "col_mean.x")
    dim = x.shape[1]
    lib = _lib.load()
    need = int(lib.wan_col_mean_workspace_bytes(batch, dim)) // 4
    if workspace is None:
        workspace = torch.empty(need, device=x.device, dtype=torch.float32)
    _need(workspace, torch.float32, "col_mean.workspace")
    if workspace.numel() < need or not workspace.is_contiguous():
        raise ValueError(f"col_mean.workspace needs {need} contiguous fp32 elements")
    if out is None:
        out = torch.empty(batch, dim, device=x.device, dtype=torch.float32)
    _need(out, torch.float32, "col_mean.out")
    _lib.check(lib.wan_col_mean_bf16(_p(x), x.stride(0), int(rows_per_batch), int(valid_rows), int(batch), dim, _p(workspace), _p(out),
                                     _stream()), "wan_col_mean_bf16")
    return out


@_on_tensor_device
def qk_quantize_fp8(q: Optional[torch.Tensor], k: Optional[torch.Tensor], rows_per_batch: int, k_mean: Optional[torch.Tensor],
                    q_scale: float, k_scale: float, q8: Optional[torch.Tensor], k8: Optional[torch.Tensor]) -> None:
    """q8 = e4m3(q * q_scale), k8 = e4m3((k - k_mean[sample]) * k_scale) from bf16 [rows, dim] views (when both are given they
    share shape and row stride; ``k_mean`` fp32 [batch, dim] or None); dense ``FP8`` [rows, dim] outputs.  Either operand may be
    None together with its output (the Ulysses head-group pipeline quantises k once and every q group on arrival)."""
    if (q is None) != (q8 is None) or (k is None) != (k8 is None) or (q is None and k is None):
        raise ValueError("qk_quantize_fp8: q comes with q8, k with k8, and at least one pair is needed")
    if k is None and k_mean is not None:
        raise ValueError("qk_quantize_fp8: k_mean without k")
    ref = q if q is not None else k
    for nm, t in (("q", q), ("k", k)):
        if t is not None:
            _need(t, torch.bfloat16, "qk_quantize_fp8." + nm)
            if t.dim() != 2 or t.shape != ref.shape or t.stride(0) != ref.stride(0):
                raise ValueError("qk_quantize_fp8: q and k must be 2-D and share shape and row stride")
    rows, dim = ref.shape
    for nm, t in (("q8", q8), ("k8", k8)):
        if t is None:
            continue
        _need(t, FP8, "qk_quantize_fp8." + nm)
        if not t.is_contiguous() or t.numel() < rows * dim:
            raise ValueError(f"qk_quantize_fp8.{nm} must be contiguous with >= rows * dim bytes")
    if k_mean is not None:
        _need(k_mean, torch.float32, "qk_quantize_fp8.k_mean")
        if not k_mean.is_contiguous() or k_mean.shape[-1] != dim:
            raise ValueError("qk_quantize_fp8.k_mean must be contiguous [batch, dim]")
    lib = _lib.load()
    _lib.check(lib.wan_qk_quantize_fp8(_p(q), _p(k), ref.stride(0), rows, dim, int(rows_per_batch), _p(k_mean), float(q_scale),
                                       float(k_scale), _p(q8), _p(k8), _stream()), "wan_qk_quantize_fp8")


@_on_tensor_device
def attention_fwd_qk8(q8: torch.Tensor, k8: torch.Tensor, vt: torch.Tensor, num_heads: int, q_exp: int, k_exp: int,
                      k_len: Optional[int] = None, out: Optional[torch.Tensor] = None,
                      workspace: Optional[AttentionWorkspace] = None) -> torch.Tensor:
    """Self-attention with QK^T on the fp8 matrix pipe (LOSSY, opt-in; include/wan_hip.h a9').
    q8 e4m3 [B,Lq,H*128] = e4m3(q * softmax_scale * log2(e) * 2^q_exp), k8 e4m3 [B,Lk,H*128] = e4m3(k * 2^k_exp)
    (``rmsnorm_rope_fp8`` writes both), vt / out as ``attention_fwd``."""
    for nm, t, dt in (("q8", q8, FP8), ("k8", k8, FP8), ("vt", vt, torch.bfloat16)):
        _need(t, dt, "attention_qk8." + nm)
        if t.dim() != 3:
            raise ValueError(f"attention_qk8.{nm} must be 3-D [B, rows, cols]")
    B, Lq, C = q8.shape
    Lk = k8.shape[1] if k_len is None else int(k_len)
    if Lk > k8.shape[1] or Lk <= 0:
        raise ValueError(f"attention_qk8: k_len={Lk} outside (0, {k8.shape[1]}]")
    head_dim = C // num_heads
    if out is None:
        out = torch.empty(B, Lq, C, device=q8.device, dtype=torch.bfloat16)
    _need(out, torch.bfloat16, "attention_qk8.out")
    if vt.shape[1] != C or k8.shape[2] != C or k8.shape[0] != B or vt.shape[0] != B:
        raise ValueError("attention_qk8: q8/k8/vt shapes disagree")
    lib = _lib.load()
    ws, ws_bytes = None, int(lib.wan_attention_workspace_bytes(B, Lq, Lk, num_heads, head_dim))
    if ws_bytes > 0:
        if workspace is None:
            key = (q8.device, _stream())
            workspace = _ATTN_WS.get(key)
            if workspace is None:
                workspace = _ATTN_WS[key] = AttentionWorkspace()
        ws = workspace.get(q8.device, ws_bytes)
    _lib.check(lib.wan_attention_fwd_qk8(_p(q8), q8.stride(1), q8.stride(0), int(q_exp), _p(k8), k8.stride(1), k8.stride(0), int(k_exp),
                                         _p(vt), vt.stride(1), vt.stride(0), _p(out), out.stride(1), out.stride(0),
                                         B, Lq, Lk, num_heads, head_dim, _p(ws), ws_bytes if ws is not None else 0, _stream()),
               "wan_attention_fwd_qk8")
    return out


@_on_tensor_device
def transpose_pad(v: torch.Tensor, ldt: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 [rows, cols] -> [cols, ldt] (zero padded columns), ldt default roundup(rows, 64)."""
    _need(v, torch.bfloat16, "transpose.v")
    rows, cols = v.shape
    ldt = ldt or round_up(rows, 64)
    if out is None:
        out = torch.empty(cols, ldt, device=v.device, dtype=torch.bfloat16)
    _need(out, torch.bfloat16, "transpose.out")
    lib = _lib.load()
    _lib.check(lib.wan_transpose_bf16(_p(v), v.stride(0), _p(out), out.stride(0), rows, cols, _stream()),
               "wan_transpose_bf16")
    return out


@_on_tensor_device
def patchify(latent: torch.Tensor, patch: Tuple[int, int, int], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """latent [Cin,F,H,W] fp32|bf16 -> bf16 tokens [L, Cin*pt*ph*pw]."""
    if latent.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError(f"patchify: latent dtype {latent.dtype} not supported (fp32 or bf16)")
    _need(latent, latent.dtype, "patchify.latent")
    if not latent.is_contiguous() or latent.dim() != 4:
        raise ValueError("patchify: latent must be contiguous [Cin,F,H,W]")
    Cin, F, H, W = latent.shape
    pt, ph, pw = patch
    L = (F // pt) * (H // ph) * (W // pw)
    K = Cin * pt * ph * pw
    if out is None:
        out = torch.empty(L, K, device=latent.device, dtype=torch.bfloat16)
    _need(out, torch.bfloat16, "patchify.out")
    lib = _lib.load()
    _lib.check(lib.wan_patchify(_p(latent), 0 if latent.dtype == torch.float32 else 1, _p(out), out.stride(0),
                                Cin, F, H, W, pt, ph, pw, _stream()), "wan_patchify")
    return out


@_on_tensor_device
def unpatchify(tokens: torch.Tensor, grid: Tuple[int, int, int], patch: Tuple[int, int, int], cout: int,
               out_dtype: torch.dtype, zero_frames: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tokens fp32 [L, pt*ph*pw*Cout] -> [Cout, F*pt, Hp*ph, Wp*pw] in out_dtype (fp32|bf16); output frames
    < zero_frames are written as zeros (the CoF mask of pipeline_wan.py:736)."""
    _need(tokens, torch.float32, "unpatchify.tokens")
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise ValueError(f"unpatchify: out dtype {out_dtype} not supported")
    F, Hp, Wp = grid
    pt, ph, pw = patch
    if tokens.shape[0] < F * Hp * Wp:
        raise ValueError("unpatchify: fewer token rows than the grid")
    shape = (cout, F * pt, Hp * ph, Wp * pw)
    if out is None:
        out = torch.empty(shape, device=tokens.device, dtype=out_dtype)
    _need(out, out_dtype, "unpatchify.out")
    if tuple(out.shape) != shape or not out.is_contiguous():
        raise ValueError(f"unpatchify: out must be contiguous {shape}")
    lib = _lib.load()
    _lib.check(lib.wan_unpatchify(_p(tokens), tokens.stride(0), _p(out), 0 if out_dtype == torch.float32 else 1,
                                  cout, F, Hp, Wp, pt, ph, pw, int(zero_frames), _stream()), "wan_unpatchify")
    return out


# ---------------------------------------------------------------------------------------------
# WanVAE kernels (channels-last bf16 activations [T, H, W, C])
# ---------------------------------------------------------------------------------------------
@_on_tensor_device
def conv_cl(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], cout: int, kernel, stride=(1, 1, 1),
            pad=(0, 0, 0), out_thw=None, hist: Optional[torch.Tensor] = None, upsample2x: bool = False,
            time_interleave: bool = False, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x bf16 [T,H,W,Cin]; w bf16 [Cout, Kpad] packed (kt,kh,kw,ci); hist bf16 [n<=2,H,W,Cin] or None.
    Returns bf16 [T_out,H_out,W_out,Cout] (time_interleave: [2*T_out,H_out,W_out,Cout/2])."""
    _need(x, torch.bfloat16, "conv_cl.x")
    _need(w, torch.bfloat16, "conv_cl.w")
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError("conv_cl.x must be contiguous [T,H,W,C]")
    T, H, W, Cin = x.shape
    To, Ho, Wo = out_thw
    nh = 0
    if hist is not None:
        _need(hist, torch.bfloat16, "conv_cl.hist")
        if not hist.is_contiguous() or tuple(hist.shape[1:]) != (H, W, Cin):
            raise ValueError("conv_cl.hist must be contiguous [n,H,W,Cin]")
        nh = hist.shape[0]
    if bias is not None:
        _need(bias, torch.float32, "conv_cl.bias")
    ch = cout // 2 if time_interleave else cout
    out = torch.empty((2 * To if time_interleave else To), Ho, Wo, ch, device=x.device, dtype=torch.bfloat16)
    if resid is not None:
        _need(resid, torch.bfloat16, "conv_cl.resid")
        if resid.shape != out.shape or not resid.is_contiguous():
            raise ValueError("conv_cl.resid must match the output")
    p = _lib.ConvParams(T, H, W, Cin, To, Ho, Wo, cout, kernel[0], kernel[1], kernel[2], stride[0], stride[1], stride[2],
                        pad[0], pad[1], pad[2], int(upsample2x), int(time_interleave))
    lib = _lib.load()
    _lib.check(lib.wan_conv_cl(_p(x), _p(hist), nh, _p(w), w.stride(0), _p(bias), _p(resid), _p(out), ch,
                               ctypes.byref(p), _stream()), "wan_conv_cl")
    return out


@_on_tensor_device
def rmsnorm_silu_cl(x: torch.Tensor, gamma: torch.Tensor, silu: bool) -> torch.Tensor:
    _need(x, torch.bfloat16, "rmsnorm_silu_cl.x")
    _need(gamma, torch.float32, "rmsnorm_silu_cl.gamma")
    if not x.is_contiguous():
        raise ValueError("rmsnorm_silu_cl.x must be contiguous")
    C = x.shape[-1]
    out = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.wan_rmsnorm_silu_cl(_p(x), _p(gamma), _p(out), x.numel() // C, C, int(silu), _stream()),
               "wan_rmsnorm_silu_cl")
    return out


@_on_tensor_device
def softmax_rows(scores: torch.Tensor, n: int, npad: int, scale: float) -> torch.Tensor:
    _need(scores, torch.float32, "softmax_rows.scores")
    rows = scores.shape[0]
    out = torch.empty(rows, npad, device=scores.device, dtype=torch.bfloat16)
    lib = _lib.load()
    _lib.check(lib.wan_softmax_rows(_p(scores), scores.stride(0), _p(out), npad, rows, n, npad, float(scale), _stream()),
               "wan_softmax_rows")
    return out


@_on_tensor_device
def video_to_cl(video: torch.Tensor, cpad: int = 8) -> torch.Tensor:
    """[C,T,H,W] fp32|bf16 -> bf16 [T,H,W,cpad] (extra channels zero)."""
    if video.dtype not in (torch.float32, torch.bfloat16):
        video = video.float()
    _need(video, video.dtype, "video_to_cl.video")
    video = video.contiguous()
    C, T, H, W = video.shape
    out = torch.empty(T, H, W, cpad, device=video.device, dtype=torch.bfloat16)
    lib = _lib.load()
    _lib.check(lib.wan_video_to_cl(_p(video), 0 if video.dtype == torch.float32 else 1, _p(out), C, cpad, T * H * W,
                                   _stream()), "wan_video_to_cl")
    return out


@_on_tensor_device
def cl_to_video(x: torch.Tensor, cv: int, out_dtype: torch.dtype, clamp: bool) -> torch.Tensor:
    """bf16 [T,H,W,C>=cv] -> [cv,T,H,W] in out_dtype (fp32|bf16), optional clamp to [-1,1]."""
    _need(x, torch.bfloat16, "cl_to_video.x")
    if not x.is_contiguous():
        raise ValueError("cl_to_video.x must be contiguous")
    T, H, W, C = x.shape
    dt = out_dtype if out_dtype in (torch.float32, torch.bfloat16) else torch.float32
    out = torch.empty(cv, T, H, W, device=x.device, dtype=dt)
    lib = _lib.load()
    _lib.check(lib.wan_cl_to_video(_p(x), C, _p(out), 0 if dt == torch.float32 else 1, cv, T * H * W, int(clamp),
                                   _stream()), "wan_cl_to_video")
    return out.to(out_dtype)


@_on_tensor_device
def lincomb(terms, out_dtype: torch.dtype) -> torch.Tensor:
    """sum_i c_i * x_i over <= 4 same-shape CUDA tensors in one pass (fp32 accumulate); `terms` is a list
    of (coefficient, tensor).  Inputs are brought to `out_dtype` (fp32 or bf16) if they differ."""
    terms = [(float(c), t) for c, t in terms if t is not None and c != 0.0]
    if not terms or len(terms) > 4:
        raise ValueError("lincomb needs 1..4 non-zero terms")
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise ValueError(f"lincomb: dtype {out_dtype} not supported")
    xs = []
    for _, t in terms:
        _need(t, t.dtype, "lincomb.x")
        xs.append(t.to(out_dtype).contiguous())
    out = torch.empty_like(xs[0])
    cs = [c for c, _ in terms] + [0.0] * (4 - len(terms))
    ps = [_p(x) for x in xs] + [None] * (4 - len(xs))
    lib = _lib.load()
    _lib.check(lib.wan_lincomb(_p(out), 0 if out_dtype == torch.float32 else 1, ps[0], ps[1], ps[2], ps[3],
                               cs[0], cs[1], cs[2], cs[3], out.numel(), _stream()), "wan_lincomb")
    return out


# ------------------------------------------------------------------ umT5 text encoder rows
def _avail(t: torch.Tensor) -> int:
    """Elements from t's first element to the end of its storage."""
    return t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()


@_on_tensor_device
def gemm_batched(a: torch.Tensor, stride_a: int, w: torch.Tensor, stride_w: int, out: torch.Tensor, stride_o: int,
                 M: int, N: int, K: int, batch: int, epilogue: int) -> torch.Tensor:
    """`batch` independent products out_z[m,n] = sum_k a_z[m,k] * w_z[n,k] (one per attention head).
    a / w / out are 2-D views (unit inner stride) of problem 0; problem z starts `stride_*` elements later
    in the same storage.  EPI_BF16 or EPI_F32."""
    _need(a, torch.bfloat16, "gemm_batched.a")
    _need(w, torch.bfloat16, "gemm_batched.w")
    _need(out, torch.float32 if epilogue == EPI_F32 else torch.bfloat16, "gemm_batched.out")
    for nm, t in (("a", a), ("w", w), ("out", out)):
        if t.dim() != 2 or t.stride(1) != 1:
            raise ValueError(f"gemm_batched.{nm} must be a 2-D view with unit inner stride")
    lda, ldw, ldo = a.stride(0), w.stride(0), out.stride(0)
    if min(batch, M, N, K) < 1 or min(stride_a, stride_w, stride_o) < 0:
        raise ValueError("gemm_batched: empty problem or negative stride")
    if (_avail(a) < (batch - 1) * stride_a + (M - 1) * lda + K or _avail(w) < (batch - 1) * stride_w + (N - 1) * ldw + K
            or _avail(out) < (batch - 1) * stride_o + (M - 1) * ldo + N):
        raise ValueError("gemm_batched: the strided problem runs past the end of an operand's storage")
    lib = _lib.load()
    _lib.check(lib.wan_gemm_bf16_batched(_p(a), lda, stride_a, _p(w), ldw, stride_w, _p(out), ldo, stride_o,
                                         M, N, K, batch, epilogue, _stream()), "wan_gemm_bf16_batched")
    return out


@_on_tensor_device
def embedding_rows(ids: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """ids int64 [n] -> fp32 [n, dim] rows of the bf16 table [vocab, dim]."""
    _need(ids, torch.int64, "embedding.ids")
    _need(table, torch.bfloat16, "embedding.table")
    if ids.dim() != 1 or table.dim() != 2 or not (ids.is_contiguous() and table.is_contiguous()):
        raise ValueError("embedding: ids must be contiguous [n] and table contiguous [vocab, dim]")
    out = torch.empty(ids.numel(), table.shape[1], device=ids.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.wan_embedding_rows(_p(ids), _p(table), table.shape[0], _p(out), ids.numel(), table.shape[1],
                                      _stream()), "wan_embedding_rows")
    return out


@_on_tensor_device
def rmsnorm_rows(x: torch.Tensor, w: torch.Tensor, eps: float, out_dtype: torch.dtype = torch.bfloat16,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """T5LayerNorm of fp32 rows [rows, dim] -> bf16 (GEMM input) or fp32."""
    _need(x, torch.float32, "rmsnorm_rows.x")
    _need(w, torch.float32, "rmsnorm_rows.w")
    if x.dim() != 2 or not x.is_contiguous() or w.numel() != x.shape[1]:
        raise ValueError("rmsnorm_rows: x must be contiguous [rows, dim] and w [dim]")
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("rmsnorm_rows: out_dtype must be float32 or bfloat16")
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    _need(out, out_dtype, "rmsnorm_rows.out")
    lib = _lib.load()
    _lib.check(lib.wan_rmsnorm_rows(_p(x), _p(w), _p(out), 0 if out_dtype == torch.float32 else 1, x.shape[0],
                                    x.shape[1], float(eps), _stream()), "wan_rmsnorm_rows")
    return out


@_on_tensor_device
def t5_softmax_bias(scores: torch.Tensor, table: torch.Tensor, lut: torch.Tensor, num_heads: int, k_len: int,
                    npad: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """scores fp32 [H, L, L] -> probs bf16 [H, L, npad] = softmax_j(scores + rel-pos bias), keys >= k_len masked."""
    _need(scores, torch.float32, "t5_softmax.scores")
    _need(table, torch.float32, "t5_softmax.table")
    _need(lut, torch.int32, "t5_softmax.lut")
    H, Lq, Lk = scores.shape
    if H != num_heads or not scores.is_contiguous() or table.shape != (table.shape[0], H) or not table.is_contiguous():
        raise ValueError("t5_softmax: scores must be contiguous [H, Lq, Lk], table contiguous [num_buckets, H]")
    if lut.numel() != Lq + Lk - 1:
        raise ValueError("t5_softmax: lut must have Lq + Lk - 1 entries")
    if out is None:
        out = torch.empty(H, Lq, npad, device=scores.device, dtype=torch.bfloat16)
    _need(out, torch.bfloat16, "t5_softmax.out")
    lib = _lib.load()
    _lib.check(lib.wan_t5_softmax_bias(_p(scores), Lk, _p(table), _p(lut), _p(out), out.stride(1), H, Lq, Lk,
                                       int(k_len), int(npad), _stream()), "wan_t5_softmax_bias")
    return out


@_on_tensor_device
def mul_bf16(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need(a, torch.bfloat16, "mul.a")
    _need(b, torch.bfloat16, "mul.b")
    if a.shape != b.shape or not (a.is_contiguous() and b.is_contiguous()):
        raise ValueError("mul_bf16: operands must be contiguous and of equal shape")
    if out is None:
        out = torch.empty_like(a)
    _need(out, torch.bfloat16, "mul.out")
    lib = _lib.load()
    _lib.check(lib.wan_mul_bf16(_p(a), _p(b), _p(out), a.numel(), _stream()), "wan_mul_bf16")
    return out
