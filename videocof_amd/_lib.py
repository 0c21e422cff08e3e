"""ctypes binding of ``libwan_hip.so`` (C ABI declared in ``include/wan_hip.h``).

The library is built in-tree by ``make`` / ``__graft_entry__.build()``.  There is
NO fallback: if the shared object is missing or a symbol is absent, importing
the ops raises ``RuntimeError`` -- the product path never runs without the HIP
kernels.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# WAN_HIP_LIB: a developer A/B hook -- load ANOTHER build of the library (e.g. the tree of an earlier commit, tools/exp_lib/) through the
# same binding, so that two builds can be timed back to back on one box.  Unset in every product / test / benchmark path.
LIB_PATH = os.environ.get("WAN_HIP_LIB") or os.path.join(_HERE, "libwan_hip.so")
ABI_VERSION = 11
ATTN_Q_PRESCALED = 1          # wan_attention_fwd flag (include/wan_hip.h)
LOG2E = 1.4426950408889634
# wan_get_tuning("last_attn_variant") (include/wan_hip.h, WAN_ATTN_VARIANT_*)
ATTN_VARIANT_NAMES = {1: "attn_fwd_w4_kernel<.,.,ref> (4-wave, lazy softmax reference)",
                      2: "attn_fwd_w4_kernel<.,false,0> (4-wave, max-free attempt) + attn_fwd_w4_kernel<.,false,1,true> (lazy-reference fix-up of flagged workgroups)",
                      3: "(retired in round 5: the 8-wave running-max kernel)",
                      4: "attn_fwd_w4_kernel<0,.,1,false,true> (4-wave, lazy softmax reference, QK^T on the fp8 matrix pipe)",
                      5: "attn_fwd_f8_kernel (4-wave, QK^T and P.V on the fp8 matrix pipe, checked max-free softmax) + "
                         "attn_fwd_w4_kernel<0,false,1,true,true> (fp8-QK^T lazy-reference fix-up of flagged workgroups)"}
ATTN_VARIANT_XCD_PINNED, ATTN_VARIANT_SPLIT_TAIL = 16, 32
GEMM_VARIANT_KERNELS = {0: "gemm_bf16_kernel", 1: "gemm256_kernel", 2: "gemm_w4_kernel", 3: "gemm_pk_kernel"}      # wan_gemm_plan / wan_gemm_ws_plan (WAN_GEMM_VARIANT_*)


def attn_variant_name(code: int) -> str:
    name = ATTN_VARIANT_NAMES.get(code & 15, f"unknown({code})")
    if code & ATTN_VARIANT_XCD_PINNED:
        name += ", heads pinned to XCDs"
    if code & ATTN_VARIANT_SPLIT_TAIL:
        name += " + split-KV tail round (4-wave SPLIT kernel + attn_combine_kernel)"
    return name

WAN_OK, WAN_ERR_INVALID, WAN_ERR_UNSUPPORTED, WAN_ERR_LAUNCH = 0, 1, 2, 3
EPI_BF16, EPI_GELU_BF16, EPI_F32, EPI_RESID_F32, EPI_BF16_T = 0, 1, 2, 3, 4


class RopeParams(Structure):
    """``wan_rope_params`` of include/wan_hip.h."""
    _fields_ = [("F", c_int), ("Hp", c_int), ("Wp", c_int), ("mode", c_int), ("f_src", c_int),
                ("ground_end", c_int), ("token_offset", c_int64), ("rows_per_batch", c_int64),
                ("max_pos", c_int)]


class BoxProbeResult(Structure):
    """``wan_box_probe_result`` of include/wan_hip.h."""
    _fields_ = [("mfma_mix_tflops", c_float), ("copy_tbps", c_float), ("mfma_ms", c_float), ("copy_ms", c_float)]


class ConvParams(Structure):
    """``wan_conv_params`` of include/wan_hip.h."""
    _fields_ = [(n, c_int) for n in ("T_in", "H_in", "W_in", "Cin", "T_out", "H_out", "W_out", "Cout",
                                     "KT", "KH", "KW", "st", "sh", "sw", "pt", "ph", "pw",
                                     "upsample2x", "time_interleave")]


class BlockWeights(Structure):
    """``wan_block_weights`` of include/wan_hip.h."""
    _fields_ = ([(n, c_int) for n in ("dim", "ffn_dim", "num_heads", "text_len")] + [("eps", c_float)] +
                [(n, c_void_p) for n in ("w_qk", "w_v", "w_o", "w_cq", "w_co", "w_ffn0", "w_ffn2",
                                         "b_qk", "b_v", "b_o", "b_cq", "b_co", "b_ffn0", "b_ffn2",
                                         "norm_q", "norm_k", "norm_cq", "norm3_w", "norm3_b")])


class BlockWorkspace(Structure):
    """``wan_block_workspace`` of include/wan_hip.h."""
    _fields_ = ([(n, c_void_p) for n in ("h", "qk", "att", "cq", "ff", "vt")] + [("ldvt", c_int64)] +
                [("attn_ws_self", c_void_p), ("attn_ws_self_bytes", c_int64),
                 ("attn_ws_cross", c_void_p), ("attn_ws_cross_bytes", c_int64),
                 ("gemm_ws", c_void_p), ("gemm_ws_bytes", c_int64)])


class DitWeights(Structure):
    """``wan_dit_weights`` of include/wan_hip.h."""
    _fields_ = ([(n, c_int) for n in ("num_layers", "in_dim", "out_dim", "pt", "ph", "pw")] +
                [("blocks", POINTER(BlockWeights))] + [(n, c_void_p) for n in ("pe_w", "pe_b", "head_w", "head_b")])


class DitWorkspace(Structure):
    """``wan_dit_workspace`` of include/wan_hip.h."""
    _fields_ = [("block", BlockWorkspace), ("x", c_void_p), ("tokens", c_void_p), ("head_out", c_void_p)]


# name -> (restype, argtypes); every symbol the header declares
SIGNATURES = {
    "wan_abi_version": (c_int, []),
    "wan_box_probe_scratch_bytes": (c_int64, []),
    "wan_box_probe": (c_int, [POINTER(BoxProbeResult), c_void_p, c_int64, c_int, c_void_p]),
    "wan_last_error": (c_char_p, []),
    "wan_set_tuning": (c_int, [c_char_p, c_int]),
    "wan_get_tuning": (c_int, [c_char_p]),
    "wan_ln_modulate": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int64,
                                c_float, c_void_p]),
    "wan_rmsnorm_rope": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                 c_float, c_void_p, c_void_p, POINTER(RopeParams), c_float, c_void_p]),
    "wan_gemm_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                              c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "wan_rmsnorm_rope_sp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                    c_float, c_void_p, c_void_p, POINTER(RopeParams), c_float, c_void_p, c_void_p, c_int, c_int,
                                    c_void_p]),
    "wan_sp_pack_heads": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_sp_unpack_heads": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_sp_unpack_vt": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_rmsnorm_rope_sp_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                          c_float, c_void_p, c_void_p, POINTER(RopeParams), c_float, c_void_p, c_void_p, c_int, c_int,
                                          c_int, c_void_p]),
    "wan_sp_pack_heads_split": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_sp_unpack_heads_split": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_dit_block_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(BlockWeights), POINTER(BlockWorkspace),
                                      c_void_p, c_void_p, POINTER(RopeParams), c_int, c_int64, c_int64, c_void_p]),
    "wan_dit_block_tail_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(BlockWeights), POINTER(BlockWorkspace),
                                           c_int, c_int64, c_void_p]),
    "wan_dit_forward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_void_p),
                                POINTER(DitWeights), POINTER(DitWorkspace), c_void_p, c_void_p, POINTER(RopeParams),
                                c_int, c_int, c_int, c_int, c_int64, c_int, c_void_p]),
    "wan_dit_block_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int64, c_int64, POINTER(c_int64), POINTER(c_int64)]),
    "wan_gemm_fp8": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                             c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "wan_gemm_fp8_ws": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "wan_gemm_fp8_ws_plan": (c_int, [c_int, c_int, c_int]),
    "wan_gemm_fp8_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "wan_gemm_fp8_pk_segment": (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_int)]),
    "wan_quantize_rows_fp8": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "wan_ln_modulate_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int64,
                                    c_float, c_void_p]),
    "wan_attention_fwd": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                  c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                  c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_int64, c_void_p]),
    "wan_attention_fwd_varlen": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                         c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                         c_int, c_int, c_int, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_int64, c_void_p]),
    "wan_attention_fwd_qk8": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_int64, c_int,
                                      c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                      c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "wan_rmsnorm_rope_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                     c_float, c_void_p, c_void_p, POINTER(RopeParams), c_float, c_float, c_void_p, c_void_p,
                                     c_void_p]),
    "wan_attention_fwd_f8": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_int64, c_int,
                                     c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                     c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "wan_vt_mx_scale_bytes": (c_int64, [c_int, c_int, c_int]),
    "wan_vt_quantize_mx": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "wan_col_mean_workspace_bytes": (c_int64, [c_int, c_int]),
    "wan_col_mean_bf16": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "wan_qk_quantize_fp8": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int64, c_void_p, c_float, c_float, c_void_p, c_void_p,
                                    c_void_p]),
    "wan_attention_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "wan_sp_unique_id": (c_int, [c_void_p]),
    "wan_sp_init": (c_int, [ctypes.POINTER(c_void_p), c_void_p, c_int, c_int]),
    "wan_sp_init_from_comm": (c_int, [ctypes.POINTER(c_void_p), c_void_p, c_int, c_int]),
    "wan_sp_rank": (c_int, [c_void_p]),
    "wan_sp_world_size": (c_int, [c_void_p]),
    "wan_sp_a2a_scatter_heads": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_sp_a2a_gather_heads": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_sp_all_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_sp_wait": (c_int, [c_void_p, c_void_p]),
    "wan_sp_ticket": (c_int64, [c_void_p]),
    "wan_sp_wait_for": (c_int, [c_void_p, c_int64, c_void_p]),
    "wan_sp_destroy": (c_int, [c_void_p]),
    "wan_attention_plan": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int64]),
    "wan_gemm_plan": (c_int, [c_int, c_int, c_int]),
    "wan_gemm_ws_plan": (c_int, [c_int, c_int, c_int]),
    "wan_gemm_ws_splits": (c_int, [c_int, c_int, c_int]),
    "wan_gemm_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "wan_gemm_pk_grid": (c_int, [c_int, c_int]),
    "wan_gemm_pk_segment": (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_int)]),
    "wan_gemm_bf16_ws": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                 c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "wan_transpose_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "wan_patchify": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                             c_int, c_int, c_int, c_void_p]),
    "wan_unpatchify": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_int, c_int, c_void_p]),
    "wan_lincomb": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float,
                            c_int64, c_void_p]),
    "wan_gemm_bf16_batched": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                      c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "wan_embedding_rows": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "wan_rmsnorm_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_void_p]),
    "wan_t5_softmax_bias": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                    c_int, c_void_p]),
    "wan_mul_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "wan_conv_cl": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                            POINTER(ConvParams), c_void_p]),
    "wan_rmsnorm_silu_cl": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "wan_softmax_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_float, c_void_p]),
    "wan_video_to_cl": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "wan_cl_to_video": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int64, c_int, c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load (once) and type the shared library.  Raises RuntimeError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP kernels are not built. Run `make` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`) in the repo root. "
            "There is no CPU/eager fallback for this path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{LIB_PATH} does not export `{name}` (stale build?)") from e
        fn.restype = res
        fn.argtypes = args
    if lib.wan_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libwan_hip ABI {lib.wan_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    """Map a wan_status_t to the exception class the reference would raise:
    shape/argument errors -> ValueError, everything else -> RuntimeError."""
    if status == WAN_OK:
        return
    msg = load().wan_last_error().decode(errors="replace")
    if status == WAN_ERR_INVALID:
        raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg} (status {status})")
