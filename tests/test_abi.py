"""The C-ABI library loads on a CPU-only box and exports every symbol include/wan_hip.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

from videocof_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "wan_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wan_[a-z0-9_]+)\s*\(", src)))


def test_library_present_and_loads():
    assert os.path.isfile(_lib.LIB_PATH), "run `make` / __graft_entry__.build() first"
    lib = _lib.load()
    assert lib.wan_abi_version() == _lib.ABI_VERSION == 11


def test_every_header_symbol_is_exported_and_bound():
    syms = header_symbols()
    assert len(syms) >= 9
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in wan_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in videocof_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == syms


def test_rope_params_struct_layout():
    # must match `wan_rope_params` (6 ints, pad, 2 int64, int, pad)
    assert ctypes.sizeof(_lib.RopeParams) == 48
    assert _lib.RopeParams.token_offset.offset == 24
    assert _lib.RopeParams.rows_per_batch.offset == 32
    assert _lib.RopeParams.max_pos.offset == 40


def test_argument_errors_map_to_python_exceptions():
    lib = _lib.load()
    # null pointers -> WAN_ERR_INVALID -> ValueError; no kernel is launched
    st = lib.wan_ln_modulate(None, None, None, 1, None, 4, 256, 4, 1e-6, None)
    assert st == _lib.WAN_ERR_INVALID
    with pytest.raises(ValueError, match="null tensor"):
        _lib.check(st, "wan_ln_modulate")
    st = lib.wan_attention_fwd(1, 128, 0, 1, 128, 0, 1, 64, 0, 1, 128, 0, 1, 8, 8, 1, 64, 0.1, 0, None, 0, None)
    assert st == _lib.WAN_ERR_UNSUPPORTED      # head_dim 64 is not built
    with pytest.raises(RuntimeError, match="head_dim"):
        _lib.check(st, "wan_attention_fwd")
    st = lib.wan_gemm_bf16(1, 100, 1, 100, None, 1, 128, 4, 128, 100, 0, None, 0, None)
    assert st == _lib.WAN_ERR_UNSUPPORTED      # K not a multiple of 64
    # the composites validate before they enqueue anything
    st = lib.wan_dit_block_forward(None, None, None, None, None, None, None, None, None, 1, 420, 420, None)
    assert st == _lib.WAN_ERR_INVALID
    st = lib.wan_dit_forward(None, 0, None, 0, None, None, None, None, None, None, None, None, None, 1, 7, 12, 20, 420, 0, None)
    assert st == _lib.WAN_ERR_INVALID
    with pytest.raises(ValueError, match="null argument"):
        _lib.check(st, "wan_dit_forward")
    # a latent that is not a multiple of the patch (weights / workspace only need to be non-null for this check to be reached)
    import ctypes
    w, ws, rp = _lib.DitWeights(), _lib.DitWorkspace(), _lib.RopeParams()
    bw = (_lib.BlockWeights * 1)()
    w.num_layers, w.blocks, w.pt, w.ph, w.pw = 1, bw, 1, 2, 2
    w.pe_w = w.pe_b = w.head_w = w.head_b = 8
    ws.x = ws.tokens = ws.head_out = 8
    ptrs = (ctypes.c_void_p * 1)(8)
    st = lib.wan_dit_forward(8, 0, 8, 0, 8, 8, ptrs, ptrs, ctypes.byref(w), ctypes.byref(ws), 8, 8, ctypes.byref(rp), 1, 7, 13, 20, 420, 0, None)
    assert st == _lib.WAN_ERR_INVALID
    with pytest.raises(ValueError, match="not a multiple of the patch"):
        _lib.check(st, "wan_dit_forward")


def test_attention_tail_plan_is_pure_host_logic():
    """wan_attention_workspace_bytes is host arithmetic (CU count falls back to 256 without a GPU): the shapes
    DESIGN.md quotes.  bytes = flags (16-byte header + one int per workgroup of the un-split grid, rounded up to 256 B)
    + batch * nsplit * heads * rows_tail * (128 + 2) * 4 for the split tail round."""
    lib = _lib.load()
    f = lib.wan_attention_workspace_bytes
    L = 67080                                   # 263 query blocks of 256; the last one holds 8 rows

    def flags(batch, lq, heads):
        return (16 + (lq + 255) // 256 * heads * batch * 4 + 255) // 256 * 256

    assert f(1, L, L, 5, 128) == flags(1, L, 5) + 1 * 7 * 5 * (L - 256 * 256) * 130 * 4      # 8-way Ulysses shard: 35 tail blocks, 7 splits
    assert f(1, L, L, 10, 128) == flags(1, L, 10) + 1 * 3 * 10 * (L - 256 * 256) * 130 * 4   # 4-way shard: 70 tail blocks, 3 splits
    assert f(1, L, L, 20, 128) == flags(1, L, 20)                                            # remainder 140 > 128 CUs: no split
    assert f(1, L, L, 40, 128) == flags(1, L, 40) + 1 * 6 * 40 * 8 * 130 * 4                 # single GPU: the 8-row last block, 6 splits
    assert f(1, 4096, L, 5, 128) == flags(1, 4096, 5) == 512                                 # fits in one round
    assert f(1, L, 512, 40, 128) == flags(1, L, 40)                                          # cross-attention: short key range
    assert f(1, L, L, 5, 64) == 0 and f(0, L, L, 5, 128) == 0                                # unsupported / empty


def test_tuning_switches_are_read_once_and_settable():
    """Developer switches: environment consulted once at first use, then wan_set_tuning / wan_get_tuning only
    (the launch paths never call getenv)."""
    lib = _lib.load()
    for key, default in (("attn_tail", 1), ("attn_fast", 1), ("attn_xcd_map", 1), ("gemm_gm", 0), ("gemm_phases", 0),
                         ("gemm_variant", 0), ("conv_xcd", 1), ("debug_checks", 0), ("attn_persist", 1), ("attn_ref", 1), ("gemm_w4", 1), ("conv_fast", 1), ("conv_patch", 1), ("conv_head", 1), ("gemm_exp", 0)):
        assert lib.wan_get_tuning(key.encode()) == default, key
    assert lib.wan_set_tuning(b"attn_tail", 0) == _lib.WAN_OK and lib.wan_get_tuning(b"attn_tail") == 0
    os.environ["WAN_ATTN_TAIL"] = "7"                    # too late: not consulted again
    assert lib.wan_get_tuning(b"attn_tail") == 0
    del os.environ["WAN_ATTN_TAIL"]
    assert lib.wan_set_tuning(b"attn_tail", 1) == _lib.WAN_OK
    assert lib.wan_get_tuning(b"no_such_key") == -1
    st = lib.wan_set_tuning(b"no_such_key", 1)
    assert st == _lib.WAN_ERR_INVALID
    with pytest.raises(ValueError, match="unknown key"):
        _lib.check(st, "wan_set_tuning")
    # the tail plan follows the switch (host arithmetic, no GPU needed)
    L = 67080
    with_tail = lib.wan_attention_workspace_bytes(1, L, L, 5, 128)
    lib.wan_set_tuning(b"attn_tail", 0)
    try:
        assert lib.wan_attention_workspace_bytes(1, L, L, 5, 128) < with_tail
    finally:
        lib.wan_set_tuning(b"attn_tail", 1)
