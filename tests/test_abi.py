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
    assert lib.wan_abi_version() == _lib.ABI_VERSION == 2


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
