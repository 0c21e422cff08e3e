"""CPU: static scan of the ISA hipcc emits for the hand-scheduled attention loops (the loop bodies are written as program order
+ sched_barriers; what must not creep back in is the compiler shuffling values between the register files or spilling inside
them -- round 3 measured 1.33 instead of 1.37 PFLOP/s from exactly that).  Compiles videocof_amd/csrc/attn_fwd.hip for gfx950
with --save-temps (hipcc cross-compiles without a GPU, ~15 s) and histograms the main KV-tile loop of each 4-wave form."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")


@pytest.fixture(scope="module")
def attn_asm(tmp_path_factory):
    d = tmp_path_factory.mktemp("isa")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-honor-nans", "-fno-slp-vectorize",
           "-I" + os.path.join(ROOT, "include"), "--save-temps", "-c", os.path.join(ROOT, "videocof_amd", "csrc", "attn_fwd.hip"),
           "-o", str(d / "attn.o")]
    subprocess.run(cmd, check=True, cwd=d, capture_output=True)
    s = next(p for p in os.listdir(d) if p.endswith("gfx950.s"))
    return open(d / s).read().split("\n")


def _kernel(text, *subs):
    start = next(i for i, l in enumerate(text) if re.match(r"^[A-Za-z_][\w.$]*:", l) and all(s in l for s in subs))
    end = next(i for i in range(start, len(text)) if ".end_amdhsa_kernel" in text[i])
    return text[start:end]


def _main_loop_histogram(body):
    i = next(i for i, l in enumerate(body) if "Loop Header" in l)
    label = body[i].split(":")[0].strip()
    back = max(j for j, m in enumerate(body) if re.search(r"s_c?branch\S*\s+" + re.escape(label) + r"\s*$", m))
    hist = {}
    for l in body[i:back + 1]:
        m = re.match(r"\s+([a-z][a-z0-9_]+)", l)
        if m:
            hist[m.group(1)] = hist.get(m.group(1), 0) + 1
    return hist


@pytest.mark.parametrize("mangled,what", [("attn_fwd_w4_kernelILi0ELb0ELi0ELb0ELb0E", "max-free"),
                                           ("attn_fwd_w4_kernelILi0ELb0ELi1ELb0ELb0E", "lazy reference in the accumulator"),
                                           ("attn_fwd_w4_kernelILi0ELb0ELi1ELb1ELb0E", "lazy reference, fix-up launch"),
                                           ("attn_fwd_w4_kernelILi0ELb1ELi1ELb0ELb0E", "lazy reference, split-KV tail"),
                                           ("attn_fwd_w4_kernelILi0ELb0ELi2ELb0ELb0E", "lazy reference, packed shift (plain q)")])
def test_attention_w4_main_loop_is_clean(attn_asm, mangled, what):
    body = _kernel(attn_asm, mangled)
    meta = "\n".join(attn_asm)
    priv = re.search(re.escape(mangled) + r"\w*\.private_seg_size, (\d+)", meta)
    assert priv and int(priv.group(1)) == 0, (what, "scratch", priv and priv.group(1))
    h = _main_loop_histogram(body)                      # two KV-tile intervals (the loop is unrolled by two)
    assert h.get("v_mfma_f32_32x32x16_bf16") == 128, (what, h.get("v_mfma_f32_32x32x16_bf16"))
    assert h.get("v_exp_f32_e32") == 128 and h.get("ds_read_b128") == 64 and h.get("buffer_load_dwordx4") == 16
    for bad in ("scratch_load_dword", "scratch_load_dwordx4", "scratch_store_dword", "v_accvgpr_read_b32", "v_accvgpr_write_b32",
                "v_accvgpr_mov_b32", "v_readlane_b32", "v_writelane_b32"):
        assert h.get(bad, 0) == 0, (what, bad, h.get(bad))
    vector = sum(c for k, c in h.items() if k.startswith("v_"))
    packed = h.get("v_pk_fma_f32", 0)
    assert vector - packed <= 480, (what, vector)        # 128 MFMA + 2 x 167 others (+ the check) per two tiles
    assert packed == (64 if "packed" in what else 0)


@pytest.mark.parametrize("mangled", ["attn_fwd_w4_kernelILi1ELb0ELi1ELb0ELb0ELb1E", "attn_fwd_w4_kernelILi1ELb0ELi2ELb0ELb0ELb1E"])
def test_cross_attention_persistent_form_counts_its_stores(attn_asm, mangled):
    """The persistent cross-attention kernel (PERSIST): the next block's K / V requests and Q fragment loads are issued BEFORE the current
    block's output stores, and `s_waitcnt vmcnt(16)` at the top of the next block waits for exactly those -- which is only right if the
    kernel issues EXACTLY 16 vector-memory instructions after them: 16 buffer_store_dwordx4 (range-checked by their descriptor, never
    branched around), no other store form, and nothing compiler-tracked in between.  The Q fragments arrive through untracked asm loads
    straight into AGPRs: no instruction but those loads and the MFMAs may name those registers (a copy would read them too early)."""
    body = _kernel(attn_asm, mangled)
    meta = "\n".join(attn_asm)
    priv = re.search(re.escape(mangled) + r"\w*\.private_seg_size, (\d+)", meta)
    assert priv and int(priv.group(1)) == 0
    code = [l for l in body if re.match(r"\s+[a-z]", l)]
    assert sum("buffer_store_dwordx4" in l for l in code) == 16
    assert not any(re.match(r"\s+(buffer_store_dwordx2|global_store|flat_store)", l) for l in code)
    assert sum("s_waitcnt vmcnt(16)" in l for l in code) == 1
    loads = [l for l in code if re.match(r"\s+global_load_dwordx4 a\[", l)]
    assert len(loads) == 48                               # 16 for the first block + 16 per prefetch site of the next block (ahead of the peeled tile when the tile count is even, after it otherwise)
    regs = set()
    for l in loads:
        m = re.search(r"a\[(\d+):(\d+)\]", l)
        regs |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    assert len(regs) == 64

    def named(l):
        out = set()
        for m in re.finditer(r"a\[(\d+):(\d+)\]", l):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\ba(\d+)\b", l):
            out.add(int(m.group(1)))
        return out
    others = [l.strip() for l in code if (named(l) & regs) and "global_load_dwordx4" not in l and "v_mfma" not in l]
    assert not others, others[:5]
    # the 16 stores form one straight run (no branch, no other vector-memory instruction among them): they are issued as a block, always
    i0 = min(i for i, l in enumerate(code) if "buffer_store_dwordx4" in l)
    i1 = max(i for i, l in enumerate(code) if "buffer_store_dwordx4" in l)
    among = [l.strip() for l in code[i0:i1 + 1] if re.match(r"\s+(buffer_load|global_load|global_store|flat_|buffer_atomic|global_atomic|s_cbranch|s_branch)", l)]
    assert not among, among[:5]


@pytest.mark.parametrize("mangled,what", [("attn_fwd_w4_kernelILi0ELb0ELi1ELb0ELb1E", "fp8 QK^T"),
                                           ("attn_fwd_w4_kernelILi0ELb1ELi1ELb0ELb1E", "fp8 QK^T, split-KV tail")])
def test_attention_qk8_main_loop_is_clean(attn_asm, mangled, what):
    """The fp8 QK^T form: 8 scaled fp8 MFMAs + 32 bf16 MFMAs per tile, K fragments in the fixed registers a[224:255] (read by the
    MFMAs straight from where the two ds_read_b128 of a fragment put them: no copies between the register files)."""
    body = _kernel(attn_asm, mangled)
    meta = "\n".join(attn_asm)
    priv = re.search(re.escape(mangled) + r"\w*\.private_seg_size, (\d+)", meta)
    assert priv and int(priv.group(1)) == 0, (what, "scratch", priv and priv.group(1))
    h = _main_loop_histogram(body)                      # two KV-tile intervals
    assert h.get("v_mfma_scale_f32_32x32x64_f8f6f4") == 16 and h.get("v_mfma_f32_32x32x16_bf16") == 64, h
    assert h.get("v_exp_f32_e32") == 128 and h.get("ds_read_b128") == 48 and h.get("buffer_load_dwordx4") == 12
    for bad in ("scratch_load_dword", "scratch_load_dwordx4", "scratch_store_dword", "v_accvgpr_read_b32", "v_accvgpr_write_b32",
                "v_accvgpr_mov_b32", "v_readlane_b32", "v_writelane_b32"):
        assert h.get(bad, 0) == 0, (what, bad, h.get(bad))
    assert sum(c for k, c in h.items() if k.startswith("v_")) <= 420, what
    loop = [l for l in body if "v_mfma_scale" in l and "a[0x" in l.replace("a[22", "a[0x").replace("a[23", "a[0x").replace("a[24", "a[0x")]
    assert loop, "the loop's fp8 MFMAs read their K fragments from the fixed AGPR range"


def test_attention_f8_main_loop_is_clean(attn_asm):
    """fp8 QK^T + fp8 P.V: 16 scaled fp8 MFMAs per tile, 64 exponentials, 32 P conversions (two values each), 16 fragment reads (K and V^T
    in the fixed registers a[192:255]), no copies between the register files, no scratch."""
    mangled = "attn_fwd_f8_kernel"
    body = _kernel(attn_asm, mangled)
    meta = "\n".join(attn_asm)
    priv = re.search(re.escape(mangled) + r"\w*\.private_seg_size, (\d+)", meta)
    assert priv and int(priv.group(1)) == 0, priv and priv.group(1)
    h = _main_loop_histogram(body)                      # two KV-tile intervals
    assert h.get("v_mfma_scale_f32_32x32x64_f8f6f4") == 32 and h.get("v_mfma_f32_32x32x16_bf16", 0) == 0, h
    assert h.get("v_exp_f32_e32") == 128 and h.get("v_cvt_scalef32_pk_fp8_f32") == 64 and h.get("ds_read_b128") == 32
    assert h.get("v_permlane32_swap_b32") == 8 and h.get("buffer_load_dwordx4") == 8 and h.get("buffer_load_dword") == 2
    for bad in ("scratch_load_dword", "scratch_load_dwordx4", "scratch_store_dword", "v_accvgpr_read_b32", "v_accvgpr_write_b32",
                "v_accvgpr_mov_b32", "v_readlane_b32", "v_writelane_b32"):
        assert h.get(bad, 0) == 0, (bad, h.get(bad))
    assert sum(c for k, c in h.items() if k.startswith("v_")) <= 420


# ------------------------------------------------------------------ the persistent stream-K GEMM (videocof_amd/csrc/gemm_bf16_pk.hip)
@pytest.fixture(scope="module")
def pk_asm(tmp_path_factory):
    d = tmp_path_factory.mktemp("isa_pk")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-honor-nans", "-DWAN_DEV_EXPERIMENTS=0",
           "-I" + os.path.join(ROOT, "include"), "--save-temps", "-c", os.path.join(ROOT, "videocof_amd", "csrc", "gemm_bf16_pk.hip"),
           "-o", str(d / "pk.o")]
    subprocess.run(cmd, check=True, cwd=d, capture_output=True)
    s = next(p for p in os.listdir(d) if p.endswith("gfx950.s"))
    return open(d / s).read().split("\n")


@pytest.mark.parametrize("form", [1, 0])
@pytest.mark.parametrize("epi,what", [(0, "bf16"), (1, "gelu"), (2, "f32"), (3, "resid"), (4, "transposed")])
def test_persistent_gemm_kernels_are_clean(pk_asm, epi, what, form):
    """What must not creep back into gemm_pk_kernel<EPI, FORM>: scratch (the kernel owns all 512 registers of a lane; hipcc spills at
    the slightest excuse, and a spilled address is reloaded behind a vmcnt(0)), a main loop that is not exactly 256 MFMAs / 64
    fragment reads / 32 LDS-DMA requests / 4 barriers per two K tiles, and the one-request-per-lane epilogues: form 1 (product)
    moves every output with 16-byte accesses of whole row segments (bf16: 32 stores per lane and tile, fp32: 64 stores / 64 loads,
    transposed: 32) and needs no lane exchange; form 0 (round 4's epilogues, the developer A/B partner) is held to what it was."""
    mangled = f"gemm_pk_kernelILi{epi}ELi{form}ELb0E"
    body = _kernel(pk_asm, mangled)
    meta = "\n".join(pk_asm)
    priv = re.search(re.escape(mangled) + r"\w*\.private_seg_size, (\d+)", meta)
    assert priv and int(priv.group(1)) == 0, (what, "scratch", priv and priv.group(1))
    assert not any("scratch_" in l for l in body), what
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = None
    for j, l in enumerate(body):                         # the K loop = the SHORTEST backward branch range that holds MFMAs
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)\s*$", l)
        if not m or labels.get(m.group(1), j) >= j:
            continue
        seg = body[labels[m.group(1)]:j + 1]
        n = sum("v_mfma_f32_16x16x32_bf16" in x for x in seg)
        if n and (best is None or len(seg) < len(best[1])):
            best = (n, seg)
    assert best is not None, what
    n, seg = best
    assert n == 256, (what, n)
    count = lambda op: sum(1 for l in seg if re.match(r"\s+" + op + r"\b", l))
    assert count("ds_read_b128") == 64 and count("buffer_load_dwordx4") == 32, (what, count("ds_read_b128"), count("buffer_load_dwordx4"))
    assert count("s_barrier") == 4
    for bad in ("v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_accvgpr_mov_b32", "v_readlane_b32", "v_writelane_b32"):
        assert count(bad) == 0, (what, bad)
    whole = lambda op: sum(1 for l in body if re.match(r"\s+" + op + r"\b", l))
    if form == 0:
        if what in ("f32", "resid"):
            assert whole("global_store_dwordx4") == 0 and whole("global_store_dword") >= 256, (what, whole("global_store_dword"))
            if what == "resid":
                assert whole("global_load_dword") >= 256
        if what in ("bf16", "gelu"):
            assert whole("v_permlane16_swap_b32") >= 64 and whole("global_store_dwordx4") >= 32
        return
    assert whole("v_permlane16_swap_b32") == 0 and whole("ds_bpermute_b32") == 0, what
    if what in ("bf16", "gelu", "transposed"):
        assert whole("global_store_dwordx4") == 32, (what, whole("global_store_dwordx4"))      # one straight-line copy: 8 x 4 stores
    if what in ("f32", "resid"):
        # >= 2 straight-line copies (bias / no bias; resid: x gate x sample seam) of 64 float4 stores each
        assert whole("global_store_dwordx4") >= 120 and whole("global_store_dword") <= 8, (what, whole("global_store_dwordx4"), whole("global_store_dword"))
        if what == "resid":
            assert whole("global_load_dwordx4") >= 64 * 6 and whole("global_load_dword") == 0


@pytest.mark.parametrize("epi,what", [(0, "bf16"), (1, "gelu"), (2, "f32"), (3, "resid"), (4, "transposed")])
def test_persistent_gemm_e4m3_kernels_are_clean(pk_asm, epi, what):
    """The e4m3 instantiation gemm_pk_kernel<EPI, 1, true> ("schedule P"): no scratch, and a main loop of exactly 128 MX-scaled
    16x16x128 MFMAs / 64 fragment reads / 32 LDS-DMA requests / 4 barriers per two K tiles with the counted wait (vmcnt(7)) in front
    of the publishing barrier of each tile and no accumulator <-> VGPR traffic; the epilogues keep the bf16 form's 16-byte accesses."""
    mangled = f"gemm_pk_kernelILi{epi}ELi1ELb1E"
    body = _kernel(pk_asm, mangled)
    meta = "\n".join(pk_asm)
    priv = re.search(re.escape(mangled) + r"\w*\.private_seg_size, (\d+)", meta)
    assert priv and int(priv.group(1)) == 0, (what, "scratch", priv and priv.group(1))
    assert not any("scratch_" in l for l in body), what
    assert not any("v_mfma_f32_16x16x32_bf16" in l for l in body), what
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = None
    for j, l in enumerate(body):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)\s*$", l)
        if not m or labels.get(m.group(1), j) >= j:
            continue
        seg = body[labels[m.group(1)]:j + 1]
        n = sum("v_mfma_scale_f32_16x16x128_f8f6f4" in x for x in seg)
        if n and (best is None or len(seg) < len(best[1])):
            best = (n, seg)
    assert best is not None, what
    n, seg = best
    assert n == 128, (what, n)
    count = lambda op: sum(1 for l in seg if re.match(r"\s+" + op + r"\b", l))
    assert count("ds_read_b128") == 64 and count("buffer_load_dwordx4") == 32, (what, count("ds_read_b128"), count("buffer_load_dwordx4"))
    assert count("s_barrier") == 4
    assert sum(1 for l in seg if re.match(r"\s+s_waitcnt vmcnt\(7\)", l)) == 2, what
    assert not any(re.match(r"\s+s_waitcnt vmcnt\(0\)", l) for l in seg), what
    for bad in ("v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_accvgpr_mov_b32", "v_readlane_b32", "v_writelane_b32", "v_mov_b32_e32"):
        assert count(bad) == 0, (what, bad)
    whole = lambda op: sum(1 for l in body if re.match(r"\s+" + op + r"\b", l))
    if what in ("bf16", "gelu", "transposed"):
        assert whole("global_store_dwordx4") == 32, (what, whole("global_store_dwordx4"))
    if what in ("f32", "resid"):
        assert whole("global_store_dwordx4") >= 120 and whole("global_store_dword") <= 8, (what, whole("global_store_dwordx4"), whole("global_store_dword"))
