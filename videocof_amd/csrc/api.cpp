// Error plumbing and ABI version of libwan_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include <hip/hip_runtime.h>

#include "common.hpp"

#ifndef WAN_DEV_EXPERIMENTS
#define WAN_DEV_EXPERIMENTS 0
#endif

static thread_local char g_err[512] = "";

void wan_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* wan_last_error(void) { return g_err; }
extern "C" int wan_abi_version(void) { return WAN_ABI_VERSION; }

// Launch planning (tile quantisation) wants the CU count; it is host arithmetic and must also work where no GPU is
// visible (the CPU-side ABI tests), hence the fallback.
int wan_cu_count() {
    static std::atomic<int> ncu{0};
    int v = ncu.load(std::memory_order_relaxed);
    if (v == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        ncu.store(v, std::memory_order_relaxed);
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// Developer switches.  The environment is read ONCE (first use, thread-safe function-local static); after that the
// hot path only loads an atomic int.  wan_set_tuning() overrides a switch at run time (A/B harnesses).
namespace {
struct TuningKey { const char* key; const char* env; int def; };
const TuningKey kTuningKeys[WAN_TUNE_COUNT] = {
    {"attn_tail", "WAN_ATTN_TAIL", 1},          // split-KV tail round of wan_attention_fwd
    {"attn_fast", "WAN_ATTN_FAST", 1},          // max-free first attempt (+ checked lazy-reference fix-up): 1 = long self-attention launches with scratch, 2 = whenever there is scratch, 0 = never
    {"attn_xcd_map", "WAN_ATTN_XCD_MAP", 1},    // heads pinned to XCDs (one head's K/V per XCD L2 at a time)
    {"gemm_gm", "WAN_GEMM_GM", 0},              // M tiles per rasterisation group of the 256^2 GEMM (0 = by shape)
    {"gemm_phases", "WAN_GEMM_PHASES", 0},      // K-loop phasing of the 256^2 GEMM (0 = default)
    {"debug_checks", "WAN_DEBUG_CHECKS", 0},    // synchronising contract checks (V^T padding finite, ...)
    {"gemm_variant", "WAN_GEMM_VARIANT", 0},    // 1 = force the 128^2 GEMM, 2 = force the 256^2 GEMM, 0 = by shape
    {"conv_xcd", "WAN_CONV_XCD", 1},            // XCD slab rasterisation of wan_conv_cl
    {"gemm_w4", "WAN_GEMM_W4", 1},              // 256^2 GEMM on the 4-wave kernel: 0 never, 1 K >= 4096, 2 K >= 8192, 3 whenever K % 128 == 0
    {"conv_fast", "WAN_CONV_FAST", 1},          // wan_conv_cl gather addresses on the branch-free 24-bit multiply path (0 = general 64-bit path)
    {"conv_patch", "WAN_CONV_PATCH", 1},        // causal 3x3x3 stride-1 convs with Cout % 96 == 0 on the LDS-patch kernel (0 = the gather kernel)
    {"attn_ref", "WAN_ATTN_REF", 1},            // lazy softmax reference of the 4-wave kernel: 1 = -m splat in the accumulator, 2 = packed subtract
    {"conv_head", "WAN_CONV_HEAD", 1},          // causal 3x3x3 convs with <= 4 output channels on the direct (vector-ALU) kernel (0 = the gather kernel)
    {"gemm_exp", "WAN_GEMM_EXP", 0},            // TIMING-ONLY experiment of the 4-wave GEMM: bit 0 / 1 = skip the W / A tile DMA of the main loop (results are garbage)
    {"gemm_ring", "WAN_GEMM_RING", 0},          // `make EXPERIMENTS=1` builds only: 4-wave GEMM over a four-stage ring of 32-k tiles (measured 5-9 % slower than two 64-k stages)
    {"gemm_pk", "WAN_GEMM_PK", 1},              // persistent stream-K form of the 4-wave 256^2 GEMM for callers that bring a workspace: 0 never, 1 where the 4-wave kernel would run, 2 whenever K % 128 == 0
    {"gemm_pk_workers", "WAN_GEMM_PK_WORKERS", 0},      // its grid (0 = one workgroup per CU); developer A/B
    {"gemm_pk_min_units", "WAN_GEMM_PK_MIN_UNITS", 0},  // smallest stream-K range in units of two K tiles (0 = a quarter of the tile's K range)
    {"gemm_pk_order", "WAN_GEMM_PK_ORDER", 0},          // 1 = whole tiles in lockstep order instead of by per-XCD ticket (developer A/B)
    {"gemm_pk_form", "WAN_GEMM_PK_FORM", 29},           // epilogues of the persistent GEMM, one bit per WAN_EPI_* value: 1 = from row-permuted operand tiles (a lane's accumulators contiguous in the output), 0 = the round-4 form (developer A/B)
    {"row_group", "WAN_ROW_GROUP", 2},                  // LN-modulate / RMSNorm+RoPE: token rows per workgroup (2 or 4: per-column parameters fetched once per group; 1 = the one-row kernels)
    {"sp_inline", "WAN_SP_INLINE", 0},                  // library communicator: 1 = every collective on the caller's stream itself (no side stream); 0 = only while that stream is being captured
    {"gemm_splitk", "WAN_GEMM_SPLITK", 1},              // split-K form of the 128^2 GEMM for small shapes that bring a workspace: 1 = by shape, 0 = never, 2..8 = force that many pieces (developer A/B)
    {"conv_mfma", "WAN_CONV_MFMA", 0},                  // matrix instruction of the VAE's LDS-patch convolution: 0 = by the per-frame plane (16x16x32 when four frames of it fill the chip; never by frame count -- chunked decodes stay bit-identical), 32 = v_mfma_f32_32x32x16_bf16, 16 = v_mfma_f32_16x16x32_bf16
    {"attn_persist", "WAN_ATTN_PERSIST", 1},            // short-KV (cross-attention) launches on the persistent form of the 4-wave kernel: one resident workgroup per CU walks the query blocks (0 = one workgroup per block, developer A/B)
};
struct Tuning {
    std::atomic<int> v[WAN_TUNE_COUNT];
    Tuning() {
        for (int i = 0; i < WAN_TUNE_COUNT; ++i) {
            const char* e = getenv(kTuningKeys[i].env);
            v[i].store(e ? atoi(e) : kTuningKeys[i].def, std::memory_order_relaxed);
        }
    }
};
Tuning& tuning() {
    static Tuning t;
    return t;
}
}  // namespace

int wan_tune(int which) { return tuning().v[which].load(std::memory_order_relaxed); }

static std::atomic<int> g_last_attn_variant{0};
void wan_note_attn_variant(int variant) { g_last_attn_variant.store(variant, std::memory_order_relaxed); }

extern "C" wan_status_t wan_set_tuning(const char* key, int value) {
    WAN_REQUIRE(key != nullptr, WAN_ERR_INVALID, "wan_set_tuning: null key");
    for (int i = 0; i < WAN_TUNE_COUNT; ++i)
        if (!strcmp(key, kTuningKeys[i].key)) {
            tuning().v[i].store(value, std::memory_order_relaxed);
            return WAN_OK;
        }
    wan_set_error("wan_set_tuning: unknown key '%s'", key);
    return WAN_ERR_INVALID;
}

extern "C" int wan_get_tuning(const char* key) {
    if (key && !strcmp(key, "last_attn_variant")) return g_last_attn_variant.load(std::memory_order_relaxed);
    if (key && !strcmp(key, "dev_experiments")) return WAN_DEV_EXPERIMENTS;      // 1: built with `make EXPERIMENTS=1` (gemm_exp variants compiled in)
    if (key)
        for (int i = 0; i < WAN_TUNE_COUNT; ++i)
            if (!strcmp(key, kTuningKeys[i].key)) return wan_tune(i);
    return -1;
}

// One-time, per-device kernel attribute set-up (hipFuncSetAttribute is per device).  `done` is a bit mask of devices.
wan_status_t wan_once_per_device(std::atomic<uint64_t>& done, wan_status_t (*init)()) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    const uint64_t bit = 1ull << dev;
    if (done.load(std::memory_order_acquire) & bit) return WAN_OK;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (done.load(std::memory_order_relaxed) & bit) return WAN_OK;
    const wan_status_t st = init();
    if (st == WAN_OK) done.fetch_or(bit, std::memory_order_release);
    return st;
}
