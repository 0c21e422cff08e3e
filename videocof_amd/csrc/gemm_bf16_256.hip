// 256 x 256 x 64 "phased" bf16 GEMM for the large nn.Linear shapes of the DiT (M ~ 6.7e4).
//
// Same math and epilogues as gemm_bf16.hip (acc[m,n] = sum_k A[m,k] W[n,k]); different schedule.
// One workgroup = 8 waves = 2 (M) x 4 (N); a wave owns 128 x 64 outputs (8 x 4 MFMA 16x16x32 tiles,
// 128 accumulator VGPRs).  The K loop is cut into 4 PHASES per 64-deep K tile; a phase is
//
//     L: issue the ds_read_b128 of the fragments the phase needs (+ LDS-DMA for the next K tile)
//     s_barrier;  s_waitcnt lgkmcnt(0)
//     M: 16 back-to-back MFMAs (one 64 x 32 quadrant of the wave tile, K = 64) at raised priority
//     s_barrier
//
// and the two wave groups (rows 0-127 / 128-255) run ONE BARRIER APART: while the waves of one
// group are in their matrix segment M, their SIMD partners of the other group are in the load
// segment L.  With a single workgroup per CU the matrix pipe therefore always has one wave per SIMD
// feeding it and LDS latency, DMA issue and barrier skew hide under the partner's MFMAs -- the role
// alternation the CDNA4 guide describes for its 8-phase template, obtained here from program order
// and a one-barrier stagger.
//
// LDS: 2 buffers x (A 256x64 + W 256x64) bf16 = 128 KiB, rows of 128 B with the same source-side
// XOR swizzle as the 128^2 kernel.  K tile t+1 streams into the other buffer during phases 2
// and 3 of tile t (regions whose last reader finished two barriers earlier) and is waited for before
// the barrier that closes tile t for BOTH groups (the late group waits one segment earlier).
//
// Measured and not kept: de-synchronising the CUs with a per-workgroup start delay in the first round (so that the
// HBM-bound fp32 read-modify-write epilogues of different CUs do not coincide): 1282 vs 1285-1304 TFLOP/s on the
// o-projection, 1165 vs 1165-1170 on ffn.2 -- noise; the CUs drift apart on their own.  The same wave tile computed
// with v_mfma_f32_32x32x16_bf16 (4 x 2 tiles, half the MFMA instructions, same LDS traffic): 5-8 % SLOWER on every
// shape (profiles/r01/gemm_mfma32x32_ab.log).
#include <stdlib.h>

#include "common.hpp"

namespace {

constexpr int kDefaultPhases = 2;
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kThreads = 512;
constexpr int kHalfBytes = 128 * BK * 2;       // 16 KiB: 128 rows x 64 k
constexpr int kOperandBytes = 2 * kHalfBytes;  // 32 KiB
constexpr int kBufBytes = 2 * kOperandBytes;   // A + W: 64 KiB
constexpr int kLdsBytes = 2 * kBufBytes;       // 128 KiB

struct GemmArgs {
    const bf16_t* A; int64_t lda;
    const bf16_t* W; int64_t ldw;
    const float* bias;
    void* out; int64_t ldo;
    const float* gate; int64_t rows_per_batch;
    int M, N, K;
    int tiles_m, tiles_n;
    int gm;               // M tiles per rasterisation group (see wan_gemm_bf16_256)
};

__device__ __forceinline__ void tile_coords(const GemmArgs& g, int& tm, int& tn) {
    const int nwg = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int GM = g.gm;                     // GM M tiles x all N tiles per group: the 32 CUs of an XCD share GM A panels
    const int per_group = GM * g.tiles_n;
    const int grp = t / per_group;
    const int first_m = grp * GM;
    const int gm = min(GM, g.tiles_m - first_m);
    const int in = t - grp * per_group;
    tm = first_m + in % gm;
    tn = in / gm;
}

#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define RAW_BARRIER() __builtin_amdgcn_s_barrier()
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int EPI, int PHASES>
__global__ __launch_bounds__(kThreads, 2) void gemm256_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool kTransposed = (EPI == WAN_EPI_BF16_T);

    int tm, tn;
    tile_coords(g, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;       // wave row group (M half), wave column (64 N columns)

    // ---- LDS-DMA sources: per operand half-tile (128 rows) wave w copies pieces 2w, 2w+1 (8 rows each)
    const int srow = lane >> 3, spc = lane & 7;
    const bf16_t* a_src[2][2];
    const bf16_t* w_src[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = h * 128 + (wid * 2 + j) * 8 + srow;     // row inside the 256-row tile
            const int c = spc ^ ((row >> 1) & 7);
            a_src[h][j] = g.A + (int64_t)min(m0 + row, g.M - 1) * g.lda + c * 8;
            w_src[h][j] = g.W + (int64_t)min(n0 + row, g.N - 1) * g.ldw + c * 8;
        }
    auto stage_a = [&](int buf, int koff) {
        char* base = smem + buf * kBufBytes;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(a_src[h][j] + koff, base + h * kHalfBytes + (wid * 2 + j) * 1024);
    };
    auto stage_w = [&](int buf, int koff) {
        char* base = smem + buf * kBufBytes + kOperandBytes;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(w_src[h][j] + koff, base + h * kHalfBytes + (wid * 2 + j) * 1024);
    };

    // ---- fragment read offsets: row = base + 16*i + (l&15), logical chunk = 4*kk + (l>>4)
    const int frow = lane & 15, kg = lane >> 4, swz = (lane >> 1) & 7;
    const int off_kk[2] = {frow * 128 + ((kg ^ swz) << 4), frow * 128 + (((kg + 4) ^ swz) << 4)};
    const int a_base = wr * kHalfBytes;                       // this wave's 128 A rows
    const int w_base = kOperandBytes + wc * 64 * 128;         // this wave's 64 W rows

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][2];                        // current M-half: 4 m-tiles x 2 kk
    bf16x8 wf[PHASES == 4 ? 2 : 4][2];      // 4-phase: current N-quadrant (2 n-tiles); 2-phase: all 4 n-tiles

    auto load_a = [&](const char* sb, int mi) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                af[i][kk] = *reinterpret_cast<const bf16x8*>(sb + a_base + (mi * 4 + i) * 2048 + off_kk[kk]);
    };
    auto load_w = [&](const char* sb, int ni) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                wf[j][kk] = *reinterpret_cast<const bf16x8*>(sb + w_base + (ni * 2 + j) * 2048 + off_kk[kk]);
    };
    auto mma = [&](int mi, int ni) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (kTransposed)
                        acc[mi * 4 + i][ni * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            af[i][kk], wf[j][kk], acc[mi * 4 + i][ni * 2 + j], 0, 0, 0);
                    else
                        acc[mi * 4 + i][ni * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            wf[j][kk], af[i][kk], acc[mi * 4 + i][ni * 2 + j], 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
    };

    auto load_w_all = [&](const char* sb) {
        if constexpr (PHASES == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    wf[j][kk] = *reinterpret_cast<const bf16x8*>(sb + w_base + j * 2048 + off_kk[kk]);
        }
    };
    auto mma_half = [&](int mi) {           // 32 MFMAs: 4 m-tiles x 4 n-tiles x 2 kk
        if constexpr (PHASES == 2) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (kTransposed)
                            acc[mi * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][kk], wf[j][kk], acc[mi * 4 + i][j], 0, 0, 0);
                        else
                            acc[mi * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][kk], af[i][kk], acc[mi * 4 + i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_s_setprio(0);
        }
    };

    const int nk = g.K / BK;
    stage_a(0, 0);
    stage_w(0, 0);
    WAIT_VM0();
    RAW_BARRIER();
    if (wr == 1) RAW_BARRIER();          // stagger: the second wave group runs one barrier late

    if constexpr (PHASES == 2) {
        // Two 32-MFMA phases per K tile (one per 64-row half of the wave tile): half the barriers per MFMA.
        // Next tile: W streams in from L1 (its region was last read four intervals earlier); A is issued in
        // the interval after the late group's last A read has been waited for (early group: top of M1,
        // late group: L1), and every wave drains before the barrier that closes the tile.
        const bool late = wr == 1;
        for (int kt = 0; kt < nk; ++kt) {
            const char* sb = smem + (kt & 1) * kBufBytes;
            const int nxt = (kt & 1) ^ 1;
            const bool more = kt + 1 < nk;
            const int koff = (kt + 1) * BK;
            load_w_all(sb);
            load_a(sb, 0);
            if (more) { stage_w(nxt, koff); if (late) stage_a(nxt, koff); }
            SCHED_FENCE(); RAW_BARRIER(); WAIT_LGKM0(); SCHED_FENCE();
            if (more && !late) stage_a(nxt, koff);
            mma_half(0);
            SCHED_FENCE(); RAW_BARRIER(); SCHED_FENCE();
            load_a(sb, 1);
            if (late) WAIT_VM0();
            SCHED_FENCE(); RAW_BARRIER(); WAIT_LGKM0(); SCHED_FENCE();
            mma_half(1);
            if (!late) WAIT_VM0();
            SCHED_FENCE(); RAW_BARRIER(); SCHED_FENCE();
        }
    } else
    for (int kt = 0; kt < nk; ++kt) {
        const char* sb = smem + (kt & 1) * kBufBytes;
        const int nxt = (kt & 1) ^ 1;
        const bool more = kt + 1 < nk;
        const int koff = (kt + 1) * BK;
        // ---- phase 1: quadrant (0,0)
        load_w(sb, 0);
        load_a(sb, 0);
        SCHED_FENCE(); RAW_BARRIER(); WAIT_LGKM0(); SCHED_FENCE();
        mma(0, 0);
        SCHED_FENCE(); RAW_BARRIER(); SCHED_FENCE();
        // ---- phase 2: quadrant (0,1); next tile's A streams in
        load_w(sb, 1);
        if (more) stage_a(nxt, koff);
        SCHED_FENCE(); RAW_BARRIER(); WAIT_LGKM0(); SCHED_FENCE();
        mma(0, 1);
        SCHED_FENCE(); RAW_BARRIER(); SCHED_FENCE();
        // ---- phase 3: quadrant (1,1); next tile's W streams in
        load_a(sb, 1);
        if (more) stage_w(nxt, koff);
        SCHED_FENCE(); RAW_BARRIER(); WAIT_LGKM0(); SCHED_FENCE();
        mma(1, 1);
        SCHED_FENCE(); RAW_BARRIER(); SCHED_FENCE();
        // ---- phase 4: quadrant (1,0); the next tile must have landed when this K tile closes
        load_w(sb, 0);
        if (wr == 1) WAIT_VM0();         // late group: its load segment is the last interval of the tile
        SCHED_FENCE(); RAW_BARRIER(); WAIT_LGKM0(); SCHED_FENCE();
        mma(1, 0);
        if (wr == 0) WAIT_VM0();
        SCHED_FENCE(); RAW_BARRIER(); SCHED_FENCE();
    }
    if (wr == 0) RAW_BARRIER();          // balance the stagger

    // ---- epilogue (same register -> element maps as gemm_bf16.hip)
    const int l15 = lane & 15, l4 = (lane >> 4) * 4;
    if constexpr (!kTransposed) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + wr * 128 + i * 16 + l15;
            if (m >= g.M) continue;
            const int64_t b = g.gate ? (int64_t)m / g.rows_per_batch : 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wc * 64 + j * 16 + l4;
                if (n >= g.N) continue;
                f32x4 v = acc[i][j];
                if (g.bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(g.bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if constexpr (EPI == WAN_EPI_GELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f32(v[r]);
                }
                if constexpr (EPI == WAN_EPI_BF16 || EPI == WAN_EPI_GELU_BF16) {
                    u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *reinterpret_cast<u32x2*>((bf16_t*)g.out + (int64_t)m * g.ldo + n) = o;
                } else if constexpr (EPI == WAN_EPI_F32) {
                    *reinterpret_cast<float4*>((float*)g.out + (int64_t)m * g.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    float4* p = reinterpret_cast<float4*>((float*)g.out + (int64_t)m * g.ldo + n);
                    float4 x = *p;
                    if (g.gate) {
                        const float4 gv = *reinterpret_cast<const float4*>(g.gate + b * g.N + n);
                        x.x += v[0] * gv.x; x.y += v[1] * gv.y; x.z += v[2] * gv.z; x.w += v[3] * gv.w;
                    } else {
                        x.x += v[0]; x.y += v[1]; x.z += v[2]; x.w += v[3];
                    }
                    *p = x;
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + l15;
            if (n >= g.N) continue;
            const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + wr * 128 + i * 16 + l4;
                if (m >= g.M) continue;
                f32x4 v = acc[i][j];
                bf16_t* p = (bf16_t*)g.out + (int64_t)n * g.ldo + m;
                if (m + 3 < g.M) {
                    u32x2 o = {pack_bf16x2(v[0] + bv, v[1] + bv), pack_bf16x2(v[2] + bv, v[3] + bv)};
                    *reinterpret_cast<u32x2*>(p) = o;
                } else {
                    for (int r = 0; r < 4 && m + r < g.M; ++r) p[r] = (bf16_t)(v[r] + bv);
                }
            }
        }
    }
}

template <int EPI, int PHASES>
wan_status_t launch256(const GemmArgs& g, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};
    const wan_status_t st = wan_once_per_device(attr_done, +[]() -> wan_status_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<EPI, PHASES>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) {
            wan_set_error("wan_gemm_bf16(256): cannot reserve %d B of LDS: %s", kLdsBytes, hipGetErrorString(e));
            return WAN_ERR_LAUNCH;
        }
        return WAN_OK;
    });
    if (st != WAN_OK) return st;
    hipLaunchKernelGGL((gemm256_kernel<EPI, PHASES>), dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(kThreads), kLdsBytes, s, g);
    WAN_CHECK_LAUNCH("wan_gemm_bf16(256)");
    return WAN_OK;
}

}  // namespace

// called by wan_gemm_bf16 (gemm_bf16.hip) for large shapes; arguments already validated there
wan_status_t wan_gemm_bf16_256(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                               void* out, int64_t ldo, int M, int N, int K, int epilogue,
                               const float* gate, int64_t rows_per_batch, hipStream_t s) {
    GemmArgs g;
    g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.bias = bias;
    g.out = out; g.ldo = ldo; g.gate = gate; g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    g.M = M; g.N = N; g.K = K;
    g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
    // M tiles per rasterisation group, measured at M = 67 080 (profiles/r01/gemm_raster_group_ab.log): 2 for wide N
    // (qk projection +5 %, ffn.0 +2 % over the former 4), 3 otherwise (+2 %); 8 and 16 lose 10-15 %.
    g.gm = g.tiles_n >= 40 ? 2 : 3;
    if (const int gm = wan_tune(WAN_TUNE_GEMM_GM); gm > 0) g.gm = gm;       // developer A/B switches (wan_set_tuning)
    const int phases = wan_tune(WAN_TUNE_GEMM_PHASES) > 0 ? wan_tune(WAN_TUNE_GEMM_PHASES) : kDefaultPhases;
#define WAN_G256(E) (phases == 4 ? launch256<E, 4>(g, s) : launch256<E, 2>(g, s))
    switch (epilogue) {
        case WAN_EPI_BF16: return WAN_G256(WAN_EPI_BF16);
        case WAN_EPI_GELU_BF16: return WAN_G256(WAN_EPI_GELU_BF16);
        case WAN_EPI_F32: return WAN_G256(WAN_EPI_F32);
        case WAN_EPI_RESID_F32: return WAN_G256(WAN_EPI_RESID_F32);
        case WAN_EPI_BF16_T: return WAN_G256(WAN_EPI_BF16_T);
        default: wan_set_error("wan_gemm_bf16: unknown epilogue %d", epilogue); return WAN_ERR_INVALID;
    }
#undef WAN_G256
}
