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

#include <type_traits>

#include "common.hpp"

#ifndef WAN_DEV_EXPERIMENTS
#define WAN_DEV_EXPERIMENTS 0
#endif

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
    // FP8 instantiation (wan_gemm_fp8): A / W point at e4m3 bytes, lda / ldw count bytes = elements; the product of the
    // quantised operands is scaled by sa[m] * sw[n] (per-token, per-output-channel) before bias and epilogue
    const float* sa; const float* sw;
    int exp;              // developer experiment (gemm_exp), TIMING ONLY: bit 0 / bit 1 = the 4-wave kernel's main loop skips its W / A tile DMA, bit 2 = every DMA reads K tile 0 / 1 (cache hits), bit 3 (with bit 0) = the W bytes are fetched by plain register loads instead
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

// FP8 = true: the SAME kernel on e4m3 operands.  A K tile is then 128 elements -- still 128 bytes per LDS row, so staging,
// swizzle, fragment addresses and barriers are unchanged -- and the two 16x16x32 bf16 MFMAs per (m, n) tile and K tile become
// ONE v_mfma_scale_f32_16x16x128_f8f6f4 (MX-scaled form, the only fp8 MFMA above the bf16 rate on gfx950) with every block
// scale = 2^0: 32 cycles for 4x the K of a 16-cycle bf16 MFMA, i.e. the same matrix-pipe cycles, LDS bytes and DMA count per
// K tile for twice the FLOPs.  A lane's 32-byte operand = the two 16-byte chunks 2 kg, 2 kg + 1 of its row; A and W use the
// same lane -> k assignment, which is all the dot product needs.
template <int EPI, int PHASES, bool FP8 = false>
__global__ __launch_bounds__(kThreads, 2) void gemm256_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool kTransposed = (EPI == WAN_EPI_BF16_T);
    constexpr int kEl = FP8 ? 1 : 2;                 // bytes per operand element
    constexpr int kTileK = FP8 ? 2 * BK : BK;        // elements per K tile (128 bytes either way)

    int tm, tn;
    tile_coords(g, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;       // wave row group (M half), wave column (64 N columns)

    // ---- LDS-DMA sources: per operand half-tile (128 rows) wave w copies pieces 2w, 2w+1 (8 rows each)
    const int srow = lane >> 3, spc = lane & 7;
    const char* a_src[2][2];
    const char* w_src[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = h * 128 + (wid * 2 + j) * 8 + srow;     // row inside the 256-row tile
            const int c = spc ^ ((row >> 1) & 7);
            a_src[h][j] = (const char*)g.A + ((int64_t)min(m0 + row, g.M - 1) * g.lda) * kEl + c * 16;
            w_src[h][j] = (const char*)g.W + ((int64_t)min(n0 + row, g.N - 1) * g.ldw) * kEl + c * 16;
        }
    auto stage_a = [&](int buf, int koff) {           // koff: byte offset of the K tile inside a row
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
    // bf16: the two k-steps of a tile read chunks kg and kg + 4; fp8: the one MFMA reads the adjacent chunks 2 kg, 2 kg + 1
    const int off_kk[2] = {frow * 128 + (((FP8 ? 2 * kg : kg) ^ swz) << 4), frow * 128 + (((FP8 ? 2 * kg + 1 : kg + 4) ^ swz) << 4)};
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
    // one (m, n) 16 x 16 tile of the K tile: two bf16 MFMAs (k-steps kk = 0, 1) or one MX-scaled fp8 MFMA over both fragments
    auto mma_tile = [&](f32x4& c, const bf16x8 (&a2)[2], const bf16x8 (&w2)[2]) {
        if constexpr (FP8) {
            typedef int i32x8 __attribute__((ext_vector_type(8)));
            const u32x4 a0 = __builtin_bit_cast(u32x4, a2[0]), a1 = __builtin_bit_cast(u32x4, a2[1]);
            const u32x4 w0 = __builtin_bit_cast(u32x4, w2[0]), w1 = __builtin_bit_cast(u32x4, w2[1]);
            const i32x8 av = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
            const i32x8 wv = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
            // formats 0, 0 = e4m3 x e4m3; block scales: byte 0 of 0x7f7f7f7f = E8M0 127 = 2^0 for every 32-element block
            if constexpr (kTransposed) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, wv, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, av, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if constexpr (kTransposed) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[kk], w2[kk], c, 0, 0, 0);
                else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[kk], a2[kk], c, 0, 0, 0);
            }
        }
    };
    auto mma = [&](int mi, int ni) {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (FP8) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma_tile(acc[mi * 4 + i][ni * 2 + j], af[i], wf[j]);
        } else {
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
        if constexpr (PHASES == 2 && FP8) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma_tile(acc[mi * 4 + i][j], af[i], wf[j]);
            __builtin_amdgcn_s_setprio(0);
        } else if constexpr (PHASES == 2) {
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

    const int nk = g.K / kTileK;
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
            const int koff = (kt + 1) * 128;           // bytes: one K tile is 128 bytes of a row in either element type
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
        const int koff = (kt + 1) * 128;
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
        // Per output column group j the bias (and fp8 weight scale) is loaded ONCE; the fp32 read-modify-write epilogue reads
        // its 8 float4 of the residual stream (and the gate rows) for two row groups back to back and waits once -- written
        // element by element the compiler emitted load / wait / store 32 times in sequence (32 memory round trips per tile).
        // Out-of-range rows / columns read a clamped (valid) address and are not stored.
        int nn[4];
        bool nok[4];
        float4 bj[4], snj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + l4;
            nok[j] = n < g.N;
            nn[j] = nok[j] ? n : 0;
            bj[j] = g.bias ? *reinterpret_cast<const float4*>(g.bias + nn[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (FP8) snj[j] = *reinterpret_cast<const float4*>(g.sw + nn[j]);
        }
        const int rpb = g.gate ? (int)g.rows_per_batch : 1;
#pragma unroll
        for (int ig = 0; ig < 8; ig += 2) {
            int mm[2];
            bool mok[2];
            float sm[2];
            float4 xr[2][4], gv[2][4];
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int m = m0 + wr * 128 + (ig + ii) * 16 + l15;
                mok[ii] = m < g.M;
                mm[ii] = mok[ii] ? m : g.M - 1;
                if constexpr (FP8) sm[ii] = g.sa[mm[ii]];
                if constexpr (EPI == WAN_EPI_RESID_F32) {
                    const int64_t brow = g.gate ? (int64_t)(mm[ii] / rpb) * g.N : 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        xr[ii][j] = *reinterpret_cast<const float4*>((const float*)g.out + (int64_t)mm[ii] * g.ldo + nn[j]);
                        gv[ii][j] = g.gate ? *reinterpret_cast<const float4*>(g.gate + brow + nn[j]) : make_float4(1.f, 1.f, 1.f, 1.f);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);          // all loads of the batch are issued before the first use waits
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = acc[ig + ii][j];
                    if constexpr (FP8) {
                        v[0] *= sm[ii] * snj[j].x; v[1] *= sm[ii] * snj[j].y; v[2] *= sm[ii] * snj[j].z; v[3] *= sm[ii] * snj[j].w;
                    }
                    v[0] += bj[j].x; v[1] += bj[j].y; v[2] += bj[j].z; v[3] += bj[j].w;
                    if constexpr (EPI == WAN_EPI_GELU_BF16) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f32(v[r]);
                    }
                    if (!(mok[ii] && nok[j])) continue;
                    const int64_t off = (int64_t)mm[ii] * g.ldo + nn[j];
                    if constexpr (EPI == WAN_EPI_BF16 || EPI == WAN_EPI_GELU_BF16) {
                        u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                        *reinterpret_cast<u32x2*>((bf16_t*)g.out + off) = o;
                    } else if constexpr (EPI == WAN_EPI_F32) {
                        *reinterpret_cast<float4*>((float*)g.out + off) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        const float4 x = xr[ii][j], gq = gv[ii][j];
                        *reinterpret_cast<float4*>((float*)g.out + off) =
                            make_float4(x.x + v[0] * gq.x, x.y + v[1] * gq.y, x.z + v[2] * gq.z, x.w + v[3] * gq.w);
                    }
                }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + l15;
            if (n >= g.N) continue;
            const float bv = g.bias ? g.bias[n] : 0.f;
            const float sn = FP8 ? g.sw[n] : 1.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + wr * 128 + i * 16 + l4;
                if (m >= g.M) continue;
                f32x4 v = acc[i][j];
                if constexpr (FP8) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= sn * g.sa[min(m + r, g.M - 1)];
                }
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

// ====================================================================================================
// 4-wave variant of the same 256 x 256 x 64 tile: one wave per SIMD with the whole 512-register file.
//
//   * a wave owns 128 x 128 outputs = 4 x 4 MFMA 32x32x16 tiles = 256 accumulator registers: 192 in AGPRs, 64 in VGPRs
//     (the MFMAs are inline asm so that C/D can be named in either file while A/B come from VGPRs; hipcc's own allocation
//     of a 256-register accumulator tile does not fit the 256 arch VGPRs);
//   * per 16-deep k-step a wave reads 4 A + 4 W fragments (ds_read_b128) for 16 MFMAs -- 0.5 LDS reads per MFMA of 32
//     cycles, against 0.375 per 16-cycle MFMA in the 8-wave kernel above -- one k-step ahead of their use;
//   * ONE barrier per K tile, placed after k-step 2 of 4: the next K tile has been in flight since k-step 3 of the previous
//     tile (>= 48 MFMA slots of flight time), `s_waitcnt vmcnt(0)` + `s_barrier` publish it, the fragments of its k-step 0
//     are fetched during k-step 3 of this tile (no LDS latency exposed at the seam), and the buffer this tile occupied
//     is handed to tile t+2 at the same barrier (every fragment of tile t has been read by then);
//   * tiles are fetched with `buffer_load_dwordx4 ... lds`: the descriptor base advances by SALU, each lane keeps two
//     constant byte offsets per operand, and the descriptor's range check zero-fills rows past M / N (no clamp);
//   * the schedule is program order: `MFMA ; sched_barrier ; <= 1 filler ; sched_barrier` per slot.
// Hazards hipcc does not see for an asm MFMA (CDNA4 guide 5.7): A/B operands only ever come from ds_read (waited for by
// the compiler's own lgkmcnt tracking), accumulate chains are 16 MFMAs apart, and the accumulators are first read by
// VALU code after an explicit 16-state s_nop.
// Measured (profiles/r02/gemm_w4_ab.log): a tie with the 8-wave phased kernel at K = 5120 (1.25-1.29 PFLOP/s), +4-6 % at
// K = 13 824, -10-15 % at K = 1536.  Unlike in attention, where the same structure gained 9 %, a GEMM K tile carries 16 LDS-DMA
// instructions per wave for 64 MFMAs and almost nothing else: each DMA costs the wave ~60-100 cycles of issue (CDNA4 guide,
// per-instruction constants) that a single wave per SIMD cannot hide behind a partner, which is what the two wave groups of
// the 8-wave kernel do for each other.  Used for deep-K shapes only (see the dispatcher).
// ====================================================================================================
constexpr int kW4Threads = 256;

__device__ __forceinline__ u32x4 lds16(const char* p) { return *reinterpret_cast<const u32x4*>(p); }

// RING (round 3): the same wave tiles, MFMA order and epilogues over a FOUR-stage LDS ring of 256 x 32-k tiles (4 x 32 KiB, rows of
// 64 B, chunk' = chunk ^ ((row >> 2) & 3)) instead of two stages of 64 k.  Why: a K tile's 64 KB come through the CU's
// vector-memory path at 64 B/clk -- half of the tile's MFMA time -- and the two-stage form has to bunch its 16 requests per wave into
// 2 of 4 k-steps (they must land before the next barrier), at twice the sustainable rate, with ~1 us of flight time.  Here every
// k-step carries 4 requests (one per 4 MFMA slots), and a request has 2-3 tiles (>= 2048 MFMA cycles) to land: tile t+4's A pieces
// leave in k-step 1 of tile t (right behind the barrier that frees stage t % 4), its W pieces in k-step 0 of tile t+1; the barrier
// in the middle of tile t waits (counted vmcnt) for tile t+1 only.  Price: one barrier per 32 MFMA slots instead of per 64, and
// 64-byte instead of 128-byte row segments per request.  MEASURED (profiles/r03/gemm_ring_ab.log, in process, numerics green under
// every epilogue): 5-9 % SLOWER on every 14B shape (q|k 1202 vs 1308, ffn.0 1222 vs 1307, ffn.2 1128 vs 1231, o 1219 vs 1285
// TFLOP/s) -- flight time and request spacing are not what the two-stage form is short of.  Compiled only with `make EXPERIMENTS=1`
// ("gemm_ring" = 1), like the gemm_exp variants.
template <int EPI, bool RING = false>
__global__ __launch_bounds__(kW4Threads) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w4_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool kTransposed = (EPI == WAN_EPI_BF16_T);
    int tm, tn;
    tile_coords(g, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int hi = lane >> 5, l31 = lane & 31;
    const int nk = g.K / BK;

    // ---- LDS-DMA: an operand tile is 32 pieces of 1 KiB (8 rows x 128 B); wave w copies pieces 8 w .. 8 w + 7.
    // Piece j covers rows 64 w + 8 j + lane / 8; LDS position (lane & 7) of a row holds source chunk (lane & 7) ^ ((row >> 1) & 7).
    // Rows 8 apart differ by 4 in the swizzle, so two per-lane byte offsets (even / odd j) + a scalar row offset cover all 8.
    int a_voff[2], w_voff[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = wid * 64 + p * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        a_voff[p] = (int)(((int64_t)row * g.lda + c * 8) * 2);
        w_voff[p] = (int)(((int64_t)row * g.ldw + c * 8) * 2);
    }
    const char* a_tile = (const char*)(g.A + (int64_t)m0 * g.lda);
    const char* w_tile = (const char*)(g.W + (int64_t)n0 * g.ldw);
    // bytes from the tile origin (row m0, column 0) to the end of the last valid row; 0 for K tiles past the end
    const int64_t a_bytes = ((int64_t)(min(g.M - m0, BM) - 1) * g.lda + g.K) * 2;
    const int64_t w_bytes = ((int64_t)(min(g.N - n0, BN) - 1) * g.ldw + g.K) * 2;
    auto rsrc = [&](const char* tile, int64_t bytes, int kt) {
        const int64_t left = kt < nk ? bytes - (int64_t)kt * BK * 2 : 0;
#if WAN_DEV_EXPERIMENTS
        const int ke = (g.exp & 4) ? (min(kt, nk - 1) & 1) : min(kt, nk - 1);      // every request hits K tiles 0 / 1 (cache-resident)
#else
        const int ke = min(kt, nk - 1);
#endif
        return __builtin_amdgcn_make_buffer_rsrc((void*)(tile + (int64_t)ke * BK * 2), 0, (int)min(left, (int64_t)0x7fffffff), 0x00020000);
    };
    auto stage_piece = [&](__amdgpu_buffer_rsrc_t r, int buf, int operand, int j, int64_t ld) {
#if WAN_DEV_EXPERIMENTS     // `make EXPERIMENTS=1`: timing-only variants behind gemm_exp (tools/kernel_check gemmx); not in the product build
        if (g.exp & (operand ? 1 : 2)) {                // what the DMA instructions cost the lone wave of a SIMD
            if ((g.exp & 8) && operand) {               // ... and what the same bytes cost as plain register loads (results discarded)
                u32x4 sink;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=a"(sink) : "v"(w_voff[j & 1]), "s"(r), "s"((int)((j >> 1) * 16 * ld * 2)));
            }
            return;
        }
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            r, (__attribute__((address_space(3))) void*)(smem + buf * kBufBytes + operand * kOperandBytes + (wid * 8 + j) * 1024), 16,
            operand ? w_voff[j & 1] : a_voff[j & 1], (int)((j >> 1) * 16 * ld * 2), 0, 0);
    };

    // ---- fragment reads: 32 rows x 16 k per fragment; lane = (row l31, k half hi); logical chunk 2 s + hi of a 128-B row
    const int sw = (l31 >> 1) & 7;
    int koff[2][4];                          // [LDS buffer][k-step]: the buffer base (64 KiB) does not fit a ds_read immediate
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) koff[b][s4] = b * kBufBytes + l31 * 128 + (((2 * s4 + hi) ^ sw) << 4);
    const int a_base = wr * 128 * 128, w_base = kOperandBytes + wc * 128 * 128;

    // ---- RING form: staging and fragment addresses of the 32-k stages
    constexpr int kStageBytes = 32768, kRingOperand = 16384;
    int ra_voff = 0, rw_voff = 0;            // piece j of wave w = rows 64 w + 16 j + lane / 4 (64-B rows), LDS chunk lane & 3 <- source chunk
    int rkoff[4][2];                         // [stage][k-step]
    if constexpr (RING) {
        const int row = wid * 64 + (lane >> 2);
        const int c = (lane & 3) ^ ((lane >> 4) & 3);          // (row >> 2) & 3 == (lane >> 4) & 3 for every piece
        ra_voff = (int)(((int64_t)row * g.lda + c * 8) * 2);
        rw_voff = (int)(((int64_t)row * g.ldw + c * 8) * 2);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) rkoff[st][ks] = st * kStageBytes + l31 * 64 + (((2 * ks + hi) ^ ((l31 >> 2) & 3)) << 4);
    }
    const int nk32 = g.K / 32;
    auto rsrc32 = [&](const char* tile, int64_t bytes, int kt) {
        const int64_t left = kt < nk32 ? bytes - (int64_t)kt * 64 : 0;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(tile + (int64_t)min(kt, nk32 - 1) * 64), 0, (int)min(left, (int64_t)0x7fffffff), 0x00020000);
    };
    auto stage_piece32 = [&](__amdgpu_buffer_rsrc_t r, int stage, int operand, int j, int64_t ld) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            r, (__attribute__((address_space(3))) void*)(smem + stage * kStageBytes + operand * kRingOperand + (wid * 4 + j) * 1024), 16,
            operand ? rw_voff : ra_voff, (int)(j * 16 * ld * 2), 0, 0);
    };
    const int ra_base = wr * 128 * 64, rw_base = kRingOperand + wc * 128 * 64;

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    u32x4 af[2][4], wf[2][4];                // [k-step parity][block]
#if WAN_DEV_EXPERIMENTS
    const bool early_w = (g.exp & 16) != 0;        // the W pieces of tile kt+2 leave in k-step 3 of tile kt as well (measured: no gain)
#else
    constexpr bool early_w = false;
#endif

#define GW4_SB() __builtin_amdgcn_sched_barrier(0)
// 12 of the 16 accumulator tiles (192 registers) are pinned to AGPRs, the last M block (4 tiles, 64 registers) to VGPRs:
// with all 256 AGPRs claimed by "+a" operands hipcc's allocator has no slack left and shuffles tiles through scratch.
#define GW4_MFMA_A(ACC, X, Y) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(X), "v"(Y))
#define GW4_MFMA_V(ACC, X, Y) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(X), "v"(Y))
#define GW4_MFMA(ACC, X, Y) do { if (i < 3) GW4_MFMA_A(ACC, X, Y); else GW4_MFMA_V(ACC, X, Y); } while (0)
    // k-step S of the K tile in LDS buffer `buf`: 16 MFMAs; the even slots fetch the fragments of the NEXT k-step (from
    // `nbuf`, k-step NS), the odd slots of the steps that carry DMA issue one piece each.
    auto kstep = [&](auto S_, int buf, int nbuf, auto NS_, auto DMA_, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rw, int dbuf,
                     __amdgpu_buffer_rsrc_t rw2) __attribute__((always_inline)) {
        constexpr int S = decltype(S_)::value, NS = decltype(NS_)::value, DMA = decltype(DMA_)::value;   // DMA: 0 none, 1 pieces 0..7 (A), 2 pieces 8..15 (W)
        (void)buf;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int slot = i * 4 + j;
                if constexpr (kTransposed) GW4_MFMA(acc[i][j], af[S & 1][i], wf[S & 1][j]);
                else GW4_MFMA(acc[i][j], wf[S & 1][j], af[S & 1][i]);
                GW4_SB();
                if (slot % 2 == 0) {
                    const int f = slot / 2;
                    if (f < 4) af[NS & 1][f] = lds16(smem + a_base + f * 32 * 128 + koff[nbuf][NS]);
                    else wf[NS & 1][f - 4] = lds16(smem + w_base + (f - 4) * 32 * 128 + koff[nbuf][NS]);
                    // experiment (gemm_exp & 16): the W pieces of tile kt+2 leave in k-step 3 of tile kt as well (one k-step more flight time)
                    if (DMA == 1 && early_w) stage_piece(rw2, dbuf, 1, slot / 2, g.ldw);
                } else if (DMA != 0) {
                    const int pj = slot / 2;
                    if (DMA == 1) stage_piece(ra, dbuf, 0, pj, g.lda);
                    else if (!early_w) stage_piece(rw, dbuf, 1, pj, g.ldw);
                }
                GW4_SB();
            }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    if constexpr (RING) {
        // k-step KS (0 / 1) of a tile: 16 MFMAs on registers [KS]; even slots read the NEXT k-step's fragments (stage NST, k-step
        // 1 - KS) into registers [1 - KS]; slots 1, 5, 9, 13 issue one DMA piece each (operand OP of the tile behind `r`, stage DST)
        auto rkstep = [&](auto KS_, auto NST_, auto OP_, auto DST_, __amdgpu_buffer_rsrc_t r) __attribute__((always_inline)) {
            constexpr int KS = decltype(KS_)::value, NST = decltype(NST_)::value, OP = decltype(OP_)::value, DST = decltype(DST_)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int slot = i * 4 + j;
                    if constexpr (kTransposed) GW4_MFMA(acc[i][j], af[KS][i], wf[KS][j]);
                    else GW4_MFMA(acc[i][j], wf[KS][j], af[KS][i]);
                    GW4_SB();
                    if (slot % 2 == 0) {
                        const int f = slot / 2;
                        if (f < 4) af[1 - KS][f] = lds16(smem + ra_base + f * 32 * 64 + rkoff[NST][1 - KS]);
                        else wf[1 - KS][f - 4] = lds16(smem + rw_base + (f - 4) * 32 * 64 + rkoff[NST][1 - KS]);
                    } else if (slot % 4 == 1) {
                        stage_piece32(r, DST, OP, slot / 4, OP ? g.ldw : g.lda);
                    }
                    GW4_SB();
                }
        };
        using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>;
        // prologue: tiles 0..2 and the A half of tile 3 in flight; tile 0 waited for
        for (int t = 0; t < 4; ++t) {
            const __amdgpu_buffer_rsrc_t ra = rsrc32(a_tile, a_bytes, t), rw = rsrc32(w_tile, w_bytes, t);
#pragma unroll
            for (int j = 0; j < 4; ++j) stage_piece32(ra, t, 0, j, g.lda);
            if (t < 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) stage_piece32(rw, t, 1, j, g.ldw);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x4F74);          // vmcnt(20): everything but tile 0 may still be in flight
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            af[0][f] = lds16(smem + ra_base + f * 32 * 64 + rkoff[0][0]);
            wf[0][f] = lds16(smem + rw_base + f * 32 * 64 + rkoff[0][0]);
        }
        auto rtile = [&](int t, auto ST_) __attribute__((always_inline)) {        // tile t lives in stage ST = t % 4
            constexpr int ST = decltype(ST_)::value;
            const __amdgpu_buffer_rsrc_t rw3 = rsrc32(w_tile, w_bytes, t + 3);      // W of tile t+3 -> stage (ST + 3) % 4 (its A went out in tile t-1)
            const __amdgpu_buffer_rsrc_t ra4 = rsrc32(a_tile, a_bytes, t + 4);      // A of tile t+4 -> stage ST, behind the barrier
            rkstep(J0{}, std::integral_constant<int, ST>{}, J1{}, std::integral_constant<int, (ST + 3) & 3>{}, rw3);
            __builtin_amdgcn_s_waitcnt(0x4F70);      // vmcnt(16): my pieces of tile t+1 have landed (t+2, t+3 may be in flight) ...
            __builtin_amdgcn_s_barrier();            // ... everybody's have, and every wave has read the last fragment of tile t
            GW4_SB();
            rkstep(J1{}, std::integral_constant<int, (ST + 1) & 3>{}, J0{}, std::integral_constant<int, ST>{}, ra4);
        };
        for (int t = 0; t < nk32; t += 4) {           // K % 128 == 0
            rtile(t, std::integral_constant<int, 0>{});
            rtile(t + 1, std::integral_constant<int, 1>{});
            rtile(t + 2, std::integral_constant<int, 2>{});
            rtile(t + 3, std::integral_constant<int, 3>{});
        }
    } else {
    // ---- prologue: K tile 0 -> buffer 0 (waited for), the A half of K tile 1 -> buffer 1 (in flight), fragments of k-step 0
    {
        const __amdgpu_buffer_rsrc_t ra = rsrc(a_tile, a_bytes, 0), rw = rsrc(w_tile, w_bytes, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) { stage_piece(ra, 0, 0, j, g.lda); stage_piece(rw, 0, 1, j, g.ldw); }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        const __amdgpu_buffer_rsrc_t ra1 = rsrc(a_tile, a_bytes, 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) stage_piece(ra1, 1, 0, j, g.lda);       // its W half goes out in k-step 0 of tile 0
        if (early_w) {
            const __amdgpu_buffer_rsrc_t rw1 = rsrc(w_tile, w_bytes, 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) stage_piece(rw1, 1, 1, j, g.ldw);
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            af[0][f] = lds16(smem + a_base + f * 32 * 128 + koff[0][0]);
            wf[0][f] = lds16(smem + w_base + f * 32 * 128 + koff[0][0]);
        }
    }
    // K tile kt in buffer b (a literal at the call site).  DMA rule: the A pieces of tile kt+2 go out in k-step 3 of tile kt
    // (right after the barrier that frees buffer b), its W pieces in k-step 0 of tile kt+1; both are waited for at the
    // barrier of tile kt+1.  Requests past the last tile carry a zero-length descriptor (no memory traffic).
    auto ktile = [&](int kt, int b) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rw_prev = rsrc(w_tile, w_bytes, kt + 1);      // W of tile kt+1 -> buffer 1-b (its A went out in tile kt-1)
        const __amdgpu_buffer_rsrc_t ra_next = rsrc(a_tile, a_bytes, kt + 2);      // A of tile kt+2 -> buffer b, after the barrier
        const __amdgpu_buffer_rsrc_t rw_next = rsrc(w_tile, w_bytes, kt + 2);      // experiment: W of tile kt+2 together with its A
        kstep(I0{}, b, b, I1{}, I2{}, ra_next, rw_prev, 1 - b, rw_next);
        kstep(I1{}, b, b, I2{}, I0{}, ra_next, rw_prev, b, rw_next);
        kstep(I2{}, b, b, I3{}, I0{}, ra_next, rw_prev, b, rw_next);
        __builtin_amdgcn_s_waitcnt(0x0F70);      // my pieces of tile kt+1 have landed ...
        __builtin_amdgcn_s_barrier();            // ... everybody's have, and every wave has read the last fragment of tile kt
        GW4_SB();
        kstep(I3{}, b, 1 - b, I0{}, I1{}, ra_next, rw_prev, b, rw_next);
    };
    for (int kt = 0; kt < nk; kt += 2) {          // nk is even (the dispatcher sends K % 128 != 0 to the 8-wave kernel)
        ktile(kt, 0);
        ktile(kt + 1, 1);
    }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // zero-length requests past the last tile still write LDS: let them finish
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // MFMA D -> the accumulator reads of the epilogue
#undef GW4_MFMA
#undef GW4_MFMA_A
#undef GW4_MFMA_V
#undef GW4_SB

    // ---- epilogue.  Swapped product (W fragment as A operand): lane (l31, hi) holds output row m = .. + l31 and, per
    // accumulator register quad q, the 4 consecutive columns n = .. + 8 q + 4 hi .. + 3.  Transposed store: roles exchanged.
    if constexpr (!kTransposed) {
        // Batches of 4 (the four register quads of one 32 x 32 tile): bias, residual stream and gate rows of a
        // batch are loaded back to back, one batch ahead of their use (element by element the compiler serialised 64 round trips per
        // tile -- and with one workgroup per CU nothing else runs meanwhile).  Out-of-range rows / columns read a clamped
        // address and are not stored.
        // (the two wave-uniform options -- bias? gate? -- select one of four straight-line copies: tested per element they put a
        // scalar branch and a conservative vmcnt wait between the loads of a batch)
        auto epilogue_rows = [&](auto has_bias, auto has_gate) {
            constexpr bool HAS_BIAS = decltype(has_bias)::value, HAS_GATE = decltype(has_gate)::value;
            const int rpb = HAS_GATE ? (int)g.rows_per_batch : 1;
            struct Batch {
                float4 bq[4], xr[4], gq[4];
                int nn[4], mm;
                bool nok[4], mok;
            };
            auto load_batch = [&](int bi, Batch& B) {          // batch bi = 32-row group bi >> 2, 32-column tile bi & 3
                const int m = m0 + wr * 128 + (bi >> 2) * 32 + l31;
                B.mok = m < g.M;
                B.mm = B.mok ? m : g.M - 1;
                const int64_t brow = HAS_GATE ? (int64_t)(B.mm / rpb) * g.N : 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int n = n0 + wc * 128 + (bi & 3) * 32 + 8 * t + 4 * hi;
                    B.nok[t] = n < g.N;
                    B.nn[t] = B.nok[t] ? n : 0;
                    if constexpr (HAS_BIAS) B.bq[t] = *reinterpret_cast<const float4*>(g.bias + B.nn[t]);
                    else B.bq[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (EPI == WAN_EPI_RESID_F32) {
                        B.xr[t] = *reinterpret_cast<const float4*>((const float*)g.out + (int64_t)B.mm * g.ldo + B.nn[t]);
                        if constexpr (HAS_GATE) B.gq[t] = *reinterpret_cast<const float4*>(g.gate + brow + B.nn[t]);
                        else B.gq[t] = make_float4(1.f, 1.f, 1.f, 1.f);
                    }
                }
            };
            auto store_batch = [&](int bi, const Batch& B) {
                const int i = bi >> 2, j = bi & 3;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[i][j][4 * q] + B.bq[q].x, acc[i][j][4 * q + 1] + B.bq[q].y, acc[i][j][4 * q + 2] + B.bq[q].z,
                               acc[i][j][4 * q + 3] + B.bq[q].w};
                    if constexpr (EPI == WAN_EPI_GELU_BF16) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f32(v[r]);
                    }
                    if (!(B.mok && B.nok[q])) continue;
                    const int64_t off = (int64_t)B.mm * g.ldo + B.nn[q];
                    if constexpr (EPI == WAN_EPI_BF16 || EPI == WAN_EPI_GELU_BF16) {
                        u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                        *reinterpret_cast<u32x2*>((bf16_t*)g.out + off) = o;
                    } else if constexpr (EPI == WAN_EPI_F32) {
                        *reinterpret_cast<float4*>((float*)g.out + off) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        *reinterpret_cast<float4*>((float*)g.out + off) =
                            make_float4(B.xr[q].x + v[0] * B.gq[q].x, B.xr[q].y + v[1] * B.gq[q].y, B.xr[q].z + v[2] * B.gq[q].z,
                                        B.xr[q].w + v[3] * B.gq[q].w);
                    }
                }
            };
            // two batches in flight: the loads of batch b + 1 are issued before batch b is combined and stored
            Batch B[2];
            load_batch(0, B[0]);
#pragma unroll
            for (int bi = 0; bi < 16; ++bi) {
                if (bi + 1 < 16) load_batch(bi + 1, B[(bi + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                store_batch(bi, B[bi & 1]);
            }
        };
        if (g.bias) {
            if (g.gate) epilogue_rows(std::true_type{}, std::true_type{});
            else epilogue_rows(std::true_type{}, std::false_type{});
        } else {
            if (g.gate) epilogue_rows(std::false_type{}, std::true_type{});
            else epilogue_rows(std::false_type{}, std::false_type{});
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 128 + j * 32 + l31;
            if (n >= g.N) continue;
            const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m0 + wr * 128 + i * 32 + 8 * q + 4 * hi;
                    if (m >= g.M) continue;
                    bf16_t* p = (bf16_t*)g.out + (int64_t)n * g.ldo + m;
                    const float v0 = acc[i][j][4 * q] + bv, v1 = acc[i][j][4 * q + 1] + bv, v2 = acc[i][j][4 * q + 2] + bv, v3 = acc[i][j][4 * q + 3] + bv;
                    if (m + 3 < g.M) {
                        u32x2 o = {pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
                        *reinterpret_cast<u32x2*>(p) = o;
                    } else {
                        const float vv[4] = {v0, v1, v2, v3};
                        for (int r = 0; r < 4 && m + r < g.M; ++r) p[r] = (bf16_t)vv[r];
                    }
                }
        }
    }
}

template <int EPI>
wan_status_t launch_w4(const GemmArgs& g, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};
    const wan_status_t st = wan_once_per_device(attr_done, +[]() -> wan_status_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
#if WAN_DEV_EXPERIMENTS
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<EPI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
#endif
        if (e != hipSuccess) {
            wan_set_error("wan_gemm_bf16(w4): cannot reserve %d B of LDS: %s", kLdsBytes, hipGetErrorString(e));
            return WAN_ERR_LAUNCH;
        }
        return WAN_OK;
    });
    if (st != WAN_OK) return st;
#if WAN_DEV_EXPERIMENTS     // `make EXPERIMENTS=1` only: the ring form is a measured alternative, not a product path
    if (wan_tune(WAN_TUNE_GEMM_RING) != 0) {
        hipLaunchKernelGGL((gemm_w4_kernel<EPI, true>), dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(kW4Threads), kLdsBytes, s, g);
        WAN_CHECK_LAUNCH("wan_gemm_bf16(w4 ring)");
        return WAN_OK;
    }
#endif
    hipLaunchKernelGGL((gemm_w4_kernel<EPI>), dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(kW4Threads), kLdsBytes, s, g);
    WAN_CHECK_LAUNCH("wan_gemm_bf16(w4)");
    return WAN_OK;
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

template <int EPI>
wan_status_t launch256_fp8(const GemmArgs& g, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};
    const wan_status_t st = wan_once_per_device(attr_done, +[]() -> wan_status_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<EPI, 2, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) {
            wan_set_error("wan_gemm_fp8: cannot reserve %d B of LDS: %s", kLdsBytes, hipGetErrorString(e));
            return WAN_ERR_LAUNCH;
        }
        return WAN_OK;
    });
    if (st != WAN_OK) return st;
#if WAN_DEV_EXPERIMENTS      // `make EXPERIMENTS=1`: the 4-phase form of the fp8 instantiation for A/B timing (gemm_phases = 4)
    if (wan_tune(WAN_TUNE_GEMM_PHASES) == 4) {
        static std::atomic<uint64_t> attr4{0};
        const wan_status_t st4 = wan_once_per_device(attr4, +[]() -> wan_status_t {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<EPI, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) == hipSuccess ? WAN_OK : WAN_ERR_LAUNCH;
        });
        if (st4 != WAN_OK) return st4;
        hipLaunchKernelGGL((gemm256_kernel<EPI, 4, true>), dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(kThreads), kLdsBytes, s, g);
        WAN_CHECK_LAUNCH("wan_gemm_fp8");
        return WAN_OK;
    }
#endif
    hipLaunchKernelGGL((gemm256_kernel<EPI, 2, true>), dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(kThreads), kLdsBytes, s, g);
    WAN_CHECK_LAUNCH("wan_gemm_fp8");
    return WAN_OK;
}

}  // namespace

extern "C" wan_status_t wan_gemm_fp8(const void* A_fp8, int64_t lda, const float* a_row_scale, const void* W_fp8, int64_t ldw,
                                     const float* w_row_scale, const float* bias, void* out, int64_t ldo, int M, int N, int K,
                                     int epilogue, const float* gate, int64_t rows_per_batch, void* stream) {
    WAN_REQUIRE(A_fp8 && W_fp8 && out && a_row_scale && w_row_scale, WAN_ERR_INVALID, "wan_gemm_fp8: null tensor");
    WAN_REQUIRE(M >= 0 && N > 0 && K > 0, WAN_ERR_INVALID, "wan_gemm_fp8: M=%d N=%d K=%d", M, N, K);
    WAN_REQUIRE(K % 128 == 0, WAN_ERR_UNSUPPORTED, "wan_gemm_fp8: K=%d must be a multiple of 128", K);
    WAN_REQUIRE(N % 4 == 0, WAN_ERR_UNSUPPORTED, "wan_gemm_fp8: N=%d must be a multiple of 4", N);
    WAN_REQUIRE(lda % 16 == 0 && ldw % 16 == 0 && lda >= K && ldw >= K, WAN_ERR_INVALID,
                "wan_gemm_fp8: lda=%lld ldw=%lld must be multiples of 16 and >= K", (long long)lda, (long long)ldw);
    if (epilogue == WAN_EPI_BF16_T)
        WAN_REQUIRE(ldo >= M && ldo % 4 == 0, WAN_ERR_INVALID, "wan_gemm_fp8: transposed ldo=%lld < M=%d or not a multiple of 4", (long long)ldo, M);
    else
        WAN_REQUIRE(ldo >= N && ldo % 4 == 0, WAN_ERR_INVALID, "wan_gemm_fp8: ldo=%lld < N=%d or not a multiple of 4", (long long)ldo, N);
    WAN_REQUIRE(gate == nullptr || (epilogue == WAN_EPI_RESID_F32 && rows_per_batch > 0), WAN_ERR_INVALID,
                "wan_gemm_fp8: gate needs WAN_EPI_RESID_F32 and rows_per_batch > 0");
    if (M == 0) return WAN_OK;
    GemmArgs g;
    g.A = (const bf16_t*)A_fp8; g.lda = lda; g.W = (const bf16_t*)W_fp8; g.ldw = ldw; g.bias = bias;
    g.out = out; g.ldo = ldo; g.gate = gate; g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    g.M = M; g.N = N; g.K = K; g.sa = a_row_scale; g.sw = w_row_scale; g.exp = 0;
    g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
    g.gm = g.tiles_n >= 40 ? 2 : 3;
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case WAN_EPI_BF16: return launch256_fp8<WAN_EPI_BF16>(g, s);
        case WAN_EPI_GELU_BF16: return launch256_fp8<WAN_EPI_GELU_BF16>(g, s);
        case WAN_EPI_F32: return launch256_fp8<WAN_EPI_F32>(g, s);
        case WAN_EPI_RESID_F32: return launch256_fp8<WAN_EPI_RESID_F32>(g, s);
        case WAN_EPI_BF16_T: return launch256_fp8<WAN_EPI_BF16_T>(g, s);
        default: wan_set_error("wan_gemm_fp8: unknown epilogue %d", epilogue); return WAN_ERR_INVALID;
    }
}

// the 4-wave form of the 256^2 tile for this K?  (host arithmetic; also behind wan_gemm_plan)
bool wan_gemm256_uses_w4(int K) {
    const int w4mode = wan_tune(WAN_TUNE_GEMM_W4);
    return K % (2 * BK) == 0 && ((w4mode == 1 && K >= 4096) || (w4mode == 2 && K >= 8192) || w4mode == 3);
}

// called by wan_gemm_bf16 (gemm_bf16.hip) for large shapes; arguments already validated there
wan_status_t wan_gemm_bf16_256(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                               void* out, int64_t ldo, int M, int N, int K, int epilogue,
                               const float* gate, int64_t rows_per_batch, hipStream_t s) {
    GemmArgs g;
    g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.bias = bias;
    g.out = out; g.ldo = ldo; g.gate = gate; g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    g.M = M; g.N = N; g.K = K; g.sa = nullptr; g.sw = nullptr; g.exp = wan_tune(WAN_TUNE_GEMM_EXP);
    g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
    // M tiles per rasterisation group, measured at M = 67 080 (profiles/r01/gemm_raster_group_ab.log): 2 for wide N
    // (qk projection +5 %, ffn.0 +2 % over the former 4), 3 otherwise (+2 %); 8 and 16 lose 10-15 %.
    g.gm = g.tiles_n >= 40 ? 2 : 3;
    if (const int gm = wan_tune(WAN_TUNE_GEMM_GM); gm > 0) g.gm = gm;       // developer A/B switches (wan_set_tuning)
    const int phases = wan_tune(WAN_TUNE_GEMM_PHASES) > 0 ? wan_tune(WAN_TUNE_GEMM_PHASES) : kDefaultPhases;
    // gemm_w4: 0 never, 1 (default) for K % 128 == 0 and K >= 4096, 2 deep K only (K >= 8192), 3 whenever K % 128 == 0.  With the
    // batched epilogues the 4-wave kernel is ahead on every 14B shape in process (o/q 2.675 vs 2.686 ms, q|k 5.43 vs 5.46, ffn.0
    // 7.33 vs 7.68, ffn.2 7.72 vs 7.98; 8-way shards +3..7 %) and by 0.45 % of a whole step in situ
    // (profiles/r02/gemm_epilogue_ab.txt); at K = 1536 it loses 10-15 % (profiles/r02/gemm_w4_ab.log).
    const bool w4 = wan_gemm256_uses_w4(K);
#define WAN_G256(E) (w4 ? launch_w4<E>(g, s) : phases == 4 ? launch256<E, 4>(g, s) : launch256<E, 2>(g, s))
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
