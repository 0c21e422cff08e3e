// WanVAE convolutions as ONE implicit-GEMM kernel on the gfx950 matrix cores.
//
// Replaces CausalConv3d (+ its 2-frame feat_cache halo), the Conv2d of Resample (nearest-exact 2x
// upsample fused into the gather; ZeroPad2d((0,1,0,1)) + stride 2), the (3,1,1) time_conv of
// upsample3d (with the channel->time interleave fused into the store) and every 1x1 conv of
// videox_fun/models/wan_vae.py:21-40, 70-164, 190-266.
//
// Layout: activations are CHANNELS-LAST bf16 [T, H, W, C]; weights are [Cout][KT][KH][KW][Cin]
// flattened to K = taps*Cin (zero padded to a multiple of 64).  Then
//     out[pixel m, n] = sum_k A[m, k] * Wt[n, k],   A[m, (tap, ci)] = in[pixel m shifted by tap][ci]
// is exactly the bf16 GEMM of gemm_bf16.hip, except that the A tile is GATHERED: every 16-byte
// chunk of an A-tile row is 8 consecutive channels of ONE input pixel, so the LDS-DMA staging
// keeps its shape and only the per-lane source address changes (a tap decode + bounds test per
// chunk; out-of-range taps, causal history before the sequence start and K padding read a zero
// page).  Frames with negative time index come from the `hist` buffer (the per-conv history that
// the reference calls feat_cache).
//
// Tile 128 pixels x (32*NT) channels x 64 k, 4 waves (2x2), v_mfma_f32_16x16x32_bf16, double
// buffered LDS, same source-side XOR swizzle as the GEMM.  NT = 1 / 3 / 4 / 6 -> BN = 32 / 96 / 128 / 192 so
// the VAE widths 96 / 192 / 384 waste no MFMA columns.
#include <stdlib.h>

#include <algorithm>

#include "common.hpp"
#include <type_traits>

namespace {

constexpr int BM = 128, BK = 64;
constexpr int kThreads = 256;
constexpr int kATile = BM * BK * 2;   // 16 KiB

__device__ __attribute__((aligned(128))) unsigned int wan_zero_page[64];   // 256 B of zeros

struct ConvArgs {
    const bf16_t* x; const bf16_t* hist; const bf16_t* w; int64_t ldw;
    const float* bias; const bf16_t* resid; bf16_t* out; int64_t ldo;
    int T_in, H_in, W_in, Cin;
    int T_out, H_out, W_out, Cout;
    int KT, KH, KW, st, sh, sw, pt, ph, pw;
    int ups, interleave, hist_frames, silu;
    int M, ntaps, nk, tiles_m, tiles_n;
    unsigned cin_magic;          // ceil(2^32 / Cin)
    unsigned div_w_mul, div_w_sh, div_h_mul, div_h_sh;   // q = (a * mul) >> sh divides by KW / KH (171, 9 for 3; 1, 0 for 1)
    int xcd_slabs;               // 1: XCD-contiguous tile slabs (default), 0: plain order (A/B switch WAN_CONV_XCD=0)
    int pth, ptw;                // conv3_patch_kernel: tiles along H / W
};

__device__ __forceinline__ int div3(int a) { return (a * 171) >> 9; }   // exact for 0 <= a < 256

// a * b + c on the full-rate 24-bit multiplier (a, b < 2^24; b wave-uniform); v_mul_lo_u32 / v_mad_u64_u32 are quarter rate
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned mul24(unsigned a, unsigned b) {
    unsigned r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(b), "v"(a));
    return r;
}

// FAST: the gather address of a chunk without a branch and without 64-bit / quarter-rate multiplies -- the host takes this
// path when every pixel index fits 24 bits and every element offset 32 bits (true for every chunked VAE call up to 720p);
// the general path measured 329 VALU instructions per K tile against 24 MFMAs (VALU-bound at 22 % of the matrix peak).
template <int NT, bool FAST>
__global__ __launch_bounds__(kThreads, NT == 6 ? 1 : 2) void conv_cl_kernel(ConvArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BN = 32 * NT;
    constexpr int kWTile = BN * BK * 2;
    constexpr int kStage = kATile + kWTile;
    constexpr int WP = BN / 8 / 4;          // W pieces (8 rows each) per wave: 1, 3, 4, 6

    // Tile coordinates.  Block b runs on XCD b % 8 (observed): each XCD takes a contiguous slab of the tile sequence
    // (bijective remap as in the GEMM) so that neighbouring pixel tiles -- whose 3x3x3 gathers overlap by whole image
    // rows -- share one XCD's L2; N fastest inside the sequence so the N tiles of one pixel tile run together.
    int t = blockIdx.x;
    if (g.xcd_slabs) {
        const int nwg = g.tiles_m * g.tiles_n;
        const int xcd = t & 7, loc = t >> 3, q = nwg >> 3, r = nwg & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tn = t % g.tiles_n, tm = t / g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;

    // ---- A gather bookkeeping: piece j of this wave = tile rows (wid*4+j)*8 .. +7
    const int srow = lane >> 3, spc = lane & 7;
    int ti0[4], hw0[4];          // hw0 packs (hi0 + 4096) << 16 | (wi0 + 4096)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + (wid * 4 + j) * 8 + srow;
        if (m < g.M) {
            const int wo = m % g.W_out;
            const int r = m / g.W_out;
            const int ho = r % g.H_out, to = r / g.H_out;
            ti0[j] = to * g.st - g.pt;
            hw0[j] = ((ho * g.sh - g.ph + 4096) << 16) | (wo * g.sw - g.pw + 4096);
        } else {
            ti0[j] = -(1 << 20);           // never valid
            hw0[j] = (4096 << 16) | 4096;
        }
    }
    // logical chunk of this lane for even / odd pieces (swizzle depends on (row >> 1) & 7)
    const int cl[2] = {spc ^ ((srow >> 1) & 7), spc ^ ((4 + (srow >> 1)) & 7)};
    const int Hlim = g.H_in << g.ups, Wlim = g.W_in << g.ups;
    const int64_t frame_elems = (int64_t)g.H_in * g.W_in * g.Cin;
    // FAST: frames are counted from the first history frame (tsel = ti + hist_frames >= 0 for every valid tap), addresses are
    // xbase + 2 * element(tsel, hi, wi, ci) with the (wave-uniform) distance to the history buffer added for tsel < hist_frames
    int hi0[4], wi0[4];
    unsigned tb0[4];
    uint64_t xbase = 0, hdelta = 0, zpage = 0;
    if constexpr (FAST) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hi0[j] = (hw0[j] >> 16) - 4096;
            wi0[j] = (hw0[j] & 0xffff) - 4096;
            tb0[j] = (unsigned)(ti0[j] + g.hist_frames);
        }
        xbase = (uint64_t)g.x - (uint64_t)g.hist_frames * (uint64_t)frame_elems * 2;
        hdelta = g.hist_frames ? (uint64_t)g.hist - xbase : 0;
        zpage = (uint64_t)wan_zero_page;
    }
    // the three candidate bases as VGPR pairs, materialised once (as SGPR values the selects cost two v_mov each per chunk, and
    // the zero page's address is re-read through the GOT every K tile)
    unsigned xb_lo = (unsigned)xbase, xb_hi = (unsigned)(xbase >> 32);
    unsigned hb_lo = (unsigned)(xbase + hdelta), hb_hi = (unsigned)((xbase + hdelta) >> 32);
    unsigned zp_lo = (unsigned)zpage, zp_hi = (unsigned)(zpage >> 32);
    if constexpr (FAST) asm volatile("" : "+v"(xb_lo), "+v"(xb_hi), "+v"(hb_lo), "+v"(hb_hi), "+v"(zp_lo), "+v"(zp_hi));
    const unsigned Tlim = (unsigned)(g.T_in + g.hist_frames);

    const bf16_t* w_src[WP];
#pragma unroll
    for (int j = 0; j < WP; ++j) {
        const int row = (wid * WP + j) * 8 + srow;
        const int c = spc ^ ((row >> 1) & 7);
        const int n = min(n0 + row, g.Cout - 1);
        w_src[j] = g.w + (int64_t)n * g.ldw + c * 8;
    }

    auto stage = [&](int buf, int kstep) {
        char* sbase = smem + buf * kStage;
        if constexpr (FAST) {
            unsigned dt[2], ci[2];
            int dh[2], dw[2];
            bool tapok[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const unsigned k = (unsigned)(kstep * BK + cl[p] * 8);
                const unsigned tap = __umulhi(k, g.cin_magic);
                ci[p] = k - mul24(tap, (unsigned)g.Cin);
                tapok[p] = tap < (unsigned)g.ntaps;
                const unsigned q1 = mul24(tap, g.div_w_mul) >> g.div_w_sh;
                dw[p] = (int)(tap - mul24(q1, (unsigned)g.KW));
                const unsigned q2 = mul24(q1, g.div_h_mul) >> g.div_h_sh;
                dh[p] = (int)(q1 - mul24(q2, (unsigned)g.KH));
                dt[p] = q2;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = j & 1;
                const unsigned tsel = tb0[j] + dt[p];
                const int hi = hi0[j] + dh[p], wi = wi0[j] + dw[p];
                const bool ok = tapok[p] & ((unsigned)hi < (unsigned)Hlim) & ((unsigned)wi < (unsigned)Wlim) & (tsel < Tlim);
                const unsigned row = mad24(tsel, (unsigned)g.H_in, (unsigned)(hi >> g.ups));
                const unsigned pix = mad24(row, (unsigned)g.W_in, (unsigned)(wi >> g.ups));
                const unsigned el = mad24(pix, (unsigned)g.Cin, ci[p]);
                const bool hist = tsel < (unsigned)g.hist_frames;
                const uint64_t b = ((uint64_t)(hist ? hb_hi : xb_hi) << 32) | (hist ? hb_lo : xb_lo);
                const uint64_t a = b + ((uint64_t)el << 1);
                const uint64_t src = ((uint64_t)(ok ? (unsigned)(a >> 32) : zp_hi) << 32) | (ok ? (unsigned)a : zp_lo);
                glds16(reinterpret_cast<const void*>(src), sbase + (wid * 4 + j) * 1024);
            }
        } else {
            // decode (tap, ci) once per swizzle parity
            int dt[2], dh[2], dw[2], ci[2];
            bool tapok[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const unsigned k = (unsigned)(kstep * BK + cl[p] * 8);
                const int tap = (int)__umulhi(k, g.cin_magic);
                ci[p] = (int)k - tap * g.Cin;
                tapok[p] = tap < g.ntaps;
                int rest = tap;
                int kw = 0, kh = 0;
                if (g.KW == 3) { const int q = div3(rest); kw = rest - 3 * q; rest = q; }
                if (g.KH == 3) { const int q = div3(rest); kh = rest - 3 * q; rest = q; }
                dt[p] = rest; dh[p] = kh; dw[p] = kw;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = j & 1;
                const int ti = ti0[j] + dt[p];
                const int hi = (hw0[j] >> 16) - 4096 + dh[p];
                const int wi = (hw0[j] & 0xffff) - 4096 + dw[p];
                const bool ok = tapok[p] && (unsigned)hi < (unsigned)Hlim && (unsigned)wi < (unsigned)Wlim &&
                                ti < g.T_in && ti >= -g.hist_frames;
                const bf16_t* src = reinterpret_cast<const bf16_t*>(wan_zero_page);
                if (ok) {
                    const int64_t pix = ((int64_t)(hi >> g.ups) * g.W_in + (wi >> g.ups)) * g.Cin + ci[p];
                    src = ti >= 0 ? g.x + ti * frame_elems + pix : g.hist + (ti + g.hist_frames) * frame_elems + pix;
                }
                glds16(src, sbase + (wid * 4 + j) * 1024);
            }
        }
        const int koff = kstep * BK;
#pragma unroll
        for (int j = 0; j < WP; ++j) glds16(w_src[j] + koff, sbase + kATile + (wid * WP + j) * 1024);
    };

    // ---- fragment read offsets (identical to gemm_bf16.hip)
    const int frow = lane & 15, kg = lane >> 4, sw_ = (lane >> 1) & 7;
    const int off_k0 = frow * 128 + ((kg ^ sw_) << 4);
    const int off_k1 = frow * 128 + (((kg + 4) ^ sw_) << 4);
    const int a_base = wr * 64 * 128;
    const int w_base = kATile + wc * (16 * NT) * 128;

    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int kt = 0; kt < g.nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < g.nk) stage(cur ^ 1, kt + 1);
        const char* sb = smem + cur * kStage;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int off = kk ? off_k1 : off_k0;
            bf16x8 af[4], wf[NT];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sb + a_base + i * 2048 + off);
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + w_base + j * 2048 + off);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }

    // ---- epilogue: lane holds pixel m = l&15, channels n = (l>>4)*4 + r.  The bias of the NT column groups is loaded once, the
    // residual values of one pixel group back to back (clamped addresses, guarded stores), and the wave-uniform options (bias?
    // residual?) select one of four straight-line copies: written element by element the compiler emitted load / wait / store once
    // per (pixel group, column group).
    const int l15 = lane & 15, l4 = (lane >> 4) * 4;
    const int Ch = g.interleave ? g.Cout >> 1 : g.Cout;
    auto write_tiles = [&](auto has_bias, auto has_resid) {
        constexpr bool HAS_BIAS = decltype(has_bias)::value, HAS_RESID = decltype(has_resid)::value;
        int col[NT], half[NT];
        bool nok[NT];
        float4 bv[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            int n = n0 + wc * (16 * NT) + j * 16 + l4;
            nok[j] = n < g.Cout;
            n = nok[j] ? n : 0;
            if constexpr (HAS_BIAS) bv[j] = *reinterpret_cast<const float4*>(g.bias + n);
            half[j] = g.interleave && n >= Ch;          // frame 2t + half, channel n - half*Ch   (wan_vae.py:138-141)
            col[j] = n - half[j] * Ch;
        }
        const int hw = g.H_out * g.W_out;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wr * 64 + i * 16 + l15;
            const bool mok = m < g.M;
            const int mm = mok ? m : 0;
            int to = 0, rem = 0;
            if (g.interleave) {
                to = mm / hw;
                rem = mm - to * hw;
            }
            int64_t off[NT];
            u32x2 rv[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int64_t orow = g.interleave ? (int64_t)(2 * to + half[j]) * hw + rem : (int64_t)mm;
                off[j] = orow * g.ldo + col[j];
                if constexpr (HAS_RESID) rv[j] = *reinterpret_cast<const u32x2*>(g.resid + off[j]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                f32x4 v = acc[i][j];
                if constexpr (HAS_BIAS) { v[0] += bv[j].x; v[1] += bv[j].y; v[2] += bv[j].z; v[3] += bv[j].w; }
                if constexpr (HAS_RESID) {
                    v[0] += bf16lo_to_f32(rv[j][0]); v[1] += bf16hi_to_f32(rv[j][0]);
                    v[2] += bf16lo_to_f32(rv[j][1]); v[3] += bf16hi_to_f32(rv[j][1]);
                }
                if (mok && nok[j]) {
                    u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *reinterpret_cast<u32x2*>(g.out + off[j]) = o;
                }
            }
        }
    };
    if (g.bias) {
        if (g.resid) write_tiles(std::true_type{}, std::true_type{});
        else write_tiles(std::true_type{}, std::false_type{});
    } else {
        if (g.resid) write_tiles(std::false_type{}, std::true_type{});
        else write_tiles(std::false_type{}, std::false_type{});
    }
}

template <int NT, bool FAST>
wan_status_t launch_conv_t(const ConvArgs& g, hipStream_t s) {
    constexpr int lds = 2 * (kATile + 32 * NT * BK * 2);
    static std::atomic<uint64_t> attr_done{0};
    const wan_status_t st = wan_once_per_device(attr_done, +[]() -> wan_status_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_cl_kernel<NT, FAST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            wan_set_error("wan_conv_cl: cannot reserve %d B of LDS: %s", lds, hipGetErrorString(e));
            return WAN_ERR_LAUNCH;
        }
        return WAN_OK;
    });
    if (st != WAN_OK) return st;
    hipLaunchKernelGGL((conv_cl_kernel<NT, FAST>), dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(kThreads), lds, s, g);
    WAN_CHECK_LAUNCH("wan_conv_cl");
    return WAN_OK;
}

template <int NT>
wan_status_t launch_conv(const ConvArgs& g, hipStream_t s) {
    // FAST needs: pixel indices < 2^24 and element offsets < 2^32 (24-bit multiplies, 32-bit element index)
    const int64_t px = (int64_t)(g.T_in + g.hist_frames) * g.H_in * g.W_in;       // frames counted from the first history frame
    const bool fits = px < (1 << 24) && px * g.Cin < (1LL << 32) && g.Cin < (1 << 24) && g.H_in < (1 << 24) && g.W_in < (1 << 24);
    return fits && wan_tune(WAN_TUNE_CONV_FAST) ? launch_conv_t<NT, true>(g, s) : launch_conv_t<NT, false>(g, s);
}

// ------------------------------------------------------------------ causal 3x3x3, stride 1: the input patch lives in LDS
// The residual blocks' CausalConv3d(3x3x3, stride 1, padding (2,1,1)) carry ~all of the VAE's FLOPs.  As an implicit GEMM
// every 16-byte chunk of the A tile is gathered separately -- 27 times per input element, each with its own address and
// bounds arithmetic (conv_cl_kernel: ~100 VALU instructions per 24 MFMAs at Cout = 96).  Here a workgroup owns 8 x 32 output
// pixels of one frame and loads their (3 frames) x 10 x 34 input patch for 32 channels ONCE into LDS (64 KiB, zero-filled
// outside the image / before the sequence start, history frames from `hist`); the 27 taps are then 27 shifted views of the same
// patch, read by ds_read_b128 at (pixel + tap shift) -- no global address arithmetic in the K loop at all.
//   K loop: for each 32-channel chunk (patch double-buffered, the next chunk's patch streams in one 1-KiB piece per step under this
//   chunk's first 16 taps), for each tap: one 96 x 32 weight slice through a 5-deep LDS ring (issued 4 steps ahead, counted vmcnt,
//   published one step before its use), the NEXT step's 4 A + 6 W fragments read into a second register set between this step's
//   first MFMAs, 12 x v_mfma_f32_32x32x16_bf16 per wave (wave tile 64 pixels x 96 channels = 2 rows x 32 columns; one wave per SIMD,
//   so the 32-cycle MFMA is the one whose issue gaps hold the step's ~50 other instructions), one barrier.
//   LDS: pixel-major, 64 B per pixel (weights: per output channel); the 16-byte chunk position is XOR-swizzled by bits 2..3 of the
//   pixel index -- conflict-free for ds_read_b128's lane groups at every tap alignment (the four pixels of a lane group that share
//   pixel & 3 sit 12 / 20 / 24 pixels apart: (pixel >> 2) & 3 takes four different values).
constexpr int PT_H = 8, PT_W = 32, PP_H = PT_H + 2, PP_W = PT_W + 2, PP_FR = PP_H * PP_W, PP_PX = 3 * PP_FR;   // 1020 patch pixels
constexpr int kPatchBytes = 1024 * 64;
constexpr int kWSlice = 96 * 64, kWRing = 5;
constexpr int kPatchLds = 2 * kPatchBytes + kWRing * kWSlice + 2048;       // all 163 840 B (2 KiB: the dummy DMA target)

// MI = 32: v_mfma_f32_32x32x16_bf16, wave tile 2 x 3 tiles (rounds 2-4; still what under-filled launches run).  MI = 16 (round 5): the same
// loop on v_mfma_f32_16x16x32_bf16 -- wave tile 4 pixel groups x 6 channel groups, ONE MFMA of K = 32 per (group, group) and tap, the
// same ten fragment reads per step (one per two MFMAs), the same LDS images (a lane reads chunk lane >> 4 of pixel / channel row lane & 15).
// The bare instruction sustains 2.0 PF/s at the power limit where 32x32x16 sustains 1.6-1.78 (tools/probe/mfma_power.hip); whether
// THIS loop -- one barrier per 384 MFMA cycles -- can use that was measured: the clock rises by 12 %, the
// matrix-pipe occupancy falls from 57 to 52 % (24 issue slots of 16 cycles carry the step's ~45 other instructions worse than 12 of
// 32), net +0.4 ... +3.4 % on the full launches.
template <int MI>
__global__ __launch_bounds__(kThreads, 1) void conv3_patch_kernel(ConvArgs g) {
    static_assert(MI == 32 || MI == 16, "32x32x16 or 16x16x32");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wring = smem + 2 * kPatchBytes;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hf = g.hist_frames;
    const int nch = g.Cin >> 5;

    // ---- persistent workgroups: block b runs on XCD b & 7 (observed); each XCD owns a contiguous slab of the tile sequence
    // (slice fastest, then W, H, frame) and its workgroups walk it interleaved, so the tiles in flight on one XCD at any time
    // are neighbours whose halos and weight slices share that XCD's L2.  One launch = one workgroup per CU (160 KiB of LDS).
    const int ntiles = g.T_out * g.pth * g.ptw * g.tiles_n;
    int tile, tile_end, tile_step;
    {
        const int b = blockIdx.x, G = gridDim.x;
        if (g.xcd_slabs && (G & 7) == 0) {
            const int xcd = b & 7, q = ntiles >> 3, r = ntiles & 7;
            const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
            tile_end = start + q + (xcd < r ? 1 : 0);
            tile = start + (b >> 3);
            tile_step = G >> 3;
        } else {
            tile = b; tile_end = ntiles; tile_step = G;
        }
    }
    if (tile >= tile_end) return;

    // ---- patch staging: DMA instruction i of this wave fills patch pixels (wid*16 + i)*16 .. +15, lane -> (pixel, chunk slot).
    // Tile-independent part, once: the pixel's (frame, row, column) inside the patch and its element offset from the tile's origin.
    int ploc[16];               // ((f * H + r) * W + c) * Cin + 8 * chunk-of-this-lane
    int pfrc[16];               // f << 16 | r << 8 | c, or -1 past the 1020th patch pixel
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int px = (wid * 16 + i) * 16 + (lane >> 2);
        const int f = px / PP_FR, rem = px - f * PP_FR;
        const int r = rem / PP_W, c = rem - r * PP_W;
        const int kgl = (lane & 3) ^ ((px >> 2) & 3);
        ploc[i] = (int)mad24(mad24(mad24((unsigned)f, (unsigned)g.H_in, (unsigned)r), (unsigned)g.W_in, (unsigned)c), (unsigned)g.Cin,
                             (unsigned)(kgl * 8));
        pfrc[i] = px < PP_PX ? (f << 16) | (r << 8) | c : -1;
    }
    const uint64_t frame_bytes = (uint64_t)g.H_in * g.W_in * g.Cin * 2;
    const uint64_t xbase = (uint64_t)g.x - (uint64_t)hf * frame_bytes;       // frames are counted from the first history frame
    const uint64_t hbase = hf ? (uint64_t)g.hist : xbase;
    const uint64_t zpage = (uint64_t)wan_zero_page;
    struct TileDesc { int t0, h0, w0, n0; unsigned valid, hist; int origin; };
    auto describe = [&](int tl) {
        TileDesc d;
        const int sl = tl % g.tiles_n; tl /= g.tiles_n;
        const int twi = tl % g.ptw; tl /= g.ptw;
        const int thi = tl % g.pth;
        d.t0 = tl / g.pth; d.h0 = thi * PT_H; d.w0 = twi * PT_W; d.n0 = sl * 96;
        // element offset of patch pixel (0, 0, 0) = input (t0 - 2 + hf, h0 - 1, w0 - 1); negative at the borders, where it is not used
        d.origin = (((d.t0 - 2 + hf) * g.H_in + d.h0 - 1) * g.W_in + d.w0 - 1) * g.Cin;
        d.valid = 0; d.hist = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = pfrc[i] >> 16, r = (pfrc[i] >> 8) & 0xff, c = pfrc[i] & 0xff;
            const int tsel = d.t0 - 2 + f + hf;
            const bool ok = pfrc[i] >= 0 && (unsigned)(d.h0 - 1 + r) < (unsigned)g.H_in && (unsigned)(d.w0 - 1 + c) < (unsigned)g.W_in && tsel >= 0;
            d.valid |= (unsigned)ok << i;
            d.hist |= (unsigned)(tsel < hf) << i;
        }
        return d;
    };
    auto issue_patch_piece = [&](const TileDesc& d, int chunk, int buf, int i) {
        const uint64_t b = (d.hist >> i) & 1 ? hbase : xbase;
        const uint64_t a = b + ((uint64_t)(unsigned)(d.origin + ploc[i] + chunk * 32) << 1);
        glds16(reinterpret_cast<const void*>((d.valid >> i) & 1 ? a : zpage), smem + buf * kPatchBytes + (wid * 16 + i) * 1024);
    };

    // ---- weight slices: six 16-row DMA pieces per 96-row slice; wave w stages piece w, waves 0 / 1 also pieces 4 / 5, waves 2 / 3
    // a dummy piece into 1 KiB of spare LDS instead (every wave issues exactly two DMA instructions per step -- the counted
    // vmcnt below relies on it -- and no exec-masked branch splits the step's basic block)
    const int nrowA = 16 * wid + (lane >> 2), nrowB = 64 + 16 * (wid & 1) + (lane >> 2);
    const int wcolA = ((lane & 3) ^ ((nrowA >> 2) & 3)) << 3, wcolB = ((lane & 3) ^ ((nrowB >> 2) & 3)) << 3;
    char* const wspare = wring + kWRing * kWSlice + (wid & 1) * 1024;
    struct WPtr { const bf16_t* a; const bf16_t* b; };
    auto wptr = [&](int n0) {
        return WPtr{g.w + (int64_t)min(n0 + nrowA, g.Cout - 1) * g.ldw + wcolA, g.w + (int64_t)min(n0 + nrowB, g.Cout - 1) * g.ldw + wcolB};
    };
    auto issue_w = [&](const WPtr& w, int chunk, int tap, int slot) {
        const int koff = tap * g.Cin + chunk * 32;
        char* dst = wring + slot * kWSlice;
        glds16(w.a + koff, dst + wid * 1024);
        glds16(w.b + koff, wid < 2 ? dst + (4 + wid) * 1024 : wspare);
    };

    // ---- fragment addresses (32x32x16: lane -> row lane & 31 of the tile, 8 channels at 16 * kstep + 8 * (lane >> 5))
    const int m32 = lane & 31, hi1 = lane >> 5;
    int pb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) pb[i] = (2 * wid + i) * PP_W + m32;
    const int woff0 = (m32 << 6) | (((hi1 ^ (m32 >> 2)) & 3) << 4);          // k-step 0; k-step 1 is this ^ 32; n-tile j adds j * 2048

    f32x16 acc[2][3];
    bf16x8 af[2][2][2], wf[2][3][2];          // [set][tile][k-step]
    auto a_addr = [&](int i, int shift) {
        const int px = pb[i] + shift;
        return (px << 6) | (((hi1 ^ (px >> 2)) & 3) << 4);
    };
    // MI == 16: lane -> (row l15 of a 16-row group, 16-byte chunk kg of its 64-byte record); pixel group gq = 16 columns of row gq >> 1
    const int l15 = lane & 15, kg = lane >> 4;
    // (the second group of a row is 16 pixels = 1024 bytes further with the SAME swizzle: (px + 16) >> 2 = (px >> 2) + 4 -- one address per row)
    int pb16[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) pb16[i] = (2 * wid + i) * PP_W + l15;
    const int woff16 = (l15 << 6) | (((kg ^ (l15 >> 2)) & 3) << 4);          // channel group jj adds jj * 1024
    f32x4 acc16[4][6];
    bf16x8 af16[2][4], wf16[2][6];            // [set][group]
    auto a_addr16 = [&](int i, int shift) {                 // row i of the wave, pixel group 0; group 1 of the row: + 1024
        const int px = pb16[i] + shift;
        return (px << 6) | (((kg ^ (px >> 2)) & 3) << 4);
    };
    TileDesc cur = describe(tile);
    WPtr wcur = wptr(cur.n0);
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) issue_patch_piece(cur, 0, 0, i);
        issue_w(wcur, 0, 0, 0);
        issue_w(wcur, 0, 1, 1);
        issue_w(wcur, 0, 2, 2);
        issue_w(wcur, 0, 3, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (MI == 32) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int a0 = a_addr(i, 0);
                af[0][i][0] = *reinterpret_cast<const bf16x8*>(smem + a0);
                af[0][i][1] = *reinterpret_cast<const bf16x8*>(smem + (a0 ^ 32));
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                wf[0][j][0] = *reinterpret_cast<const bf16x8*>(wring + woff0 + j * 2048);
                wf[0][j][1] = *reinterpret_cast<const bf16x8*>(wring + (woff0 ^ 32) + j * 2048);
            }
        } else {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) af16[0][gq] = *reinterpret_cast<const bf16x8*>(smem + a_addr16(gq >> 1, 0) + (gq & 1) * 1024);
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) wf16[0][jj] = *reinterpret_cast<const bf16x8*>(wring + woff16 + jj * 1024);
        }
    }
    int slot_use = 0;          // ring slot of the current step's slice; the slice issued now goes 4 slots further
    int gc = 0;                // chunks done by this workgroup: the patch buffer of a chunk is gc & 1

    // One 32-channel chunk = 27 steps.  `nd` / `nchunk`: the tile and chunk whose patch streams in under this chunk (the same
    // tile's next chunk, or chunk 0 of this workgroup's next tile); the weight slices issued 4 steps ahead wrap into it as well.
    auto run_chunk = [&](const WPtr& wc, int c, const TileDesc& nd, const WPtr& wn, int nchunk) {
        const char* pbuf = smem + (gc & 1) * kPatchBytes;
        const char* pother = smem + ((gc + 1) & 1) * kPatchBytes;
#pragma unroll
        for (int s = 0; s < 27; ++s) {
            const int slot_next = slot_use == kWRing - 1 ? 0 : slot_use + 1;
            // this step's 12 MFMAs run on the fragment set loaded one step ago; the NEXT step's fragments (its slice was published
            // by the previous barrier, the patch is stable) are read between the first ten of them, one read per MFMA, so their LDS
            // latency hides under the rest.  The order is pinned (sched_barrier): left to itself the scheduler sinks the reads to
            // the end of the step, where the barrier waits their latency out.
            {
                const int cs = s & 1, nx = cs ^ 1;
                const char* pnext = s == 26 ? pother : pbuf;
                const int tnext = s == 26 ? 0 : s + 1;
                const int shift = (tnext / 9) * PP_FR + ((tnext / 3) % 3) * PP_W + (tnext % 3);
                const char* wb = wring + slot_next * kWSlice;
                if constexpr (MI == 32) {
                    const int an[2] = {a_addr(0, shift), a_addr(1, shift)};
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        const int ks = k / 6, i = (k % 6) / 3, j = k % 3;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cs][j][ks], af[cs][i][ks], acc[i][j], 0, 0, 0);
                        if (k < 10) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (k < 4) af[nx][k >> 1][k & 1] = *reinterpret_cast<const bf16x8*>(pnext + (an[k >> 1] ^ ((k & 1) << 5)));
                            else wf[nx][(k - 4) >> 1][k & 1] = *reinterpret_cast<const bf16x8*>(wb + ((woff0 ^ ((k & 1) << 5)) + ((k - 4) >> 1) * 2048));
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else {
                    // 24 MFMAs of 16 cycles; the next step's ten fragments one per TWO MFMAs (the same reads per matrix-pipe cycle)
                    const int an16[2] = {a_addr16(0, shift), a_addr16(1, shift)};
#pragma unroll
                    for (int k = 0; k < 24; ++k) {
                        const int gq = k / 6, jj = k % 6;
                        acc16[gq][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf16[cs][jj], af16[cs][gq], acc16[gq][jj], 0, 0, 0);
                        if (k < 20 && (k & 1) == 0) {
                            const int f = k >> 1;
                            __builtin_amdgcn_sched_barrier(0);
                            if (f < 4) af16[nx][f] = *reinterpret_cast<const bf16x8*>(pnext + an16[f >> 1] + (f & 1) * 1024);
                            else wf16[nx][f - 4] = *reinterpret_cast<const bf16x8*>(wb + woff16 + (f - 4) * 1024);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
            // weight slice of step + 4 (in the last 4 steps: of the chunk that follows) and one of the 16 pieces of that chunk's patch
            if (s + 4 < 27) issue_w(wc, c, s + 4, slot_use == 0 ? kWRing - 1 : slot_use - 1);
            else issue_w(wn, nchunk, s + 4 - 27, slot_use == 0 ? kWRing - 1 : slot_use - 1);
            if (s < 16) issue_patch_piece(nd, nchunk, (gc + 1) & 1, s);
            // the slice of step + 2 must have landed before the barrier publishes it.  vmcnt retires in order: behind that slice
            // in the queue are the slices of steps + 3 and + 4 (2 DMA instructions each) and the patch pieces issued in this
            // step and the two before it (steps 0..15 issue one each)
            // (a bare s_barrier: __syncthreads() carries a fence that waits for vmcnt(0) and would serialise the ring)
            __builtin_amdgcn_sched_barrier(0);          // all MFMAs of the step are issued before it waits
            {
                constexpr auto pieces = [](int a) { return a >= 0 && a < 16 ? 1 : 0; };
                const int np = pieces(s - 2) + pieces(s - 1) + pieces(s);
                if (np == 3) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
                else if (np == 2) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                else if (np == 1) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);          // nothing of this step sinks below its barrier, nothing of the next rises
            slot_use = slot_next;
        }
        // 27 steps per chunk: the set prefetched by the last step is set 1, the next chunk starts on set 0
        if constexpr (MI == 32) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { af[0][i][0] = af[1][i][0]; af[0][i][1] = af[1][i][1]; }
#pragma unroll
            for (int j = 0; j < 3; ++j) { wf[0][j][0] = wf[1][j][0]; wf[0][j][1] = wf[1][j][1]; }
        } else {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) af16[0][gq] = af16[1][gq];
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) wf16[0][jj] = wf16[1][jj];
        }
        ++gc;
    };

    for (; tile < tile_end; tile += tile_step) {
        if constexpr (MI == 32) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        } else {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) acc16[gq][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int c = 0; c + 1 < nch; ++c) run_chunk(wcur, c, cur, wcur, c + 1);
        // the last chunk prefetches this workgroup's next tile (past the end: an all-invalid patch -- zero-page reads -- and this
        // tile's slices again, into ring slots nobody reads any more)
        TileDesc nxt = cur;
        nxt.valid = 0;
        if (tile + tile_step < tile_end) nxt = describe(tile + tile_step);
        const WPtr wnxt = wptr(nxt.n0);
        run_chunk(wcur, nch - 1, nxt, wnxt, 0);

        // ---- epilogue.  The accumulators hold, per lane, 4-channel groups of one pixel: stored directly that is 24 scattered 8-byte
        // stores per lane (measured: 7.6 us per tile, a quarter of the kernel).  Instead each wave transposes its two 32-pixel rows
        // through LDS -- fp32, in its own 16 KiB of the patch buffer the tile's last chunk just finished with (only this wave's
        // DMA pieces ever land there, and it issues the next ones after this) -- and writes whole pixels: 16 bytes per lane,
        // consecutive lanes consecutive addresses (a 32-pixel row of a 96-channel tensor is one contiguous 6 KiB run), the bias
        // and the residual added in fp32 on the way out (one rounding, as in the gather kernel).
        {
            char* stg = smem + ((gc - 1) & 1) * kPatchBytes + wid * 16384;
            constexpr int kStgPx = 96 * 4 + 16;                     // bytes per pixel in the staging rows (+16: spreads the banks)
            // the two wave-uniform options (bias? residual?) select one of four straight-line copies, and all loads of a row --
            // staging reads, bias, residual -- are issued before the first is used: tested per element the compiler put a scalar
            // branch and a wait between them, 12 memory round trips per tile with nothing else running on the CU
            auto write_rows = [&](auto has_bias, auto has_resid) {
                constexpr bool HAS_BIAS = decltype(has_bias)::value, HAS_RESID = decltype(has_resid)::value;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if constexpr (MI == 32) {
#pragma unroll
                        for (int j = 0; j < 3; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                *reinterpret_cast<f32x4*>(stg + m32 * kStgPx + (j * 32 + q * 8 + hi1 * 4) * 4) =
                                    f32x4{acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
                    } else {
                        // D = mfma(weights, pixels): lane (l15, kg) holds channels 16 jj + 4 kg .. + 3 of pixel l15 of its group
#pragma unroll
                        for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                            for (int jj = 0; jj < 6; ++jj)
                                *reinterpret_cast<f32x4*>(stg + (gg * 16 + l15) * kStgPx + (jj * 16 + kg * 4) * 4) = acc16[2 * i + gg][jj];
                    }
                    const int h = cur.h0 + 2 * wid + i;
                    const int64_t orow0 = ((int64_t)cur.t0 * g.H_out + h) * g.W_out + cur.w0;
                    f32x4 lo[6], hi[6];
                    float4 b0[6], b1[6];
                    u32x4 rv[6];
                    int64_t off[6];
                    bool ok[6];
#pragma unroll
                    for (int it = 0; it < 6; ++it) {
                        const int e = it * 64 + lane, px = e / 12, ch = e - px * 12;
                        const int n = cur.n0 + ch * 8;
                        lo[it] = *reinterpret_cast<const f32x4*>(stg + px * kStgPx + ch * 32);
                        hi[it] = *reinterpret_cast<const f32x4*>(stg + px * kStgPx + ch * 32 + 16);
                        ok[it] = h < g.H_out && cur.w0 + px < g.W_out;
                        off[it] = ok[it] ? (orow0 + px) * g.ldo + n : 0;          // out of range: a valid address, not stored
                        if constexpr (HAS_BIAS) {
                            b0[it] = *reinterpret_cast<const float4*>(g.bias + n);
                            b1[it] = *reinterpret_cast<const float4*>(g.bias + n + 4);
                        }
                        if constexpr (HAS_RESID) rv[it] = *reinterpret_cast<const u32x4*>(g.resid + off[it]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int it = 0; it < 6; ++it) {
                        float v[8] = {lo[it][0], lo[it][1], lo[it][2], lo[it][3], hi[it][0], hi[it][1], hi[it][2], hi[it][3]};
                        if constexpr (HAS_BIAS) {
                            v[0] += b0[it].x; v[1] += b0[it].y; v[2] += b0[it].z; v[3] += b0[it].w;
                            v[4] += b1[it].x; v[5] += b1[it].y; v[6] += b1[it].z; v[7] += b1[it].w;
                        }
                        if constexpr (HAS_RESID) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) { v[2 * k] += bf16lo_to_f32(rv[it][k]); v[2 * k + 1] += bf16hi_to_f32(rv[it][k]); }
                        }
                        if (ok[it])
                            *reinterpret_cast<u32x4*>(g.out + off[it]) =
                                u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                    }
                }
            };
            if (g.bias) {
                if (g.resid) write_rows(std::true_type{}, std::true_type{});
                else write_rows(std::true_type{}, std::false_type{});
            } else {
                if (g.resid) write_rows(std::false_type{}, std::true_type{});
                else write_rows(std::false_type{}, std::false_type{});
            }
        }
        cur = nxt;
        wcur = wnxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail's dummy slices and patch pieces still write LDS
}

wan_status_t launch_conv3_patch(ConvArgs g, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};
    const wan_status_t st = wan_once_per_device(attr_done, +[]() -> wan_status_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_patch_kernel<32>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kPatchLds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_patch_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, kPatchLds);
        if (e != hipSuccess) {
            wan_set_error("wan_conv_cl: cannot reserve %d B of LDS: %s", kPatchLds, hipGetErrorString(e));
            return WAN_ERR_LAUNCH;
        }
        return WAN_OK;
    });
    if (st != WAN_OK) return st;
    g.pth = (g.H_out + PT_H - 1) / PT_H;
    g.ptw = (g.W_out + PT_W - 1) / PT_W;
    g.tiles_n = (g.Cout + 95) / 96;
    const int64_t ntiles = (int64_t)g.T_out * g.pth * g.ptw * g.tiles_n;
    static std::atomic<int> cus[64];                       // compute units per device (one persistent workgroup each)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    int ncu = cus[dev].load(std::memory_order_relaxed);
    if (ncu == 0) {
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        cus[dev].store(ncu, std::memory_order_relaxed);
    }
    const int64_t nwg = ntiles < ncu ? ntiles : ncu;
    // conv_mfma: 0 (default) = by the PER-FRAME shape -- the 16x16x32 form when a chunk of four frames fills the chip (>= one tile
    // per CU: +0.4 ... +3.4 % on the decoder / encoder stages at 480p, sustained clock 1.58 -> 1.78 GHz at 57 -> 52 % matrix-pipe
    // occupancy), the 32x32x16 form for small planes; 32 / 16 force one (profiles/r05/vae_conv_mfma16_ab.log).  The rule must
    // not look at T_out: the two forms add the 32 channels of a tap in different groupings, so they differ in the last bf16 bit,
    // and a decode in chunks of 1, 2, 4 ... frames has to give the same bits (wan_vae.py decode(); the one-latent-frame launch of
    // a 60 x 104 plane pays 7 % for that, 0.082 -> 0.088 ms)
    const int mi = wan_tune(WAN_TUNE_CONV_MFMA);
    const int64_t tiles_frame = (int64_t)g.pth * g.ptw * g.tiles_n;
    if (mi == 16 || (mi == 0 && tiles_frame * 4 >= ncu)) hipLaunchKernelGGL(conv3_patch_kernel<16>, dim3((unsigned)nwg), dim3(kThreads), kPatchLds, s, g);
    else hipLaunchKernelGGL(conv3_patch_kernel<32>, dim3((unsigned)nwg), dim3(kThreads), kPatchLds, s, g);
    WAN_CHECK_LAUNCH("wan_conv_cl");
    return WAN_OK;
}

// ------------------------------------------------------------------ causal 3x3x3, stride 1, <= 4 output channels: the decoder head
// decoder.head.2 maps 96 channels to 3 at FULL resolution (wan_vae.py:476): in conv_cl_kernel it fills 3 of the 32 columns of the
// narrowest tile and pays the whole per-chunk gather arithmetic (0.56 ms per output frame, 13 % of a decode).  Here a workgroup
// owns 8 x 32 output pixels of one frame, stages their (3 frames) x 10 x 34 input patch in LDS 32 channels at a time (the same
// pixel-major, XOR-swizzled 64-byte records as conv3_patch_kernel) together with the chunk's 27 x COUT weight rows: with
// v_mfma_f32_16x16x32_bf16 a weight fragment is 16 output channels (COUT real, the rest one shared zero row) x 32 input channels
// = 16 bytes per lane, so one tap of 64 pixels is ONE weight ds_read_b128, four pixel ds_read_b128 and four MFMAs -- no global
// address arithmetic and no weight traffic in the loop.
// (A vector-ALU form -- one pixel per thread, v_dot2c_f32_bf16 against scalar-loaded weights -- was measured first and LOST to
// the gather kernel, 0.337 vs 0.328 s per decode: 100 scalar instructions and an exposed s_load round trip per tap.)
// Workgroup w runs on XCD w & 7 and the frames of one spatial tile are consecutive on ONE XCD, so the 3x temporal re-read of
// every input plane is served by that XCD's L2.
constexpr int HD_H = 8, HD_W = 32, HD_PH = HD_H + 2, HD_PW = HD_W + 2, HD_FR = HD_PH * HD_PW, HD_PX = 3 * HD_FR;   // 1020
constexpr int kHeadLds = 1024 * 64 + 27 * 4 * 64 + 64;

template <int COUT>
__global__ __launch_bounds__(256, 2) void conv3_head_kernel(ConvArgs g) {
    static_assert(COUT <= 16, "one 16-column MFMA tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = g.pth * g.ptw;
    int tile, frame;
    {
        // XCD x walks tiles x, x + 8, ...: all frames of one tile back to back on one XCD
        const int w = blockIdx.x, xcd = w & 7, sq = w >> 3;
        const int ti = sq / g.T_out;
        frame = sq - ti * g.T_out;
        tile = ti * 8 + xcd;
        if (tile >= ntile) return;
    }
    const int thi = tile / g.ptw, twi = tile - thi * g.ptw;
    const int h0 = thi * HD_H, w0 = twi * HD_W;
    const int hf = g.hist_frames;
    const int H = g.H_in, W = g.W_in, Cin = g.Cin;
    // ---- staging plan (chunk-independent): this thread's i-th piece = patch pixel pp, 16-byte slot q
    int src_off[16];            // element offset of (pixel, 8-channel piece) inside x or hist, -1 = zero fill
    unsigned hist_mask = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int idx = tid + i * 256;                          // 0 .. 4095 (>= 4080: zero tail)
        const int pp = idx >> 2, q = idx & 3;
        const int f = pp / HD_FR, rem = pp - f * HD_FR;
        const int pr = rem / HD_PW, pc = rem - pr * HD_PW;
        const int t = frame - 2 + f, hh = h0 - 1 + pr, ww = w0 - 1 + pc;
        const bool ok = pp < HD_PX && hh >= 0 && hh < H && ww >= 0 && ww < W && t >= -hf;
        if (t < 0) hist_mask |= 1u << i;
        const int tt = t < 0 ? t + hf : t;
        src_off[i] = ok ? ((tt * H + hh) * W + ww) * Cin + q * 8 : -1;
    }
    // ---- MFMA roles: A = 16 pixels x 32 channels (lane: pixel lane & 15, channels 8 * (lane >> 4) ..), B = 32 channels x 16
    // output channels (lane: output channel lane & 15, same channel octet).  Wave `wid` owns tile rows 2 wid, 2 wid + 1 as four
    // 16-pixel blocks mb: row 2 wid + (mb >> 1), columns 16 (mb & 1) ..
    const int m = lane & 15, kq = lane >> 4;
    char* const wlds = smem + 1024 * 64;
    const int wrd = (m < COUT ? m * 64 : 27 * COUT * 64) + kq * 16;      // this lane's row of a tap's weight record (or the zero row)
    f32x4 acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nch = Cin >> 5;
    for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();                                        // the previous chunk's reads are done
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = tid + i * 256;
            const int pp = idx >> 2, q = idx & 3;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (src_off[i] >= 0) {
                if ((hist_mask >> i) & 1) v = *reinterpret_cast<const u32x4*>(g.hist + (src_off[i] + ch * 32));
                else v = *reinterpret_cast<const u32x4*>(g.x + (src_off[i] + ch * 32));
            }
            *reinterpret_cast<u32x4*>(smem + pp * 64 + ((q ^ ((pp >> 2) & 3)) << 4)) = v;
        }
        // the chunk's weights: 27 taps x COUT rows x 32 channels (64 B per row) behind the patch; one 64-byte zero row serves the
        // 16 - COUT padding columns of the MFMA tile
        for (int i = tid; i < 27 * COUT * 4; i += 256) {
            const int tap = i / (COUT * 4), rem = i - tap * (COUT * 4), co = rem >> 2, q = rem & 3;
            *reinterpret_cast<u32x4*>(wlds + i * 16) = *reinterpret_cast<const u32x4*>(g.w + (co * (int)g.ldw + tap * Cin + ch * 32 + q * 8));
        }
        if (tid < 4) *reinterpret_cast<u32x4*>(wlds + 27 * COUT * 64 + tid * 16) = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const bf16x8 wfrag = *reinterpret_cast<const bf16x8*>(wlds + wrd + ((kt * 3 + kh) * 3 + kw) * (m < COUT ? COUT * 64 : 0));
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        const int pp = (kt * HD_PH + 2 * wid + (mb >> 1) + kh) * HD_PW + 16 * (mb & 1) + m + kw;
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(smem + pp * 64 + ((kq ^ ((pp >> 2) & 3)) << 4));
                        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wfrag, acc[mb], 0, 0, 0);
                    }
                }
    }
    // ---- D: lane holds output channel lane & 15 of pixels 4 * (lane >> 4) .. + 3 of each block
    if (m < COUT) {
        const float bv = g.bias ? g.bias[m] : 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int ho = h0 + 2 * wid + (mb >> 1);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int wo = w0 + 16 * (mb & 1) + 4 * kq + r4;
                if (ho < g.H_out && wo < g.W_out)
                    g.out[((int64_t)(frame * g.H_out + ho) * g.W_out + wo) * g.ldo + m] = (bf16_t)(acc[mb][r4] + bv);
            }
        }
    }
}

wan_status_t launch_conv3_head(ConvArgs& g, hipStream_t s) {
    static std::atomic<uint64_t> done{0};
    const wan_status_t st = wan_once_per_device(done, +[]() -> wan_status_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_head_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, kHeadLds);
        if (e != hipSuccess) { wan_set_error("wan_conv_cl: cannot reserve LDS: %s", hipGetErrorString(e)); return WAN_ERR_LAUNCH; }
        return WAN_OK;
    });
    if (st != WAN_OK) return st;
    g.pth = (g.H_out + HD_H - 1) / HD_H; g.ptw = (g.W_out + HD_W - 1) / HD_W;
    const int ntile = g.pth * g.ptw, per = (ntile + 7) / 8;
    const int64_t nwg = (int64_t)per * 8 * g.T_out;
    hipLaunchKernelGGL(conv3_head_kernel<4>, dim3((unsigned)nwg), dim3(256), kHeadLds, s, g);
    WAN_CHECK_LAUNCH("wan_conv_cl (head)");
    return WAN_OK;
}

// ------------------------------------------------------------------ per-pixel RMS_norm (+SiLU)
// F.normalize(x, dim=channel) * sqrt(C) * gamma  (wan_vae.py:43-58), optional SiLU.  A pixel's C channels are
// C/8 16-byte chunks; LPP = 8/16/32/64 lanes share a pixel (the next power of two >= C/8), so a wave normalises
// 64/LPP pixels at once (C = 96 at full resolution: 4 pixels per wave, 12 of every 16 lanes active instead of 12
// of 64) and each wave walks PIX_ITERS pixel groups to keep more bytes in flight.
template <int LPP>
__global__ __launch_bounds__(256) void rmsnorm_silu_cl_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                              bf16_t* __restrict__ out, int64_t rows, int C, int silu) {
    constexpr int PPW = 64 / LPP;            // pixels per wave and step
    constexpr int PIX_ITERS = 4;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (LPP - 1);        // chunk index inside the pixel
    const int pw = lane / LPP;               // pixel slot inside the wave
    const int nchunk = C >> 3;
    const bool act_c = sub < nchunk;
    float gv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (act_c) {
        const float4 ga = reinterpret_cast<const float4*>(gamma)[sub * 2], gb = reinterpret_cast<const float4*>(gamma)[sub * 2 + 1];
        gv[0] = ga.x; gv[1] = ga.y; gv[2] = ga.z; gv[3] = ga.w; gv[4] = gb.x; gv[5] = gb.y; gv[6] = gb.z; gv[7] = gb.w;
    }
    const float sqrt_c = sqrtf((float)C);
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t row0 = wave * (PPW * PIX_ITERS) + pw;
    u32x4 v[PIX_ITERS];
#pragma unroll
    for (int it = 0; it < PIX_ITERS; ++it) {
        const int64_t row = row0 + (int64_t)it * PPW;
        v[it] = u32x4{0, 0, 0, 0};
        if (act_c && row < rows) v[it] = reinterpret_cast<const u32x4*>(x + row * C)[sub];
    }
#pragma unroll
    for (int it = 0; it < PIX_ITERS; ++it) {
        const int64_t row = row0 + (int64_t)it * PPW;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float a = bf16lo_to_f32(v[it][j]), b = bf16hi_to_f32(v[it][j]); ss += a * a + b * b; }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);      // stays inside the LPP-lane group
        const float scale = sqrt_c / fmaxf(sqrtf(ss), 1e-12f);
        if (act_c && row < rows) {
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = bf16lo_to_f32(v[it][j]) * scale * gv[2 * j], b = bf16hi_to_f32(v[it][j]) * scale * gv[2 * j + 1];
                if (silu) { a = a / (1.f + __expf(-a)); b = b / (1.f + __expf(-b)); }
                o[j] = pack_bf16x2(a, b);
            }
            reinterpret_cast<u32x4*>(out + row * C)[sub] = o;
        }
    }
}

// ------------------------------------------------------------------ row softmax for the VAE's single-head attention
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, int64_t lds_, bf16_t* __restrict__ p,
                                                           int64_t ldp, int n, int npad, float scale) {
    __shared__ float red[4];
    const float* sr = s + (int64_t)blockIdx.x * lds_;
    bf16_t* pr = p + (int64_t)blockIdx.x * ldp;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, sr[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) sum += __expf((sr[i] - mx) * scale);
    sum = block_sum<4>(sum, red);
    const float inv = 1.f / sum;
    for (int i = threadIdx.x; i < npad; i += 256)
        pr[i] = (bf16_t)(i < n ? __expf((sr[i] - mx) * scale) * inv : 0.f);
}

// ------------------------------------------------------------------ layout converters at the VAE boundary
template <typename T>
__global__ __launch_bounds__(256) void video_to_cl_kernel(const T* __restrict__ v, bf16_t* __restrict__ out, int Cv, int Cpad,
                                                          int64_t npix) {
    // [Cv, npix] planar -> [npix, Cpad] channels-last, extra channels zero
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256)
        for (int c = 0; c < Cpad; ++c) out[i * Cpad + c] = (bf16_t)(c < Cv ? (float)v[(int64_t)c * npix + i] : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void cl_to_video_kernel(const bf16_t* __restrict__ x, int64_t ld, T* __restrict__ out, int Cv,
                                                          int64_t npix, int clamp) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256)
        for (int c = 0; c < Cv; ++c) {
            float f = (float)x[i * ld + c];
            if (clamp) f = fminf(fmaxf(f, -1.f), 1.f);
            out[(int64_t)c * npix + i] = (T)f;
        }
}

}  // namespace

extern "C" wan_status_t wan_conv_cl(const void* x, const void* hist, int hist_frames, const void* w, int64_t ldw,
                                    const float* bias, const void* resid, void* out, int64_t ldo,
                                    const wan_conv_params* p, void* stream) {
    WAN_REQUIRE(x && w && out && p, WAN_ERR_INVALID, "wan_conv_cl: null tensor");
    WAN_REQUIRE(p->Cin > 0 && p->Cin % 8 == 0, WAN_ERR_UNSUPPORTED, "wan_conv_cl: Cin=%d must be a multiple of 8", p->Cin);
    WAN_REQUIRE(p->Cout > 0 && p->Cout % 4 == 0, WAN_ERR_UNSUPPORTED, "wan_conv_cl: Cout=%d must be a multiple of 4", p->Cout);
    WAN_REQUIRE((p->KT == 1 || p->KT == 3) && (p->KH == 1 || p->KH == 3) && (p->KW == 1 || p->KW == 3), WAN_ERR_UNSUPPORTED,
                "wan_conv_cl: kernel (%d,%d,%d) (extents must be 1 or 3)", p->KT, p->KH, p->KW);
    WAN_REQUIRE(p->T_in > 0 && p->H_in > 0 && p->W_in > 0 && p->T_out > 0 && p->H_out > 0 && p->W_out > 0, WAN_ERR_INVALID,
                "wan_conv_cl: bad extents");
    WAN_REQUIRE(p->st > 0 && p->sh > 0 && p->sw > 0 && p->pt >= 0 && p->ph >= 0 && p->pw >= 0 && p->pt < 4096 && p->ph < 4096,
                WAN_ERR_INVALID, "wan_conv_cl: bad stride/pad");
    WAN_REQUIRE(p->H_out * p->sh + 8 < 28000 && p->W_out * p->sw + 8 < 28000, WAN_ERR_UNSUPPORTED,
                "wan_conv_cl: spatial extent too large for the packed coordinates");
    WAN_REQUIRE(hist_frames >= 0 && hist_frames <= 2 && (hist_frames == 0 || hist != nullptr), WAN_ERR_INVALID,
                "wan_conv_cl: hist_frames=%d needs a history buffer", hist_frames);
    const int ntaps = p->KT * p->KH * p->KW;
    const int K = ntaps * p->Cin;
    const int Kpad = (K + BK - 1) / BK * BK;
    WAN_REQUIRE(ldw >= Kpad && ldw % 8 == 0, WAN_ERR_INVALID, "wan_conv_cl: ldw=%lld must be >= roundup(K=%d,64) and %%8", (long long)ldw, K);
    WAN_REQUIRE(!p->time_interleave || (p->Cout % 8 == 0), WAN_ERR_INVALID, "wan_conv_cl: interleave needs even channel halves %%4");
    const int64_t M = (int64_t)p->T_out * p->H_out * p->W_out;
    WAN_REQUIRE(M < (1LL << 31), WAN_ERR_UNSUPPORTED, "wan_conv_cl: too many output pixels");
    WAN_REQUIRE(ldo % 4 == 0 && ldo >= (p->time_interleave ? p->Cout / 2 : p->Cout), WAN_ERR_INVALID, "wan_conv_cl: ldo=%lld", (long long)ldo);
    ConvArgs g;
    g.x = (const bf16_t*)x; g.hist = (const bf16_t*)hist; g.w = (const bf16_t*)w; g.ldw = ldw; g.bias = bias;
    g.resid = (const bf16_t*)resid; g.out = (bf16_t*)out; g.ldo = ldo;
    g.T_in = p->T_in; g.H_in = p->H_in; g.W_in = p->W_in; g.Cin = p->Cin;
    g.T_out = p->T_out; g.H_out = p->H_out; g.W_out = p->W_out; g.Cout = p->Cout;
    g.KT = p->KT; g.KH = p->KH; g.KW = p->KW; g.st = p->st; g.sh = p->sh; g.sw = p->sw; g.pt = p->pt; g.ph = p->ph; g.pw = p->pw;
    g.ups = p->upsample2x ? 1 : 0; g.interleave = p->time_interleave ? 1 : 0; g.hist_frames = hist_frames; g.silu = 0;
    g.M = (int)M; g.ntaps = ntaps; g.nk = Kpad / BK;
    g.cin_magic = (unsigned)((0x100000000ULL + p->Cin - 1) / p->Cin);
    g.div_w_mul = p->KW == 3 ? 171 : 1; g.div_w_sh = p->KW == 3 ? 9 : 0;
    g.div_h_mul = p->KH == 3 ? 171 : 1; g.div_h_sh = p->KH == 3 ? 9 : 0;
    g.xcd_slabs = wan_tune(WAN_TUNE_CONV_XCD) != 0;      // developer A/B switch (wan_set_tuning)
    g.tiles_m = (g.M + BM - 1) / BM;
    hipStream_t s = (hipStream_t)stream;
    {   // causal 3x3x3 / stride 1 with whole 96-channel output slices: the LDS-patch kernel
        const int64_t px = (int64_t)(p->T_in + hist_frames) * p->H_in * p->W_in;
        const int mode = wan_tune(WAN_TUNE_CONV_PATCH);
        const bool shape_ok = ntaps == 27 && p->st == 1 && p->sh == 1 && p->sw == 1 && p->pt == 2 && p->ph == 1 && p->pw == 1 &&
                              !p->upsample2x && !p->time_interleave && p->Cin % 32 == 0 && p->Cout % 96 == 0 &&
                              p->T_out == p->T_in && p->H_out == p->H_in && p->W_out == p->W_in &&
                              px < (1 << 24) && px * p->Cin < (1LL << 31) && ldo % 8 == 0 &&          // element offsets in int
                              ((uintptr_t)out & 15) == 0 && ((uintptr_t)resid & 15) == 0;          // 16-byte output stores
        if (mode && shape_ok) return launch_conv3_patch(g, s);      // also with fewer tiles than CUs (1 latent frame: 544 vs 285 TFLOP/s)
    }
    {   // causal 3x3x3 / stride 1 with <= 4 output channels (the decoder head): direct convolution on the vector ALUs
        const int64_t px = (int64_t)(p->T_in + hist_frames) * p->H_in * p->W_in;
        const bool shape_ok = ntaps == 27 && p->st == 1 && p->sh == 1 && p->sw == 1 && p->pt == 2 && p->ph == 1 && p->pw == 1 &&
                              !p->upsample2x && !p->time_interleave && p->Cin % 32 == 0 && p->Cout == 4 && resid == nullptr &&
                              p->T_out == p->T_in && p->H_out == p->H_in && p->W_out == p->W_in && px * p->Cin < (1LL << 31) &&
                              ldw % 2 == 0 && ((uintptr_t)out & 7) == 0 && ldo % 4 == 0;
        if (wan_tune(WAN_TUNE_CONV_HEAD) != 0 && shape_ok) return launch_conv3_head(g, s);
    }
    if (p->Cout <= 32) { g.tiles_n = 1; return launch_conv<1>(g, s); }      // latent convs (16 / 32), the head when the switch is off
    if (p->Cout <= 96) { g.tiles_n = (p->Cout + 95) / 96; return launch_conv<3>(g, s); }
    if (p->Cout % 192 == 0) { g.tiles_n = p->Cout / 192; return launch_conv<6>(g, s); }
    g.tiles_n = (p->Cout + 127) / 128;
    return launch_conv<4>(g, s);
}

extern "C" wan_status_t wan_rmsnorm_silu_cl(const void* x, const float* gamma, void* out, int64_t rows, int C, int silu,
                                            void* stream) {
    WAN_REQUIRE(x && gamma && out, WAN_ERR_INVALID, "wan_rmsnorm_silu_cl: null tensor");
    WAN_REQUIRE(C > 0 && C % 8 == 0 && C <= 512, WAN_ERR_UNSUPPORTED, "wan_rmsnorm_silu_cl: C=%d (multiple of 8, <= 512)", C);
    if (rows <= 0) return WAN_OK;
    const int nchunk = C / 8;
    const int lpp = nchunk <= 8 ? 8 : (nchunk <= 16 ? 16 : (nchunk <= 32 ? 32 : 64));
    const int64_t rows_per_wg = 4 * (64 / lpp) * 4;                 // 4 waves x pixels per wave x PIX_ITERS
    const dim3 grid((unsigned)((rows + rows_per_wg - 1) / rows_per_wg)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* xp = (const bf16_t*)x;
    bf16_t* op = (bf16_t*)out;
    switch (lpp) {
        case 8: hipLaunchKernelGGL(rmsnorm_silu_cl_kernel<8>, grid, block, 0, st, xp, gamma, op, rows, C, silu); break;
        case 16: hipLaunchKernelGGL(rmsnorm_silu_cl_kernel<16>, grid, block, 0, st, xp, gamma, op, rows, C, silu); break;
        case 32: hipLaunchKernelGGL(rmsnorm_silu_cl_kernel<32>, grid, block, 0, st, xp, gamma, op, rows, C, silu); break;
        default: hipLaunchKernelGGL(rmsnorm_silu_cl_kernel<64>, grid, block, 0, st, xp, gamma, op, rows, C, silu); break;
    }
    WAN_CHECK_LAUNCH("wan_rmsnorm_silu_cl");
    return WAN_OK;
}

extern "C" wan_status_t wan_softmax_rows(const float* scores, int64_t lds, void* probs_bf16, int64_t ldp, int64_t rows,
                                         int n, int npad, float scale, void* stream) {
    WAN_REQUIRE(scores && probs_bf16, WAN_ERR_INVALID, "wan_softmax_rows: null tensor");
    WAN_REQUIRE(n > 0 && npad >= n && lds >= n && ldp >= npad && rows >= 0, WAN_ERR_INVALID, "wan_softmax_rows: n=%d npad=%d", n, npad);
    if (rows == 0) return WAN_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, scores, lds,
                       (bf16_t*)probs_bf16, ldp, n, npad, scale);
    WAN_CHECK_LAUNCH("wan_softmax_rows");
    return WAN_OK;
}

extern "C" wan_status_t wan_video_to_cl(const void* video, int in_dtype, void* out_bf16, int Cv, int Cpad, int64_t npix,
                                        void* stream) {
    WAN_REQUIRE(video && out_bf16, WAN_ERR_INVALID, "wan_video_to_cl: null tensor");
    WAN_REQUIRE(Cv > 0 && Cpad >= Cv && Cpad % 8 == 0 && npix >= 0, WAN_ERR_INVALID, "wan_video_to_cl: Cv=%d Cpad=%d", Cv, Cpad);
    WAN_REQUIRE(in_dtype == 0 || in_dtype == 1, WAN_ERR_INVALID, "wan_video_to_cl: in_dtype=%d", in_dtype);
    if (npix == 0) return WAN_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>((npix + 255) / 256, 8192);
    if (in_dtype == 0)
        hipLaunchKernelGGL(video_to_cl_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)video,
                           (bf16_t*)out_bf16, Cv, Cpad, npix);
    else
        hipLaunchKernelGGL(video_to_cl_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)video,
                           (bf16_t*)out_bf16, Cv, Cpad, npix);
    WAN_CHECK_LAUNCH("wan_video_to_cl");
    return WAN_OK;
}

extern "C" wan_status_t wan_cl_to_video(const void* x_bf16, int64_t ld, void* out, int out_dtype, int Cv, int64_t npix,
                                        int clamp, void* stream) {
    WAN_REQUIRE(x_bf16 && out, WAN_ERR_INVALID, "wan_cl_to_video: null tensor");
    WAN_REQUIRE(Cv > 0 && ld >= Cv && npix >= 0, WAN_ERR_INVALID, "wan_cl_to_video: Cv=%d ld=%lld", Cv, (long long)ld);
    WAN_REQUIRE(out_dtype == 0 || out_dtype == 1, WAN_ERR_INVALID, "wan_cl_to_video: out_dtype=%d", out_dtype);
    if (npix == 0) return WAN_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>((npix + 255) / 256, 8192);
    if (out_dtype == 0)
        hipLaunchKernelGGL(cl_to_video_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, ld,
                           (float*)out, Cv, npix, clamp);
    else
        hipLaunchKernelGGL(cl_to_video_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, ld,
                           (bf16_t*)out, Cv, npix, clamp);
    WAN_CHECK_LAUNCH("wan_cl_to_video");
    return WAN_OK;
}
