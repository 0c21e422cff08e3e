// nn.Linear on the gfx950 matrix cores:  acc[m,n] = sum_k A[m,k] * W[n,k]  (bf16 in, fp32 accumulate)
//
// Tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves (2x2), each wave 64x64 as 4x4
// v_mfma_f32_16x16x32_bf16 tiles.  Both operands are K-contiguous ([rows][K]) so A and W tiles
// share one LDS image: row-major [128][64] bf16 (128-byte rows, eight 16-byte chunks).
//
// Staging is LDS-DMA (global_load_lds_dwordx4): one wave instruction lands 8 rows x 128 B lane-
// linearly, so the bank swizzle is applied to the per-lane SOURCE address and mirrored on the
// ds_read_b128 side (physical chunk = logical chunk ^ ((row >> 1) & 7), conflict-free for the
// 16-row MFMA fragment read).  Two LDS buffers: tile t+1 streams in while tile t is multiplied.
//
// MFMA operand order is chosen per epilogue so that every lane owns 4 CONSECUTIVE output elements
// in the output's contiguous dimension:
//   row-major epilogues:  D = mfma(Wfrag, Afrag) -> lane holds m = l&15, n = (l>>4)*4 + r
//   transposed epilogue:  D = mfma(Afrag, Wfrag) -> lane holds n = l&15, m = (l>>4)*4 + r
//
// Workgroup -> tile mapping is XCD-aware: block b runs on XCD b%8 (observed, speed only); each XCD
// walks a contiguous slab of the tile sequence ordered as 8(M) x all(N) groups so that the 64 tiles
// resident on an XCD share 8 A-panels and 8 W-panels through its private L2.
#include <stdlib.h>

#include <algorithm>

#include "common.hpp"

// gemm_bf16_pk.hip: the persistent stream-K form of the 4-wave 256 x 256 kernel (callers that bring a workspace)
wan_status_t wan_gemm_bf16_pk(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                              void* out, int64_t ldo, int M, int N, int K, int epilogue,
                              const float* gate, int64_t rows_per_batch, void* workspace, hipStream_t s);
int64_t wan_gemm_pk_workspace_bytes(int M, int N);
wan_status_t wan_gemm_fp8_pk(const void* A, int64_t lda, const float* a_row_scale, const void* W, int64_t ldw, const float* w_row_scale,
                             const float* bias, void* out, int64_t ldo, int M, int N, int K, int epilogue,
                             const float* gate, int64_t rows_per_batch, void* workspace, hipStream_t s);
// gemm_bf16_256.hip: the 256 x 256 phased kernel used for large shapes
wan_status_t wan_gemm_bf16_256(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                               void* out, int64_t ldo, int M, int N, int K, int epilogue,
                               const float* gate, int64_t rows_per_batch, hipStream_t s);

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kThreads = 256;
constexpr int kTileBytes = BM * BK * 2;        // 16 KiB per operand tile
constexpr int kStageBytes = 2 * kTileBytes;    // A + W
constexpr int kLdsBytes = 2 * kStageBytes;     // double buffered: 64 KiB

struct GemmArgs {
    const bf16_t* A; int64_t lda;
    const bf16_t* W; int64_t ldw;
    const float* bias;
    void* out; int64_t ldo;
    const float* gate; int64_t rows_per_batch;
    int M, N, K;
    int tiles_m, tiles_n;
    int64_t sA, sW, sO;      // element strides between the problems of a batched launch (blockIdx.y)
    // split-K form (SPLITK instantiation, round 5): blockIdx.y = split index s of `splitk`; a workgroup multiplies K tiles
    // [s * nk / splitk, (s + 1) * nk / splitk) of its output tile, stores its fp32 accumulators to its slot of the caller's
    // workspace and takes a ticket on the tile's arrival counter; the LAST arriver adds the pieces IN SPLIT ORDER (its own from
    // registers) and runs the normal epilogue -- bitwise reproducible, nobody waits for anybody (as in gemm_bf16_pk.hip).
    int splitk;
    int* counters;           // workspace head: one arrival counter per output tile (zeroed by a memset node ahead of the launch)
    char* slots;             // behind the counters: [tile][split] slots of 128 x 128 fp32, register-major
};

__device__ __forceinline__ void tile_coords(const GemmArgs& g, int& tm, int& tn) {
    // bijective XCD remap (guide T1): XCD x gets tiles [start_x, start_x + cnt_x)
    const int nwg = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    // grouped order: 8 M-tiles x all N-tiles per group, M fastest inside the group
    constexpr int GM = 8;
    const int per_group = GM * g.tiles_n;
    const int grp = t / per_group;
    const int first_m = grp * GM;
    const int gm = min(GM, g.tiles_m - first_m);
    const int in = t - grp * per_group;
    tm = first_m + in % gm;
    tn = in / gm;
}

template <int EPI, bool SPLITK = false>
__global__ __launch_bounds__(kThreads, 2) void gemm_bf16_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool kTransposed = (EPI == WAN_EPI_BF16_T);

    int tm, tn;
    tile_coords(g, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    if (!SPLITK && gridDim.y > 1) {         // wan_gemm_bf16_batched: problem blockIdx.y
        g.A += blockIdx.y * g.sA;
        g.W += blockIdx.y * g.sW;
        g.out = (char*)g.out + blockIdx.y * g.sO * ((EPI == WAN_EPI_F32 || EPI == WAN_EPI_RESID_F32) ? 4 : 2);
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;

    // ---- staging addresses: wave `wid` DMA-copies pieces wid*4 .. wid*4+3 (8 rows each) of A and of W
    const int srow = lane >> 3;                 // row inside an 8-row piece
    const int spc = lane & 7;                   // physical 16-byte chunk
    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wid * 4 + j) * 8 + srow;           // tile row 0..127
        const int c = spc ^ ((row >> 1) & 7);               // logical chunk this lane must fetch
        const int am = min(m0 + row, g.M - 1);
        const int wn = min(n0 + row, g.N - 1);
        a_src[j] = g.A + (int64_t)am * g.lda + c * 8;
        w_src[j] = g.W + (int64_t)wn * g.ldw + c * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * kStageBytes;
        const int koff = kt * BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(a_src[j] + koff, base + (wid * 4 + j) * 1024);
            glds16(w_src[j] + koff, base + kTileBytes + (wid * 4 + j) * 1024);
        }
    };

    // ---- fragment read offsets (bytes inside a tile)
    const int frow = lane & 15;
    const int kg = lane >> 4;                   // k-group: 8 contiguous k per lane
    const int sw = (lane >> 1) & 7;             // == ((row >> 1) & 7) for every fragment row of this lane
    const int off_k0 = frow * 128 + ((kg ^ sw) << 4);            // logical chunk kg      (kk = 0)
    const int off_k1 = frow * 128 + (((kg + 4) ^ sw) << 4);      // logical chunk kg + 4  (kk = 1)
    const int a_base = wr * 64 * 128;
    const int w_base = kTileBytes + wc * 64 * 128;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_all = g.K / BK;
    const int split = SPLITK ? (int)blockIdx.y : 0;
    const int kt0 = SPLITK ? (int)((int64_t)split * nk_all / g.splitk) : 0;
    const int nk = SPLITK ? (int)((int64_t)(split + 1) * nk_all / g.splitk) : nk_all;
    stage(0, kt0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): tile 0 landed
    __syncthreads();

    for (int kt = kt0; kt < nk; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sb = smem + cur * kStageBytes;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int off = kk ? off_k1 : off_k0;
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(sb + a_base + i * 2048 + off);
                wf[i] = *reinterpret_cast<const bf16x8*>(sb + w_base + i * 2048 + off);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (kTransposed)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
                }
        }
        __builtin_amdgcn_s_waitcnt(0);   // next tile landed (vmcnt) + our ds_reads retired
        __syncthreads();
    }

    if constexpr (SPLITK) {
        // publish my piece with write-through (sc1) stores (no L2 write-back needed to make it visible: CDNA4 guide, "publish-large"),
        // drain them, take a ticket; only the last arriver of the tile goes on
        const int tile = tm * g.tiles_n + tn;
        char* const tile_slots = g.slots + (int64_t)tile * g.splitk * (BM * BN * 4);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(tile_slots + (int64_t)split * (BM * BN * 4)), 0, BM * BN * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, ((i * 4 + j) * kThreads + tid) * 16, 0, /*sc1*/ 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        volatile int* const lds_flag = reinterpret_cast<volatile int*>(smem);
        if (tid == 0) lds_flag[0] = __hip_atomic_fetch_add(g.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(lds_flag[0]) != g.splitk - 1) return;
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        f32x4 sum[4][4];
        for (int sidx = 0; sidx < g.splitk; ++sidx) {            // in split order, whoever arrived last
            const f32x4* src = reinterpret_cast<const f32x4*>(tile_slots + (int64_t)sidx * (BM * BN * 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 t = sidx == split ? acc[i][j] : src[(i * 4 + j) * kThreads + tid];
                    sum[i][j] = sidx == 0 ? t : sum[i][j] + t;
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = sum[i][j];
    }

    // ---- epilogue
    const int l15 = lane & 15, l4 = (lane >> 4) * 4;
    if constexpr (!kTransposed) {
        // The bias of a column group is loaded once; the fp32 read-modify-write epilogue reads the residual stream (and the gate
        // rows) of two row groups back to back and waits once (element by element the compiler emitted load / wait / store per
        // 4-element group).  Out-of-range rows / columns read a clamped (valid) address and are not stored.
        int nn[4];
        bool nok[4];
        float4 bj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + l4;      // N % 4 == 0 -> whole 4-group in or out
            nok[j] = n < g.N;
            nn[j] = nok[j] ? n : 0;
            bj[j] = g.bias ? *reinterpret_cast<const float4*>(g.bias + nn[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int rpb = g.gate ? (int)g.rows_per_batch : 1;
#pragma unroll
        for (int ig = 0; ig < 4; ig += 2) {
            int mm[2];
            bool mok[2];
            float4 xr[2][4], gv[2][4];
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int m = m0 + wr * 64 + (ig + ii) * 16 + l15;
                mok[ii] = m < g.M;
                mm[ii] = mok[ii] ? m : g.M - 1;
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
                    } else {   // WAN_EPI_RESID_F32
                        const float4 x = xr[ii][j], gq = gv[ii][j];
                        *reinterpret_cast<float4*>((float*)g.out + off) =
                            make_float4(x.x + v[0] * gq.x, x.y + v[1] * gq.y, x.z + v[2] * gq.z, x.w + v[3] * gq.w);
                    }
                }
        }
    } else {
        // out[n, m..m+3]: lane holds n = l&15, m = (l>>4)*4 + r
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + l15;
            if (n >= g.N) continue;
            const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wr * 64 + i * 16 + l4;
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

template <int EPI>
wan_status_t launch(const GemmArgs& g, hipStream_t s, int batch = 1) {
    static std::atomic<uint64_t> attr_done{0};
    const wan_status_t st = wan_once_per_device(attr_done, +[]() -> wan_status_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) {
            wan_set_error("wan_gemm_bf16: cannot reserve %d B of LDS: %s", kLdsBytes, hipGetErrorString(e));
            return WAN_ERR_LAUNCH;
        }
        return WAN_OK;
    });
    if (st != WAN_OK) return st;
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)batch), block(kThreads);
    hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, grid, block, kLdsBytes, s, g);
    WAN_CHECK_LAUNCH("wan_gemm_bf16");
    return WAN_OK;
}

// split-K launch of the 128^2 kernel (small shapes that bring a workspace: see wan_gemm_splitk below)
template <int EPI>
wan_status_t launch_splitk(const GemmArgs& g, hipStream_t s, int64_t counter_bytes) {
    static std::atomic<uint64_t> attr_done{0};
    const wan_status_t st = wan_once_per_device(attr_done, +[]() -> wan_status_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<EPI, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) {
            wan_set_error("wan_gemm_bf16_ws: cannot reserve %d B of LDS: %s", kLdsBytes, hipGetErrorString(e));
            return WAN_ERR_LAUNCH;
        }
        return WAN_OK;
    });
    if (st != WAN_OK) return st;
    if (hipMemsetAsync(g.counters, 0, (size_t)counter_bytes, s) != hipSuccess) {
        wan_set_error("wan_gemm_bf16_ws: cannot clear the arrival counters: %s", hipGetErrorString(hipGetLastError()));
        return WAN_ERR_LAUNCH;
    }
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)g.splitk), block(kThreads);
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, true>), grid, block, kLdsBytes, s, g);
    WAN_CHECK_LAUNCH("wan_gemm_bf16_ws (split-K)");
    return WAN_OK;
}

}  // namespace

bool wan_gemm256_uses_w4(int K);        // gemm_bf16_256.hip

// Small shapes on the 128^2 kernel (configs[0]: M = 2 304 tokens): when its output tiles do not fill the chip's 2 x CUs workgroup
// slots and K is deep enough, the K range of every tile is cut into `splits` pieces (2 .. 4) so that the launch is (close to) one
// full round of shorter workgroups -- ffn.2 at M = 2 304 (N = 1 536, K = 8 960) is 216 tiles of 140 serial K steps on 256 CUs, the
// same product as 432 workgroups of 70.  Needs the caller's workspace (wan_gemm_bf16_ws); 1 = no split.
static int wan_gemm_splitk(int M, int N, int K) {
    if (wan_tune(WAN_TUNE_GEMM_SPLITK) == 0) return 1;
    const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int slots = 2 * wan_cu_count();
    const int nk = K / BK;
    if (const int f = wan_tune(WAN_TUNE_GEMM_SPLITK); f > 1) return (f <= 8 && nk >= 2 * f) ? f : 1;      // developer override
    if (tiles * 4 > (int64_t)slots * 3 || nk < 64) return 1;                // >= 3/4 of a round already, or nothing to cut
    // measured at M = 2 304 (profiles/r05/gemm_yardstick_small_splitk.log): K = 8 960 in two pieces 0.105 -> 0.087 ms; K = 1 536 in two pieces
    // 0.027 -> 0.033 ms -- the counter memset, the 64 KB round trip per piece and the second launch wave cost more than 12 K tiles
    int splits = (int)std::min<int64_t>(slots / tiles, 4);
    while (splits > 1 && nk / splits < 32) --splits;                       // at least 32 K tiles per piece
    return splits < 2 ? 1 : splits;
}
static int64_t splitk_counter_bytes(int M, int N) {
    const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    return (tiles * 4 + 4095) / 4096 * 4096;
}
static int64_t splitk_workspace_bytes(int M, int N, int splits) {
    const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    return splitk_counter_bytes(M, N) + tiles * splits * (int64_t)(BM * BN * 4);
}

// Which kernel family wan_gemm_bf16 dispatches a shape to (host arithmetic, no GPU needed).  Large shapes -> the 256^2 tile
// (one workgroup per CU: 8-wave phased kernel, or its 4-wave form for deep K), unless its tiles would leave more than half of
// the CUs idle (M ~ 1e3: the text encoder, the VAE's attention block): four times as many 128^2 tiles at two per CU fill the
// chip better.  gemm_variant = 1|2 is a developer A/B switch (wan_set_tuning), not a product option.
extern "C" int wan_gemm_plan(int M, int N, int K) {
    const int variant = wan_tune(WAN_TUNE_GEMM_VARIANT);
    const int64_t tiles256 = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
    const bool big = M >= 1024 && N >= 256 && 2 * tiles256 > wan_cu_count();
    if (!(variant == 2 || (variant == 0 && big))) return WAN_GEMM_VARIANT_128;
    return wan_gemm256_uses_w4(K) ? WAN_GEMM_VARIANT_256_W4 : WAN_GEMM_VARIANT_256_W8;
}

// The persistent stream-K form (gemm_bf16_pk.hip) takes a product when the caller brought a workspace and a 256^2 kernel would
// have run it (gemm_pk = 1, default): every "big" shape with K % 128 == 0 and K >= 1024.  Round 4 stopped at K >= 4096 (where the
// 4-wave per-tile kernel ran); round 5 measured the K = 1536 Linears of the 1.3B model at M = 67 080 (profiles/r05/
// gemm_yardstick_1p3b_gate.log): persistent 0.567 / 0.335 / 0.386 / 0.282 / 1.551 ms against 0.638 / 0.342 / 0.443 / 0.325 / 1.654 for the
// 8-wave per-tile kernel (q|k, V^T, o + resid, cross q, ffn.0) -- at 24 K tiles per output tile the per-tile pipeline fill is
// >= 8 % of a tile, which a continuous K-tile stream does not pay.  Shallower K (the VAE attention block's K = 384) stays where it
// was: there the epilogue dominates and a second workgroup per CU hides it.
// gemm_pk = 2: whenever its shape rules allow (K % 128 == 0, at least one 256^2 tile each way); 0: never.
extern "C" int wan_gemm_ws_plan(int M, int N, int K) {
    const int pk = wan_tune(WAN_TUNE_GEMM_PK);
    const int base = wan_gemm_plan(M, N, K);
    // (shallow K only with at least four rounds of tiles: at M = 2 304 the K = 1536 ffn.0 of the 1.3B model is 315 tiles on 256 CUs --
    // mostly stream-K pieces, whose fix-up traffic costs more than the per-tile pipeline fill it saves: 0.095 vs 0.086 ms)
    const int64_t tiles256 = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
    if (pk == 1 && base != WAN_GEMM_VARIANT_128 && K % 128 == 0 && (K >= 4096 || (K >= 1024 && tiles256 >= 4 * (int64_t)wan_cu_count())))
        return WAN_GEMM_VARIANT_256_PK;
    if (pk == 2 && K % 128 == 0 && M >= 256 && N >= 256) return WAN_GEMM_VARIANT_256_PK;
    return base;
}

extern "C" int64_t wan_gemm_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int plan = wan_gemm_ws_plan(M, N, K);
    if (plan == WAN_GEMM_VARIANT_256_PK) return wan_gemm_pk_workspace_bytes(M, N);
    if (plan == WAN_GEMM_VARIANT_128 && K % BK == 0) {
        const int splits = wan_gemm_splitk(M, N, K);
        if (splits > 1) return splitk_workspace_bytes(M, N, splits);
    }
    return 0;
}

// how many pieces wan_gemm_bf16_ws cuts the K range of this shape's tiles into (1: no split; host arithmetic)
extern "C" int wan_gemm_ws_splits(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK != 0 || wan_gemm_ws_plan(M, N, K) != WAN_GEMM_VARIANT_128) return 1;
    return wan_gemm_splitk(M, N, K);
}

extern "C" wan_status_t wan_gemm_bf16_ws(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                         void* out, int64_t ldo, int M, int N, int K, int epilogue,
                                         const float* gate, int64_t rows_per_batch, void* workspace, int64_t workspace_bytes,
                                         void* stream) {
    // small shapes: the 128^2 kernel with its K range cut into pieces when that fills the chip (the arguments are validated by
    // wan_gemm_bf16 first: a split launch of an invalid call must not happen)
    if (workspace != nullptr && M > 0 && N > 0 && K > 0 && K % BK == 0 && wan_gemm_ws_plan(M, N, K) == WAN_GEMM_VARIANT_128) {
        const int splits = wan_gemm_splitk(M, N, K);
        if (splits > 1 && workspace_bytes >= splitk_workspace_bytes(M, N, splits) && ((uintptr_t)workspace & 15) == 0 && A && W && out &&
            N % 4 == 0 && lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ldo % 4 == 0 &&
            (epilogue == WAN_EPI_BF16_T ? ldo >= M : ldo >= N) && (gate == nullptr || (epilogue == WAN_EPI_RESID_F32 && rows_per_batch > 0))) {
            GemmArgs g;
            g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.bias = bias;
            g.out = out; g.ldo = ldo; g.gate = gate; g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
            g.M = M; g.N = N; g.K = K;
            g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
            g.sA = g.sW = g.sO = 0;
            g.splitk = splits; g.counters = (int*)workspace; g.slots = (char*)workspace + splitk_counter_bytes(M, N);
            hipStream_t s = (hipStream_t)stream;
            const int64_t cb = splitk_counter_bytes(M, N);
            switch (epilogue) {
                case WAN_EPI_BF16: return launch_splitk<WAN_EPI_BF16>(g, s, cb);
                case WAN_EPI_GELU_BF16: return launch_splitk<WAN_EPI_GELU_BF16>(g, s, cb);
                case WAN_EPI_F32: return launch_splitk<WAN_EPI_F32>(g, s, cb);
                case WAN_EPI_RESID_F32: return launch_splitk<WAN_EPI_RESID_F32>(g, s, cb);
                case WAN_EPI_BF16_T: return launch_splitk<WAN_EPI_BF16_T>(g, s, cb);
                default: break;          // wan_gemm_bf16 reports it
            }
        }
    }
    // (a gate whose samples are shorter than a wave's 128 rows: the persistent kernel's epilogue allows one sample seam per wave)
    if (workspace == nullptr || M <= 0 || N <= 0 || K <= 0 || wan_gemm_ws_plan(M, N, K) != WAN_GEMM_VARIANT_256_PK ||
        (gate != nullptr && rows_per_batch < 128))
        return wan_gemm_bf16(A, lda, W, ldw, bias, out, ldo, M, N, K, epilogue, gate, rows_per_batch, stream);
    WAN_REQUIRE(A && W && out, WAN_ERR_INVALID, "wan_gemm_bf16_ws: null tensor");
    WAN_REQUIRE(N % 4 == 0, WAN_ERR_UNSUPPORTED, "wan_gemm_bf16_ws: N=%d must be a multiple of 4", N);
    WAN_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K, WAN_ERR_INVALID,
                "wan_gemm_bf16_ws: lda=%lld ldw=%lld must be multiples of 8 and >= K", (long long)lda, (long long)ldw);
    if (epilogue == WAN_EPI_BF16_T)
        WAN_REQUIRE(ldo >= M && ldo % 4 == 0, WAN_ERR_INVALID, "wan_gemm_bf16_ws: transposed ldo=%lld < M=%d or not a multiple of 4", (long long)ldo, M);
    else
        WAN_REQUIRE(ldo >= N && ldo % 4 == 0, WAN_ERR_INVALID, "wan_gemm_bf16_ws: ldo=%lld < N=%d or not a multiple of 4", (long long)ldo, N);
    WAN_REQUIRE(gate == nullptr || (epilogue == WAN_EPI_RESID_F32 && rows_per_batch > 0), WAN_ERR_INVALID,
                "wan_gemm_bf16_ws: gate needs WAN_EPI_RESID_F32 and rows_per_batch > 0");
    WAN_REQUIRE(workspace_bytes >= wan_gemm_pk_workspace_bytes(M, N), WAN_ERR_INVALID,
                "wan_gemm_bf16_ws: workspace of %lld bytes, wan_gemm_workspace_bytes(%d, %d, %d) = %lld", (long long)workspace_bytes, M, N, K,
                (long long)wan_gemm_pk_workspace_bytes(M, N));
    WAN_REQUIRE(((uintptr_t)workspace & 15) == 0, WAN_ERR_INVALID, "wan_gemm_bf16_ws: workspace must be 16-byte aligned");
    return wan_gemm_bf16_pk(A, lda, W, ldw, bias, out, ldo, M, N, K, epilogue, gate, rows_per_batch, workspace, (hipStream_t)stream);
}

// The e4m3 Linear with a caller workspace: the persistent stream-K kernel's FP8 instantiation (gemm_bf16_pk.hip, "schedule P") where
// the bf16 product of the same TILE count would run persistent -- a K tile is 128 e4m3 elements, so the plan is asked about K / 2 --
// or where the bf16 product of the same SHAPE would and K >= 4096 (the 8-way Ulysses shard's M = 8 392: 660 tiles of 40 K tiles; measured
// 1.09-1.33x the per-tile kernel there, profiles/r06/gemm_fp8_sp8_shard.log); wan_gemm_fp8 (the 8-wave per-tile kernel) otherwise.
// Same contract as wan_gemm_bf16_ws: the workspace (wan_gemm_fp8_workspace_bytes(M, N, K) bytes) is not shared with another stream.
extern "C" int wan_gemm_fp8_ws_plan(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 256 != 0) return WAN_GEMM_VARIANT_256_W8;
    if (wan_gemm_ws_plan(M, N, K / 2) == WAN_GEMM_VARIANT_256_PK) return WAN_GEMM_VARIANT_256_PK;
    return (K >= 4096 && wan_gemm_ws_plan(M, N, K) == WAN_GEMM_VARIANT_256_PK) ? WAN_GEMM_VARIANT_256_PK : WAN_GEMM_VARIANT_256_W8;
}

extern "C" int64_t wan_gemm_fp8_workspace_bytes(int M, int N, int K) {
    return wan_gemm_fp8_ws_plan(M, N, K) == WAN_GEMM_VARIANT_256_PK ? wan_gemm_pk_workspace_bytes(M, N) : 0;
}

extern "C" wan_status_t wan_gemm_fp8_ws(const void* A_fp8, int64_t lda, const float* a_row_scale, const void* W_fp8, int64_t ldw,
                                        const float* w_row_scale, const float* bias, void* out, int64_t ldo, int M, int N, int K,
                                        int epilogue, const float* gate, int64_t rows_per_batch, void* workspace, int64_t workspace_bytes,
                                        void* stream) {
    if (workspace == nullptr || wan_gemm_fp8_ws_plan(M, N, K) != WAN_GEMM_VARIANT_256_PK || (gate != nullptr && rows_per_batch < 128))
        return wan_gemm_fp8(A_fp8, lda, a_row_scale, W_fp8, ldw, w_row_scale, bias, out, ldo, M, N, K, epilogue, gate, rows_per_batch, stream);
    WAN_REQUIRE(A_fp8 && W_fp8 && out && a_row_scale && w_row_scale, WAN_ERR_INVALID, "wan_gemm_fp8_ws: null tensor");
    WAN_REQUIRE(N % 4 == 0, WAN_ERR_UNSUPPORTED, "wan_gemm_fp8_ws: N=%d must be a multiple of 4", N);
    WAN_REQUIRE(lda % 16 == 0 && ldw % 16 == 0 && lda >= K && ldw >= K, WAN_ERR_INVALID,
                "wan_gemm_fp8_ws: lda=%lld ldw=%lld must be multiples of 16 and >= K", (long long)lda, (long long)ldw);
    WAN_REQUIRE(((uintptr_t)w_row_scale & 15) == 0, WAN_ERR_INVALID, "wan_gemm_fp8_ws: w_row_scale must be 16-byte aligned");
    if (epilogue == WAN_EPI_BF16_T)
        WAN_REQUIRE(ldo >= M && ldo % 4 == 0, WAN_ERR_INVALID, "wan_gemm_fp8_ws: transposed ldo=%lld < M=%d or not a multiple of 4", (long long)ldo, M);
    else
        WAN_REQUIRE(ldo >= N && ldo % 4 == 0, WAN_ERR_INVALID, "wan_gemm_fp8_ws: ldo=%lld < N=%d or not a multiple of 4", (long long)ldo, N);
    WAN_REQUIRE(gate == nullptr || (epilogue == WAN_EPI_RESID_F32 && rows_per_batch > 0), WAN_ERR_INVALID,
                "wan_gemm_fp8_ws: gate needs WAN_EPI_RESID_F32 and rows_per_batch > 0");
    WAN_REQUIRE(workspace_bytes >= wan_gemm_pk_workspace_bytes(M, N), WAN_ERR_INVALID,
                "wan_gemm_fp8_ws: workspace of %lld bytes, wan_gemm_fp8_workspace_bytes(%d, %d, %d) = %lld", (long long)workspace_bytes, M, N, K,
                (long long)wan_gemm_pk_workspace_bytes(M, N));
    WAN_REQUIRE(((uintptr_t)workspace & 15) == 0, WAN_ERR_INVALID, "wan_gemm_fp8_ws: workspace must be 16-byte aligned");
    return wan_gemm_fp8_pk(A_fp8, lda, a_row_scale, W_fp8, ldw, w_row_scale, bias, out, ldo, M, N, K, epilogue, gate, rows_per_batch, workspace,
                           (hipStream_t)stream);
}

extern "C" wan_status_t wan_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                      void* out, int64_t ldo, int M, int N, int K, int epilogue,
                                      const float* gate, int64_t rows_per_batch, void* stream) {
    WAN_REQUIRE(A && W && out, WAN_ERR_INVALID, "wan_gemm_bf16: null tensor");
    WAN_REQUIRE(M >= 0 && N > 0 && K > 0, WAN_ERR_INVALID, "wan_gemm_bf16: M=%d N=%d K=%d", M, N, K);
    WAN_REQUIRE(K % BK == 0, WAN_ERR_UNSUPPORTED, "wan_gemm_bf16: K=%d must be a multiple of %d", K, BK);
    WAN_REQUIRE(N % 4 == 0, WAN_ERR_UNSUPPORTED, "wan_gemm_bf16: N=%d must be a multiple of 4", N);
    WAN_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K, WAN_ERR_INVALID,
                "wan_gemm_bf16: lda=%lld ldw=%lld must be multiples of 8 and >= K", (long long)lda, (long long)ldw);
    if (epilogue == WAN_EPI_BF16_T)
        WAN_REQUIRE(ldo >= M && ldo % 4 == 0, WAN_ERR_INVALID, "wan_gemm_bf16: transposed ldo=%lld < M=%d or not a multiple of 4", (long long)ldo, M);
    else
        WAN_REQUIRE(ldo >= N && ldo % 4 == 0, WAN_ERR_INVALID, "wan_gemm_bf16: ldo=%lld < N=%d or not a multiple of 4", (long long)ldo, N);
    WAN_REQUIRE(gate == nullptr || (epilogue == WAN_EPI_RESID_F32 && rows_per_batch > 0), WAN_ERR_INVALID,
                "wan_gemm_bf16: gate needs WAN_EPI_RESID_F32 and rows_per_batch > 0");
    if (M == 0) return WAN_OK;
    if (wan_gemm_plan(M, N, K) != WAN_GEMM_VARIANT_128)
        return wan_gemm_bf16_256(A, lda, W, ldw, bias, out, ldo, M, N, K, epilogue, gate, rows_per_batch, (hipStream_t)stream);
    GemmArgs g;
    g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.bias = bias;
    g.out = out; g.ldo = ldo; g.gate = gate; g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    g.M = M; g.N = N; g.K = K;
    g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
    g.sA = g.sW = g.sO = 0;
    g.splitk = 1; g.counters = nullptr; g.slots = nullptr;
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case WAN_EPI_BF16: return launch<WAN_EPI_BF16>(g, s);
        case WAN_EPI_GELU_BF16: return launch<WAN_EPI_GELU_BF16>(g, s);
        case WAN_EPI_F32: return launch<WAN_EPI_F32>(g, s);
        case WAN_EPI_RESID_F32: return launch<WAN_EPI_RESID_F32>(g, s);
        case WAN_EPI_BF16_T: return launch<WAN_EPI_BF16_T>(g, s);
        default: wan_set_error("wan_gemm_bf16: unknown epilogue %d", epilogue); return WAN_ERR_INVALID;
    }
}

extern "C" wan_status_t wan_gemm_bf16_batched(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw,
                                              int64_t strideW, void* out, int64_t ldo, int64_t strideO,
                                              int M, int N, int K, int batch, int epilogue, void* stream) {
    WAN_REQUIRE(A && W && out, WAN_ERR_INVALID, "wan_gemm_bf16_batched: null tensor");
    WAN_REQUIRE(M >= 0 && N > 0 && K > 0 && batch >= 0 && batch <= 65535, WAN_ERR_INVALID,
                "wan_gemm_bf16_batched: M=%d N=%d K=%d batch=%d", M, N, K, batch);
    WAN_REQUIRE(K % BK == 0 && N % 4 == 0, WAN_ERR_UNSUPPORTED, "wan_gemm_bf16_batched: K=%d %% 64 and N=%d %% 4 must be 0", K, N);
    WAN_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && strideA % 8 == 0 && strideW % 8 == 0, WAN_ERR_INVALID,
                "wan_gemm_bf16_batched: lda=%lld ldw=%lld strideA=%lld strideW=%lld must be multiples of 8, ld >= K",
                (long long)lda, (long long)ldw, (long long)strideA, (long long)strideW);
    WAN_REQUIRE(ldo >= N && ldo % 4 == 0 && strideO % 4 == 0, WAN_ERR_INVALID,
                "wan_gemm_bf16_batched: ldo=%lld strideO=%lld", (long long)ldo, (long long)strideO);
    WAN_REQUIRE(epilogue == WAN_EPI_BF16 || epilogue == WAN_EPI_F32, WAN_ERR_UNSUPPORTED,
                "wan_gemm_bf16_batched: epilogue %d (only WAN_EPI_BF16 / WAN_EPI_F32)", epilogue);
    if (M == 0 || batch == 0) return WAN_OK;
    GemmArgs g;
    g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.bias = nullptr;
    g.out = out; g.ldo = ldo; g.gate = nullptr; g.rows_per_batch = 1;
    g.M = M; g.N = N; g.K = K;
    g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
    g.sA = strideA; g.sW = strideW; g.sO = strideO;
    g.splitk = 1; g.counters = nullptr; g.slots = nullptr;
    hipStream_t s = (hipStream_t)stream;
    return epilogue == WAN_EPI_BF16 ? launch<WAN_EPI_BF16>(g, s, batch) : launch<WAN_EPI_F32>(g, s, batch);
}
