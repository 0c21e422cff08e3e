// Persistent stream-K form of the 4-wave 256 x 256 x 64 bf16 GEMM (round 4).
//
// Same wave tiles, MFMA order, LDS image, staging and epilogues as gemm_w4_kernel (gemm_bf16_256.hip); what changes is who
// walks the output tiles.  There, one workgroup per tile: every tile pays its own pipeline fill (K tile 0 requested, waited for,
// published) and its own ramp-down, a launch is ceil(tiles / CUs) rounds of identical length whatever the remainder, and the
// workgroups of a round reach their (HBM-bound) epilogues together.  Here ONE workgroup per CU stays resident and consumes a
// sequence of SEGMENTS -- (tile, K range) pairs -- as one continuous stream of K tiles:
//
//   * the DMA pipeline never drains: the last K tiles of a segment request the first K tiles of the next one (other A / W panel,
//     same LDS ring), which land under the epilogue;
//   * work is cut per XCD (worker w = blockIdx: XCD w % 8, lane c = w / 8 of W = grid / 8): each XCD owns a contiguous slab of the
//     XCD-rasterised tile sequence (GM x all-N groups, as before); lane c takes tiles c, c + W, c + 2W, ... of the slab whole
//     ("data-parallel" rounds) and the remainder S = slab mod W tiles are cut STREAM-K fashion: their S * K/128 units (a unit =
//     two K tiles, so that every segment starts on LDS buffer 0) are dealt to the lanes in W_sk equal contiguous ranges.  A range
//     covers the tail of one tile and/or the head of the next; every worker does the same amount of work and there is no tail round
//     (5260 tiles on 256 CUs: 20.55 rounds instead of 21; the 8-way Ulysses shard's 660 tiles: 2.58 instead of 3);
//   * a partial segment stores its fp32 accumulators to the workspace (register-major, 16 B per lane and instruction: fully
//     coalesced), releases them at agent scope and takes a ticket on the tile's arrival counter; the LAST arriver acquires, adds
//     the pieces IN K ORDER (its own from registers) -- so the result does not depend on who arrived last: bitwise reproducible --
//     and runs the normal epilogue.  No workgroup ever waits for another one (no spin loops: nothing to dead-lock, whatever the
//     dispatch order or residency), and a full tile never touches the workspace.
//
// Contract differences to wan_gemm_bf16: the caller passes a workspace (wan_gemm_workspace_bytes) that no other stream uses
// concurrently; its first 4 KiB (arrival counters) are zeroed by a memset node ahead of the launch.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

#ifndef WAN_DEV_EXPERIMENTS
#define WAN_DEV_EXPERIMENTS 0
#endif

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kThreads = 256;
constexpr int kOperandBytes = BM * BK * 2;     // 32 KiB
constexpr int kBufBytes = 2 * kOperandBytes;   // A + W: 64 KiB
constexpr int kLdsBytes = 2 * kBufBytes;       // 128 KiB
constexpr int kLdsFlag = kLdsBytes;            // three ints behind the ring: the arrival ticket and the tile tickets, broadcast to the workgroup
constexpr int kCounterBytes = 4096;            // arrival counters (one int per stream-K tile) at the head of the workspace
constexpr int64_t kSlotBytes = (int64_t)BM * BN * 4;
constexpr int kMaxWorkers = 512;               // arrival counters [0, 512), ticket counters at 512 + 16 x: one 4 KiB header

struct PkArgs {
    const bf16_t* A; int64_t lda;
    const bf16_t* W; int64_t ldw;
    const float* bias;
    void* out; int64_t ldo;
    const float* gate; int64_t rows_per_batch;
    int M, N, K;
    int tiles_m, tiles_n;
    int gm;               // M tiles per rasterisation group
    int nworkers;         // grid size (multiple of 8)
    int min_units;        // smallest stream-K range worth a worker (units of two K tiles)
    int form;             // 1 (product): epilogues on row-permuted operand tiles (see the kernel); 0: the round-4 epilogues (developer A/B)
    int exp;              // `make EXPERIMENTS=1` builds only (gemm_exp), TIMING ONLY: bit 5 = no epilogue at all, bit 6 = s_memtime stamps (where a workgroup's cycles go)
    int dynamic;          // whole tiles by ticket from the per-XCD counters (1) or in lockstep order (0: developer A/B)
    int* counters;        // workspace head: [0, nworkers) arrival counters of the split tiles, [512 + 16 x] the ticket counter of XCD x
    char* slots;          // workspace + kCounterBytes: 2 slots of 256 KiB per worker
    // FP8 instantiation (wan_gemm_fp8_ws): A / W point at e4m3 bytes, lda / ldw count bytes = elements; the product of the quantised
    // operands is scaled by sa[m] * sw[n] (per token row, per output channel) on its way into the epilogue
    const float* sa; const float* sw;
    int ktile;            // host side only (the kernels know it at compile time): elements per K tile, 64 (bf16) or 128 (e4m3)
};

// One worker's view of its XCD slab: pure integer functions of (problem, grid, blockIdx) -- the host computes nothing per tile.
struct Slab {
    int start, n;         // first tile (in the XCD-rasterised sequence) and tile count of this XCD's slab
    int W, c, x;          // lanes per XCD, my lane, my XCD
    int rounds;           // whole-tile rounds
    int S, n2;            // stream-K tiles of the slab, units (two K tiles) per tile
    int U, Wsk;           // stream-K units, lanes that take a range
    int nk;
};
// (j * U and u * Wsk stay far below 2^31: U <= lanes * K/128, lanes <= 128)
__host__ __device__ __forceinline__ int range_begin(const Slab& s, int j) { return (int)((unsigned)(j * s.U) / (unsigned)s.Wsk); }
__host__ __device__ __forceinline__ int find_range(const Slab& s, int u) {       // the range that contains unit u
    int p = (int)((unsigned)(u * s.Wsk) / (unsigned)s.U);
    while (p + 1 < s.Wsk && range_begin(s, p + 1) <= u) ++p;
    while (p > 0 && range_begin(s, p) > u) --p;
    return p;
}

struct Seg {
    int valid;
    int tm, tn;
    int kb, ke;           // K tiles [kb, ke), both even
    int partial;          // 0: the whole K range of the tile (direct epilogue); 1: a piece
    int slot, cnt;        // partial: my workspace slot, the tile's arrival counter
    int j_lo, j_hi, me;   // partial: lanes holding the tile's pieces in K order, and mine
    int tau;              // partial: stream-K tile index inside the slab
};

__host__ __device__ __forceinline__ void tile_of(const PkArgs& g, int t, int& tm, int& tn) {
    const int per_group = g.gm * g.tiles_n;
    const int grp = t / per_group;
    const int first_m = grp * g.gm;
    const int gm = min(g.gm, g.tiles_m - first_m);
    const int in = t - grp * per_group;
    tm = first_m + in % gm;
    tn = in / gm;
}

__host__ __device__ __forceinline__ Slab make_slab(const PkArgs& g, int worker, int ktile) {
    Slab s;
    const int nt = g.tiles_m * g.tiles_n;
    s.x = worker & 7; s.c = worker >> 3; s.W = g.nworkers >> 3;
    const int q = nt >> 3, r = nt & 7;
    s.start = s.x < r ? s.x * (q + 1) : r * (q + 1) + (s.x - r) * q;
    s.n = q + (s.x < r ? 1 : 0);
    s.nk = g.K / ktile;
    s.n2 = s.nk >> 1;
    s.rounds = s.n / s.W;
    s.S = s.n - s.rounds * s.W;
    s.U = s.S * s.n2;
    s.Wsk = s.U > 0 ? max(1, min(s.W, s.U / max(1, g.min_units))) : 0;
    return s;
}

__host__ __device__ __forceinline__ Seg no_seg(const Slab& s) {
    Seg e;
    e.valid = 0; e.tm = e.tn = 0; e.kb = 0; e.ke = s.nk; e.partial = 0; e.slot = e.cnt = 0; e.j_lo = e.j_hi = e.me = 0; e.tau = 0;
    return e;
}
// whole tile `ticket` (0 .. rounds * W - 1) of the slab's data-parallel part
__host__ __device__ __forceinline__ Seg dp_seg(const PkArgs& g, const Slab& s, int ticket) {
    Seg e = no_seg(s);
    if (ticket < 0 || ticket >= s.rounds * s.W) return e;
    e.valid = 1;
    tile_of(g, s.start + ticket, e.tm, e.tn);
    return e;
}
// segment k (0, 1, ...) of my stream-K range; valid = 0 past its end (or when my lane has no range)
__host__ __device__ __forceinline__ Seg sk_seg(const PkArgs& g, const Slab& s, int k) {
    Seg e = no_seg(s);
    if (s.c >= s.Wsk) return e;
    const int u0 = range_begin(s, s.c), u1 = range_begin(s, s.c + 1);
    if (u1 <= u0) return e;
    const int tau0 = u0 / s.n2, nseg = (u1 - 1) / s.n2 - tau0 + 1;
    if (k >= nseg) return e;
    const int tau = tau0 + k;
    const int b = max(u0, tau * s.n2), en = min(u1, (tau + 1) * s.n2);
    e.valid = 1;
    tile_of(g, s.start + s.rounds * s.W + tau, e.tm, e.tn);
    e.kb = 2 * (b - tau * s.n2); e.ke = 2 * (en - tau * s.n2);
    e.partial = !(e.kb == 0 && e.ke == s.nk);
    e.tau = tau;
    if (e.partial) {
        e.slot = 2 * (s.x * s.W + s.c) + (k == 0 ? 0 : 1);
        e.cnt = s.x * s.W + tau;
        e.j_lo = find_range(s, tau * s.n2);
        e.j_hi = find_range(s, (tau + 1) * s.n2 - 1);
        e.me = s.c;
    }
    return e;
}
// The schedule when every lane draws its tickets in lockstep (lane c gets c, c + W, c + 2W, ...): what wan_gemm_pk_segment reports.
// On the device the whole-tile tickets are drawn from a per-XCD counter (whoever is free takes the next tile of the slab, so the
// tiles in flight on an XCD stay neighbours); the stream-K ranges are fixed per lane.
__host__ __device__ __forceinline__ Seg get_seg(const PkArgs& g, const Slab& s, int i) {
    return i < s.rounds ? dp_seg(g, s, i * s.W + s.c) : sk_seg(g, s, i - s.rounds);
}

__device__ __forceinline__ u32x4 lds16(const char* p) { return *reinterpret_cast<const u32x4*>(p); }

// The matrix instruction is v_mfma_f32_16x16x32_bf16 (a 128 x 128 wave tile = 8 x 8 tiles, 64 accumulators of 4 registers), not the
// 32x32x16 form of gemm_w4_kernel: on random operands, with every CU busy, the chip sustains 1.99-2.05 PFLOP/s of bare 16x16x32 MFMAs
// against 1.59-1.78 of 32x32x16 (2.45 either way on zeros: tools/probe/mfma_power.hip, profiles/r04/mfma_power.log) -- twice the K depth
// per accumulator update is half the accumulator traffic, and the power that saves comes back as clock.
// FORM 1 (round 5, the product form): every epilogue runs from the UNswapped product D = mfma(A fragment, W fragment) -- lane
// (l15, kg) holds the output column of W-fragment row l15 and the output rows of A-fragment rows 4 kg .. 4 kg + 3 of every 16 x 16
// tile -- and WHICH row of the operand panel sits in which LDS row is chosen so that a lane's accumulators are CONTIGUOUS in the
// output: the LDS-DMA lands a panel row wherever its per-lane source offset says, so the permutation costs nothing (the fragment
// reads, the swizzle and the MFMA sequence are untouched; only the eight scalar piece offsets of an operand change).
//   * bf16 / GELU outputs (W rows permuted, kWPerm = 8): LDS row 16 j + l of a wave's 128-row W block holds panel row 8 l + j, so
//     acc[i][0..7][r] of lane l15 are the 8 ADJACENT columns 8 l15 .. + 7 of output row 16 i + 4 kg + r: one 16-byte store per
//     (i, r), 16 lanes = the wave's whole 256-byte row segment, four rows per instruction (32 stores per lane and tile).
//     Round 4's form (swapped product: a lane = one row, lanes = consecutive ROWS) issued one 16-byte request per LANE -- the
//     16 B per clock and CU that tools/probe/cu_store_rate.hip measures for that pattern against 34 for the row-group forms.
//   * fp32 outputs, plain and read-modify-write (kWPerm = 4): LDS row 16 j + l holds panel row 64 (j >> 2) + 4 l + (j & 3), so
//     acc[i][4 h .. 4 h + 3][r] are the 4 adjacent columns 64 h + 4 l15 .. + 3: two float4 accesses per (i, r), each 16 lanes x 16 B =
//     256 contiguous bytes of a row (round 4: eight dword accesses of 64 contiguous bytes -- a quarter of the instructions, the
//     form the quad-transpose experiment of round 4 paid 3 072 DPP moves for).
//   * transposed (V^T) output (A rows permuted): LDS row 16 i + a of a wave's 128-row A block holds panel row
//     32 (i >> 1) + 8 (a >> 2) + 4 (i & 1) + (a & 3), so (acc[2 t][j][0..3], acc[2 t + 1][j][0..3]) are 8 consecutive tokens of one
//     channel: one 16-byte store, the four kg lanes of a column = 64 contiguous bytes (round 4: 8-byte stores, 32 contiguous bytes).
// Every output element is the same MFMA chain as before (same k order), so results do not change.
// FP8 = true (round 6; wan_gemm_fp8_ws, opt-in lossy mode): the SAME stream, segments, LDS image, stream-K combine and FORM-1 epilogues
// on e4m3 operands.  A K tile is 128 elements -- still 128 bytes per LDS row -- and the two bf16 MFMAs per (m, n) tile and K tile
// become ONE v_mfma_scale_f32_16x16x128_f8f6f4 (every block scale 2^0): 32 matrix-pipe cycles for 4 x the K of a 16-cycle bf16 MFMA,
// i.e. the same cycles, LDS bytes and requests per K tile for twice the FLOPs.  A lane's 32-byte operand = the two adjacent 16-byte
// chunks 2 kg, 2 kg + 1 of its row (A and W use the same lane -> k assignment, which is all a dot product needs).
// That MFMA consumes BOTH chunks of a fragment, so a K tile cannot be consumed half by half as in schedule D.  What replaces it
// ("schedule P"): the 64 MFMAs of a K tile run in two PHASES over the W fragments -- phase 0 = (i, j < 4), phase 1 = (i, j >= 4),
// i-major inside a phase -- so that registers die in an order the NEXT tile can follow with the same 128 fragment registers:
//   * W 0..3 are dead after slot 31 and are refilled (next tile) during phase 1; W 4..7 are dead after slot 63 and are refilled
//     (this tile!) during phase 0 -- they are first needed at slot 32;
//   * A i is last read at slot 35 + 4 i and first needed at slot 4 i of the next tile: every refill has >= 28 slots (~900 cycles);
//   * all reads of a tile's buffer are issued by slot 8 of its own tile (A 7 and W 4..7, whose registers the previous tile held to
//     its end): barrier B1 at slot 12 (behind `lgkmcnt(0)`: those reads have returned) frees the buffer, the 16 requests of K tile t + 2
//     go out at slots 13, 16, ... 58, and barrier
//     B2 at slot 32 (`vmcnt(7)`: the seven requests issued since B1 stay in flight) publishes K tile t + 1, whose last request left
//     38 slots (~1 200 cycles) earlier.
// Two barriers, 32 fragment reads and 16 requests per 2 048 matrix-pipe cycles -- what schedule D has -- with half the MFMA issues.
template <int EPI, int FORM, bool FP8 = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_pk_kernel(PkArgs g) {
    static_assert(!FP8 || FORM == 1, "the e4m3 instantiation runs the row-permuted (FORM 1) epilogues only");
    constexpr int kEl = FP8 ? 1 : 2;                 // bytes per operand element
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // D = mfma(A fragment, W fragment): lane (l15, kg) holds output COLUMN .. + l15 and the four consecutive ROWS .. + 4 kg .. + 3 of a
    // tile -- the form of the transposed store, and (round 4) of the fp32 epilogues: there a register of the tile is 4 rows x 16
    // CONSECUTIVE columns across the 16 lanes of a row group, i.e. 64 contiguous bytes per row group and dword access, where the
    // swapped product puts consecutive ROWS on consecutive lanes and every lane's 16 bytes become a request of their own.
    constexpr bool kOutF32 = (EPI == WAN_EPI_F32 || EPI == WAN_EPI_RESID_F32);
    constexpr bool kTransposed = FORM == 1 || EPI == WAN_EPI_BF16_T || kOutF32;
    constexpr bool kRowMajorOut = (EPI != WAN_EPI_BF16_T);
    constexpr int kWPerm = FORM != 1 ? 0 : (EPI == WAN_EPI_BF16_T ? 0 : (kOutF32 ? 4 : 8));
    constexpr bool kAPerm = FORM == 1 && EPI == WAN_EPI_BF16_T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int l15 = lane & 15, kg = lane >> 4;
    const Slab slab = make_slab(g, (int)blockIdx.x, FP8 ? 2 * BK : BK);

    // ---- LDS-DMA lane offsets (see gemm_w4_kernel): piece j of wave w covers rows 64 w + 8 j + lane / 8 of an operand tile
    // (LDS row of a piece = wid * 64 + 8 j + lane / 8; its swizzle depends on j & 1 only.  Unpermuted, the panel row is the LDS row:
    // the lane part carries 8 (j & 1), the scalar part 16 (j >> 1).  Permuted, the lane part is the same for both parities and the
    // scalar part carries everything that depends on j: a_srow / w_srow below.)
    int a_voff[2], w_voff[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = wid * 64 + p * 8 + (lane >> 3);           // LDS row
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int arow = kAPerm ? 128 * (wid >> 1) + 64 * (wid & 1) + 8 * (lane >> 5) + ((lane >> 3) & 3) : row;
        const int wrow = kWPerm == 8 ? 128 * (wid >> 1) + 4 * (wid & 1) + 8 * (lane >> 3)
                       : kWPerm == 4 ? 128 * (wid >> 1) + 64 * (wid & 1) + 4 * (lane >> 3) : row;
        a_voff[p] = (int)((int64_t)arow * g.lda * kEl + c * 16);
        w_voff[p] = (int)((int64_t)wrow * g.ldw * kEl + c * 16);
    }
    // panel rows piece j adds to the lane part (compile-time per piece)
    auto a_srow = [](int j) { return kAPerm ? 32 * (j >> 2) + 16 * (j & 1) + 4 * ((j >> 1) & 1) : 16 * (j >> 1); };
    auto w_srow = [](int j) { return kWPerm == 8 ? 64 * (j & 1) + (j >> 1) : kWPerm == 4 ? 32 * (j & 1) + (j >> 1) : 16 * (j >> 1); };
    // ---- fragment reads: 16 rows x 32 k per fragment; lane = (row l15, k group kg); logical chunk 4 h + kg of a 128-B row (h = k half)
    const int swz = (lane >> 1) & 7;
    int koff[2][2];                          // [LDS buffer][k half]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h) koff[b][h] = b * kBufBytes + l15 * 128 + (((FP8 ? 2 * kg + h : 4 * h + kg) ^ swz) << 4);
    const int a_base = wr * 128 * 128, w_base = kOperandBytes + wc * 128 * 128;

    // ---- the stream: the current segment and the one behind it.  Whole tiles come by ticket from the XCD's counter (one returning
    // atomic per tile, drawn by thread 0 a whole segment before the ticket is needed and handed to the workgroup through LDS at the
    // segment switch); once the tickets run out a lane walks the segments of its stream-K range.
    int* const ticket_counter = g.counters + 512 + 16 * slab.x;
    volatile int* const lds_ticket = reinterpret_cast<volatile int*>(smem + kLdsFlag);
    bool dp_done = false;
    int sk_k = 0, lock_i = 0;
    auto draw = [&]() -> int {                 // thread 0 only
        return g.dynamic ? __hip_atomic_fetch_add(ticket_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    };
    auto produce = [&](int ticket) {           // the next segment of my sequence, given the ticket drawn for it (uniform)
        if (!dp_done) {
            const Seg e = g.dynamic ? dp_seg(g, slab, ticket) : (lock_i < slab.rounds ? dp_seg(g, slab, lock_i * slab.W + slab.c) : no_seg(slab));
            ++lock_i;
            if (e.valid) return e;
            dp_done = true;
        }
        return sk_seg(g, slab, sk_k++);
    };
    // every field of a segment is wave-uniform by construction; say so (the tickets travel through LDS, which the compiler's
    // uniformity analysis cannot see through -- without this the descriptors end up in VGPRs and every DMA in a waterfall loop)
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto uniform = [&](Seg e) {
        e.valid = uni(e.valid); e.tm = uni(e.tm); e.tn = uni(e.tn); e.kb = uni(e.kb); e.ke = uni(e.ke); e.partial = uni(e.partial);
        e.slot = uni(e.slot); e.cnt = uni(e.cnt); e.j_lo = uni(e.j_lo); e.j_hi = uni(e.j_hi); e.me = uni(e.me); e.tau = uni(e.tau);
        return e;
    };
    int pending = 0;                           // thread 0: the ticket drawn for the segment after `nxt`
    if (tid == 0) {
        lds_ticket[1] = draw(); lds_ticket[2] = draw();
        pending = draw();
    }
    __syncthreads();
    const int t0 = __builtin_amdgcn_readfirstlane(lds_ticket[1]), t1 = __builtin_amdgcn_readfirstlane(lds_ticket[2]);
    Seg cur = uniform(produce(t0));
    if (!cur.valid) return;
    Seg nxt = uniform(produce(t1));

    // A panel = the rows of one output tile in A and in W: origin pointers and the byte distance from the origin (row m0 / n0,
    // column 0) to the end of the last valid row (the descriptors' range check zero-fills rows past M / N)
    struct Panel { const char* a; const char* w; int ab, wb; };
    auto panel = [&](const Seg& e) {
        Panel p;
        const int m0 = e.tm * BM, n0 = e.tn * BN;
        p.a = (const char*)g.A + (int64_t)m0 * g.lda * kEl;
        p.w = (const char*)g.W + (int64_t)n0 * g.ldw * kEl;
        p.ab = (int)min(((int64_t)(min(g.M - m0, BM) - 1) * g.lda + g.K) * kEl, (int64_t)0x7fffffff);
        p.wb = (int)min(((int64_t)(min(g.N - n0, BN) - 1) * g.ldw + g.K) * kEl, (int64_t)0x7fffffff);
        return p;
    };
    Panel pc = panel(cur), pn = panel(nxt);
    // The two request streams.  A descriptor = (address of the K tile inside the panel, bytes left from there); it advances by one K
    // tile (128 B) per request and jumps to the next segment's panel when the position passes the end of the current segment --
    // a handful of SALU instructions, issued in MFMA shadows.
    struct Stream { const char* base; int left; };
    auto stream_at = [&](int op, int pos) {                  // pos: a K tile index of the current segment
        Stream st;
        st.base = (op ? pc.w : pc.a) + (int64_t)pos * (BK * 2);
        st.left = (op ? pc.wb : pc.ab) - pos * (BK * 2);
        return st;
    };
    // the request after this one: the first K tile of the next segment (`jump`), or 128 bytes further in the same panel
    auto advance = [&](Stream& st, int op, bool jump) {
        const char* nb = (op ? pn.w : pn.a) + (int64_t)nxt.kb * (BK * 2);
        const int nl = nxt.valid ? (op ? pn.wb : pn.ab) - nxt.kb * (BK * 2) : 0;
        st.base = jump ? nb : st.base + BK * 2;
        st.left = jump ? nl : max(st.left - BK * 2, 0);       // past the end of the last segment: really zero-length requests
    };
    auto rsrc_of = [&](const Stream& st) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)st.base, 0, st.left, 0x00020000);
    };
    auto stage_piece = [&](__amdgpu_buffer_rsrc_t r, int buf, int operand, int j, int64_t ld) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            r, (__attribute__((address_space(3))) void*)(smem + buf * kBufBytes + operand * kOperandBytes + (wid * 8 + j) * 1024), 16,
            operand ? w_voff[j & 1] : a_voff[j & 1], (int)((operand ? w_srow(j) : a_srow(j)) * ld * kEl), 0, 0);
    };

    f32x4 acc[8][8];                         // [m tile][n tile]
    u32x4 af[2][8], wf[2][8];                // [k half][tile]: the fragments of a whole K tile live in registers

#define GP_SB() __builtin_amdgcn_sched_barrier(0)
// m tiles 0..5 (48 accumulators, 192 registers) are pinned to AGPRs, m tiles 6, 7 to VGPRs: with all 256 AGPRs claimed by "+a"
// operands hipcc's allocator has no slack left and shuffles tiles through scratch (gemm_w4_kernel)
#define GP_MFMA_A(ACC, X, Y) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(X), "v"(Y))
#define GP_MFMA_V(ACC, X, Y) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(X), "v"(Y))
#define GP_MFMA(ACC, X, Y) do { if (i < 6) GP_MFMA_A(ACC, X, Y); else GP_MFMA_V(ACC, X, Y); } while (0)
// FP8: one MX-scaled MFMA over both 16-byte chunks of a fragment pair; formats 0, 0 = e4m3 x e4m3; every block scale = E8M0 127 = 2^0
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    unsigned sc1 = 0;
    if constexpr (FP8) asm volatile("v_mov_b32 %0, 0x7f7f7f7f" : "=v"(sc1));
#define GP_MFMA8_A(ACC, X, Y) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(ACC) : "v"(X), "v"(Y), "v"(sc1))
#define GP_MFMA8_V(ACC, X, Y) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(ACC) : "v"(X), "v"(Y), "v"(sc1))
#define GP_MFMA8(ACC, X, Y) do { if (i < 6) GP_MFMA8_A(ACC, X, Y); else GP_MFMA8_V(ACC, X, Y); } while (0)
    Stream sa, sq;
    // fragment f of a k half in fetch order = first-use order of the MFMA sequence below: A 0, W 0..7, A 1..7
    auto fetch = [&](int h, int f, int buf) __attribute__((always_inline)) {
        if (f == 0) af[h][0] = lds16(smem + a_base + koff[buf][h]);
        else if (f <= 8) wf[h][f - 1] = lds16(smem + w_base + (f - 1) * 16 * 128 + koff[buf][h]);
        else af[h][f - 8] = lds16(smem + a_base + (f - 8) * 16 * 128 + koff[buf][h]);
    };
    // The main-loop schedule ("schedule D" of round 4; the one-barrier-per-K-tile schedule it replaced is gone): shrink the LDS residency of a K tile so that its buffer can take requests for most of the time.
    //   slots   0..31   the 16 fragments of k half 1 (one per 2 MFMAs): by slot 32 every fragment of position kt is in registers
    //   slot   32       barrier B1: buffer b is free
    //   slots  33..123  the 16 pieces (A 0..7, W 0..7) of position kt + 2 -> buffer b, one per 6 MFMAs (96 cycles: 43 B/clk per CU,
    //                   two thirds of the vector-memory path's rate) -- every piece has >= 97 MFMA slots (~1550 cycles) to land
    //   slot   96       barrier B2 behind vmcnt(11): position kt + 1 (requested during the previous K tile) has landed; the 11 pieces
    //                   of position kt + 2 issued so far stay in flight
    //   slots  96..126  the 16 fragments of k half 0 of position kt + 1 (buffer 1 - b)
    auto ktile_d = [&](int kt, int b) __attribute__((always_inline)) {
        const bool jump = kt + 2 == cur.ke;
        __amdgpu_buffer_rsrc_t ra, rw;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s_ = h * 64 + i * 8 + j;
            if (s_ == 32) { __builtin_amdgcn_s_barrier(); GP_SB(); ra = rsrc_of(sa); rw = rsrc_of(sq); }
            if (s_ == 96) { asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); __builtin_amdgcn_s_barrier(); GP_SB(); }
            if constexpr (kTransposed) GP_MFMA(acc[i][j], af[h][i], wf[h][j]);
            else GP_MFMA(acc[i][j], wf[h][j], af[h][i]);
            GP_SB();
            if (s_ < 32 && s_ % 2 == 0) fetch(1, s_ / 2, b);
            else if (s_ == 1) advance(sa, 0, jump);
            else if (s_ == 3) advance(sq, 1, jump);
            else if (s_ >= 33 && (s_ - 33) % 6 == 0) {
                const int k = (s_ - 33) / 6;
                if (k < 8) stage_piece(ra, b, 0, k, g.lda);
                else stage_piece(rw, b, 1, k - 8, g.ldw);
            } else if (s_ >= 96 && s_ % 2 == 0) fetch(0, (s_ - 96) / 2, 1 - b);
            GP_SB();
        }
    };

    // ---- schedule P (FP8; see the kernel header).  fetch8(f, buf): BOTH chunks of fragment f (0..7: A i; 8..15: W j)
    auto fetch8 = [&](int f, int buf) __attribute__((always_inline)) {
        if (f < 8) { af[0][f] = lds16(smem + a_base + f * 16 * 128 + koff[buf][0]); af[1][f] = lds16(smem + a_base + f * 16 * 128 + koff[buf][1]); }
        else { wf[0][f - 8] = lds16(smem + w_base + (f - 8) * 16 * 128 + koff[buf][0]); wf[1][f - 8] = lds16(smem + w_base + (f - 8) * 16 * 128 + koff[buf][1]); }
    };
    auto pair8 = [](const u32x4& lo, const u32x4& hi) __attribute__((always_inline)) {
        return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    };
// (developer A/B builds only, never set by the Makefile: where barrier B1 sits and how densely the requests follow it --
// profiles/r06/gemm_fp8_schedule_p_variants.log: B1 at slot 9 / 12 / 16 within 0.5 %, one request per 2 MFMAs instead of 3 loses 2-5 %)
#ifndef WAN_PKP_B1
#define WAN_PKP_B1 12
#endif
#ifndef WAN_PKP_STRIDE
#define WAN_PKP_STRIDE 3
#endif
    constexpr int kB1 = WAN_PKP_B1, kReq0 = kB1 + 1, kReqStride = WAN_PKP_STRIDE;
    static_assert(kB1 >= 9 && kReq0 + 15 * kReqStride < 64, "all reads of the buffer are issued by slot 8; the 16 requests fit the tile");
    constexpr int kInFlightAtB2 = kReq0 >= 32 ? 0 : (31 - kReq0) / kReqStride + 1 > 16 ? 16 : (31 - kReq0) / kReqStride + 1;     // requests issued before slot 32
    auto ktile_p = [&](int kt, int b) __attribute__((always_inline)) {
        const bool jump = kt + 2 == cur.ke;
        __amdgpu_buffer_rsrc_t ra, rw;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int s_ = ph * 32 + i * 4 + jj, j = ph * 4 + jj;
            // B1: my reads of buffer b have RETURNED (not merely been issued) before I say so -- measured free (the last one is 4 slots old:
            // profiles/r06/gemm_fp8_schedule_p_b1_wait.log), so the hand-over of the buffer does not lean on DMA latency
            if (s_ == kB1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            if (s_ == kB1) { __builtin_amdgcn_s_barrier(); GP_SB(); ra = rsrc_of(sa); rw = rsrc_of(sq); }
            if (s_ == 32) { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(kInFlightAtB2) : "memory"); __builtin_amdgcn_s_barrier(); GP_SB(); }
            {
                const i32x8 a8 = pair8(af[0][i], af[1][i]), w8 = pair8(wf[0][j], wf[1][j]);
                GP_MFMA8(acc[i][j], a8, w8);
            }
            GP_SB();
            // this tile's late fragments (their registers were the previous tile's to its last slot)
            if (s_ == 0) fetch8(7, b);
            else if (s_ == 1) advance(sa, 0, jump);
            else if (s_ == 3) advance(sq, 1, jump);
            else if (s_ == 2 || s_ == 4 || s_ == 6 || s_ == 8) fetch8(8 + 4 + (s_ - 2) / 2, b);
            // K tile kt + 1 (buffer 1 - b, published at slot 32): W 0..3 and A 0..6 into registers that have just died
            else if (s_ == 33) fetch8(8 + 0, 1 - b);
            else if (s_ == 35) fetch8(8 + 1, 1 - b);
            else if (s_ == 36) fetch8(0, 1 - b);
            else if (s_ == 38) fetch8(8 + 2, 1 - b);
            else if (s_ == 41) fetch8(1, 1 - b);
            else if (s_ == 42) fetch8(8 + 3, 1 - b);
            else if (s_ == 44) fetch8(2, 1 - b);
            else if (s_ == 48) fetch8(3, 1 - b);
            else if (s_ == 53) fetch8(4, 1 - b);
            else if (s_ == 56) fetch8(5, 1 - b);
            else if (s_ == 60) fetch8(6, 1 - b);
            // K tile kt + 2 -> buffer b: 16 requests, one per kReqStride MFMAs (3: 96 cycles, as in schedule D)
            if (s_ >= kReq0 && s_ < kReq0 + 16 * kReqStride && (s_ - kReq0) % kReqStride == 0) {
                const int k = (s_ - kReq0) / kReqStride;
                if (k < 8) stage_piece(ra, b, 0, k, g.lda);
                else stage_piece(rw, b, 1, k - 8, g.ldw);
            }
            GP_SB();
        }
    };

    // ---- stream prologue: position cur.kb -> buffer 0 (waited for), the A half of position cur.kb + 1 -> buffer 1 (in flight)
    {
        sa = stream_at(0, cur.kb); sq = stream_at(1, cur.kb);
        const __amdgpu_buffer_rsrc_t ra = rsrc_of(sa), rw = rsrc_of(sq);
#pragma unroll
        for (int j = 0; j < 8; ++j) { stage_piece(ra, 0, 0, j, g.lda); stage_piece(rw, 0, 1, j, g.ldw); }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        sa = stream_at(0, cur.kb + 1); sq = stream_at(1, cur.kb + 1);       // a segment has at least two K tiles
        const __amdgpu_buffer_rsrc_t ra1 = rsrc_of(sa), rw1 = rsrc_of(sq);
#pragma unroll
        for (int j = 0; j < 8; ++j) stage_piece(ra1, 1, 0, j, g.lda);
#pragma unroll
        for (int j = 0; j < 8; ++j) stage_piece(rw1, 1, 1, j, g.ldw);            // the schedule keeps a whole K tile in flight
    }

#if WAN_DEV_EXPERIMENTS
    long long t_start = 0, t_loop = 0, t_epi = 0, t_wait = 0, t_drain = 0, t_mark = __builtin_readcyclecounter();
#define GP_STAMP(ACC) do { if (g.exp & 64) { const long long n_ = __builtin_readcyclecounter(); ACC += n_ - t_mark; t_mark = n_; } } while (0)
#else
#define GP_STAMP(ACC) do { } while (0)
#endif
    for (;;) {
        // ---- segment start: K tile cur.kb sits in buffer 0 (published), the A half of cur.kb + 1 is in flight
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (FP8) {
#pragma unroll
            for (int f = 0; f < 7; ++f) fetch8(f, 0);              // A 0..6
#pragma unroll
            for (int f = 8; f < 12; ++f) fetch8(f, 0);             // W 0..3 (A 7 and W 4..7: slots 0..8 of the first K tile)
            GP_STAMP(t_start);
            for (int kt = cur.kb; kt < cur.ke; kt += 2) {
                ktile_p(kt, 0);
                ktile_p(kt + 1, 1);
            }
        } else {
#pragma unroll
        for (int f = 0; f < 16; ++f) fetch(0, f, 0);
        GP_STAMP(t_start);
        for (int kt = cur.kb; kt < cur.ke; kt += 2) {
            ktile_d(kt, 0);
            ktile_d(kt + 1, 1);
        }
        }
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // MFMA D -> the accumulator reads below
        GP_STAMP(t_loop);

        // ---- partial segment: publish my piece, take a ticket; only the last arriver goes on to the epilogue
        bool reduce = false;
        if (cur.partial) {
            // write-through (sc1) stores: the piece goes to memory past the XCD's L2, so publishing it needs no L2 write-back (a release
            // fence would flush everybody's dirty lines of this XCD: CDNA4 guide, "publish-large") -- drain my stores, then the ticket
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(g.slots + (int64_t)cur.slot * kSlotBytes), 0, (int)kSlotBytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, ((i * 8 + j) * kThreads + tid) * 16, 0, /*sc1*/ 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const int old = __hip_atomic_fetch_add(g.counters + cur.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lds_ticket[0] = old;
            }
            __syncthreads();
            const int old = lds_ticket[0];
            reduce = __builtin_amdgcn_readfirstlane(old) == cur.j_hi - cur.j_lo;
            if (reduce) {
                if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __syncthreads();
            }
        }

#if WAN_DEV_EXPERIMENTS
        if (g.exp & 512) {      // TIMING ONLY: drain everything in flight (the next segment's first K tiles) before the epilogue starts -> t_wait
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GP_STAMP(t_wait);
        }
        if ((g.exp & 32) == 0)
#endif
        if (!cur.partial || reduce) {
            const int m0 = cur.tm * BM, n0 = cur.tn * BN;
            // last arriver of a split tile: the pieces added IN K ORDER (mine from registers), tile by tile, back into my
            // accumulators -- the epilogue below is then the same code for whole and for split tiles
            if (reduce) {
                // 16 tiles (64 registers) at a time: the 16 loads of a piece's chunk are in flight together -- one memory round trip per
                // (chunk, piece) instead of one per tile (64 x pieces dependent round trips cost more than the K loop of the piece itself)
                auto reduce_chunk = [&](auto CH_) {
                    constexpr int CH = decltype(CH_)::value;
                    f32x4 v[16];
                    for (int p = cur.j_lo; p <= cur.j_hi; ++p) {
                        f32x4 t[16];
                        if (p == cur.me) {
#pragma unroll
                            for (int k = 0; k < 16; ++k) t[k] = acc[(CH * 16 + k) >> 3][(CH * 16 + k) & 7];
                        } else {
                            // lane p's segment in this tile sits in its slot 0 when the tile is the first one of its range, else slot 1
                            const int first_tau = range_begin(slab, p) / slab.n2;
                            const f32x4* src = reinterpret_cast<const f32x4*>(
                                g.slots + (int64_t)(2 * (slab.x * slab.W + p) + (first_tau == cur.tau ? 0 : 1)) * kSlotBytes);
#pragma unroll
                            for (int k = 0; k < 16; ++k) t[k] = src[(CH * 16 + k) * kThreads + tid];
                        }
                        if (p == cur.j_lo) {
#pragma unroll
                            for (int k = 0; k < 16; ++k) v[k] = t[k];
                        } else {
#pragma unroll
                            for (int k = 0; k < 16; ++k) v[k] += t[k];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[(CH * 16 + k) >> 3][(CH * 16 + k) & 7] = v[k];
                };
                reduce_chunk(std::integral_constant<int, 0>{}); reduce_chunk(std::integral_constant<int, 1>{});
                reduce_chunk(std::integral_constant<int, 2>{}); reduce_chunk(std::integral_constant<int, 3>{});
            }
            const int l4 = kg * 4;
            int seam_b_lo = 0, seam_next = 0x7fffffff;
            if constexpr (FORM == 1 && (EPI == WAN_EPI_BF16 || EPI == WAN_EPI_GELU_BF16)) {
                // bf16 outputs from the W-row-permuted tile (kWPerm = 8): acc[i][j][r] of lane (l15, kg) is output row mw + 16 i + r,
                // column nb + j -- 8 adjacent columns per (i, r): one 16-byte store, a wave instruction = 4 rows x 256 contiguous bytes
                const int mw = m0 + wr * 128 + l4, nb = n0 + wc * 128 + 8 * l15;
                const bool c0 = nb < g.N, c1 = nb + 4 < g.N;             // N % 4 == 0: a 4-column group is wholly in or out
                const bool aligned = (g.ldo & 7) == 0 && (((uintptr_t)g.out) & 15) == 0;       // wave-uniform
                float bq[8], sq8[8];
                {
                    const float4 b0 = (g.bias && c0) ? *reinterpret_cast<const float4*>(g.bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 b1 = (g.bias && c1) ? *reinterpret_cast<const float4*>(g.bias + nb + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    bq[0] = b0.x; bq[1] = b0.y; bq[2] = b0.z; bq[3] = b0.w; bq[4] = b1.x; bq[5] = b1.y; bq[6] = b1.z; bq[7] = b1.w;
                    if constexpr (FP8) {        // the output channels' weight scales
                        const float4 s0 = c0 ? *reinterpret_cast<const float4*>(g.sw + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
                        const float4 s1 = c1 ? *reinterpret_cast<const float4*>(g.sw + nb + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                        sq8[0] = s0.x; sq8[1] = s0.y; sq8[2] = s0.z; sq8[3] = s0.w; sq8[4] = s1.x; sq8[5] = s1.y; sq8[6] = s1.z; sq8[7] = s1.w;
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v[8];
                        float sm = 1.f;
                        if constexpr (FP8) sm = g.sa[min(mw + 16 * i + r, g.M - 1)];       // the token row's activation scale
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if constexpr (FP8) v[j] = __builtin_fmaf(acc[i][j][r], sm * sq8[j], bq[j]);
                            else v[j] = acc[i][j][r] + bq[j];
                            if constexpr (EPI == WAN_EPI_GELU_BF16) v[j] = gelu_tanh_f32(v[j]);
                        }
                        const u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                        const int m = mw + 16 * i + r;
                        if (m >= g.M) continue;
                        bf16_t* op = (bf16_t*)g.out + (int64_t)m * g.ldo + nb;
#if WAN_DEV_EXPERIMENTS
                        if (g.exp & 2048) { asm volatile("" :: "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3])); continue; }      // TIMING ONLY: everything but the stores
#endif
                        if (aligned && c1) {
#if WAN_DEV_EXPERIMENTS
                            if (g.exp & 4096) { __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(op)); continue; }      // A/B: streaming (nt) stores
#endif
                            *reinterpret_cast<u32x4*>(op) = o;
                        } else {
                            if (c0) *reinterpret_cast<u32x2*>(op) = u32x2{o[0], o[1]};
                            if (c1) *reinterpret_cast<u32x2*>(op + 4) = u32x2{o[2], o[3]};
                        }
                    }
            } else if constexpr (FORM == 1 && kOutF32) {
                // fp32 outputs (plain and read-modify-write) from the W-row-permuted tile (kWPerm = 4): acc[i][4 h + q][r] of lane
                // (l15, kg) is output row mw + 16 i + r, column nb + 64 h + q -- two float4 accesses per (i, r), each 256 contiguous
                // bytes of a row across the 16 lanes of a row group.  A batch = one m tile (8 float4 per lane); the residual rows of
                // batch i + 1 are requested before batch i is combined and stored.  bias / gate: two float4 each per lane and tile.
                auto epilogue_f32 = [&](auto has_bias, auto has_gate, auto seam) {
                    constexpr bool HAS_BIAS = decltype(has_bias)::value, HAS_GATE = decltype(has_gate)::value, SEAM = decltype(seam)::value;
                    constexpr bool RESID = EPI == WAN_EPI_RESID_F32;
                    const int mw = m0 + wr * 128 + l4, nb = n0 + wc * 128 + 4 * l15;
                    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
                    bool nok[2]; int nn[2]; float4 bq[2], gq[2], sq4[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        nok[h] = nb + 64 * h < g.N;
                        nn[h] = nok[h] ? nb + 64 * h : 0;                // clamped: a valid address, not stored
                        bq[h] = HAS_BIAS ? *reinterpret_cast<const float4*>(g.bias + nn[h]) : zero;
                        sq4[h] = FP8 ? *reinterpret_cast<const float4*>(g.sw + nn[h]) : one;
                        gq[h] = (HAS_GATE && !SEAM) ? *reinterpret_cast<const float4*>(g.gate + (int64_t)seam_b_lo * g.N + nn[h]) : one;
                    }
                    struct Rows { float4 x[4][2]; };
                    auto row_of = [&](int i, int r) { return min(mw + i * 16 + r, g.M - 1); };
                    auto load_rows = [&](int i, Rows& R) {
                        if constexpr (RESID) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float* xp = (const float*)g.out + (int64_t)row_of(i, r) * g.ldo;
#pragma unroll
                                for (int h = 0; h < 2; ++h) R.x[r][h] = *reinterpret_cast<const float4*>(xp + nn[h]);
                            }
                        }
                    };
                    auto store_rows = [&](int i, const Rows& R) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = mw + i * 16 + r;
                            if (m >= g.M) continue;
                            float* op = (float*)g.out + (int64_t)m * g.ldo;
                            const float* gp = SEAM ? g.gate + (int64_t)(seam_b_lo + (m >= seam_next ? 1 : 0)) * g.N : nullptr;
                            float sm = 1.f;
                            if constexpr (FP8) sm = g.sa[m];
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                if (!nok[h]) continue;
                                float4 v;
                                if constexpr (FP8)
                                    v = make_float4(__builtin_fmaf(acc[i][4 * h][r], sm * sq4[h].x, bq[h].x), __builtin_fmaf(acc[i][4 * h + 1][r], sm * sq4[h].y, bq[h].y),
                                                    __builtin_fmaf(acc[i][4 * h + 2][r], sm * sq4[h].z, bq[h].z), __builtin_fmaf(acc[i][4 * h + 3][r], sm * sq4[h].w, bq[h].w));
                                else
                                    v = make_float4(acc[i][4 * h][r] + bq[h].x, acc[i][4 * h + 1][r] + bq[h].y,
                                                    acc[i][4 * h + 2][r] + bq[h].z, acc[i][4 * h + 3][r] + bq[h].w);
                                if constexpr (RESID) {
                                    const float4 gv = SEAM ? *reinterpret_cast<const float4*>(gp + nn[h]) : gq[h];
                                    const float4 x = R.x[r][h];
                                    v = make_float4(x.x + v.x * gv.x, x.y + v.y * gv.y, x.z + v.z * gv.z, x.w + v.w * gv.w);
                                }
                                *reinterpret_cast<float4*>(op + nn[h]) = v;
                            }
                        }
                    };
                    Rows R[2];
                    load_rows(0, R[0]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (i + 1 < 8) load_rows(i + 1, R[(i + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        store_rows(i, R[i & 1]);
                    }
                };
                using T = std::true_type; using F = std::false_type;
                if constexpr (EPI == WAN_EPI_RESID_F32) {
                    if (g.gate) {
                        // (as in the round-4 form below: one scalar division per wave and tile; at most one sample seam inside a wave's
                        // 128 rows, and the wave that holds it runs the SEAM copy, which fetches the gate per row)
                        seam_b_lo = __builtin_amdgcn_readfirstlane(min(m0 + wr * 128, g.M - 1) / (int)g.rows_per_batch);
                        seam_next = (seam_b_lo + 1) * (int)g.rows_per_batch;
                        const bool one_sample = min(m0 + wr * 128 + 127, g.M - 1) < seam_next;       // wave-uniform
                        if (one_sample) { if (g.bias) epilogue_f32(T{}, T{}, F{}); else epilogue_f32(F{}, T{}, F{}); }
                        else { if (g.bias) epilogue_f32(T{}, T{}, T{}); else epilogue_f32(F{}, T{}, T{}); }
                    } else {
                        if (g.bias) epilogue_f32(T{}, F{}, F{}); else epilogue_f32(F{}, F{}, F{});
                    }
                } else {
                    if (g.bias) epilogue_f32(T{}, F{}, F{}); else epilogue_f32(F{}, F{}, F{});
                }
            } else if constexpr (FORM == 1) {
                // transposed (V^T) store from the A-row-permuted tile: (acc[2 t][j][0..3], acc[2 t + 1][j][0..3]) of lane (l15, kg) are
                // the 8 consecutive tokens mt .. mt + 7 (mt = m0 + wr * 128 + 32 t + 8 kg) of channel n = .. + 16 j + l15: one 16-byte
                // store; the four kg lanes of a channel cover 64 contiguous bytes
                const bool aligned = (g.ldo & 7) == 0 && (((uintptr_t)g.out) & 15) == 0;       // wave-uniform
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = n0 + wc * 128 + j * 16 + l15;
                    const bool nok = n < g.N;
                    const float bv = (g.bias && nok) ? g.bias[n] : 0.f;
                    if (!nok) continue;
                    const float sn = FP8 ? g.sw[n] : 1.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int m = m0 + wr * 128 + 32 * t + 8 * kg;
                        if (m >= g.M) continue;
                        f32x4 a0 = acc[2 * t][j], a1 = acc[2 * t + 1][j];
                        if constexpr (FP8) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                a0[r] *= sn * g.sa[min(m + r, g.M - 1)];
                                a1[r] *= sn * g.sa[min(m + 4 + r, g.M - 1)];
                            }
                        }
                        const float vv[8] = {a0[0] + bv, a0[1] + bv, a0[2] + bv, a0[3] + bv, a1[0] + bv, a1[1] + bv, a1[2] + bv, a1[3] + bv};
                        bf16_t* p = (bf16_t*)g.out + (int64_t)n * g.ldo + m;
                        if (m + 7 < g.M && aligned) {
                            *reinterpret_cast<u32x4*>(p) = u32x4{pack_bf16x2(vv[0], vv[1]), pack_bf16x2(vv[2], vv[3]), pack_bf16x2(vv[4], vv[5]), pack_bf16x2(vv[6], vv[7])};
                        } else {
#pragma unroll
                            for (int hf = 0; hf < 2; ++hf) {
                                if (m + 4 * hf + 3 < g.M) {
                                    *reinterpret_cast<u32x2*>(p + 4 * hf) = u32x2{pack_bf16x2(vv[4 * hf], vv[4 * hf + 1]), pack_bf16x2(vv[4 * hf + 2], vv[4 * hf + 3])};
                                } else {
                                    for (int r = 0; r < 4 && m + 4 * hf + r < g.M; ++r) p[4 * hf + r] = (bf16_t)vv[4 * hf + r];
                                }
                            }
                        }
                    }
                }
            } else if constexpr (!kTransposed) {
                // bf16 outputs.  Swapped product (W fragment as A operand): lane (l15, kg) holds output row m = .. + l15 and, per tile, the 4
                // consecutive columns n = .. + 4 kg .. + 3 (two bf16 pairs).  A batch = one m tile x four n tiles; the bias of a batch is
                // loaded one batch ahead of its use.  Out-of-range rows / columns read a clamped address and are not stored.
                // (wave-uniform: all 128 columns of the wave exist and every 8-column group is 16-byte aligned)
                const bool wide = n0 + wc * 128 + 128 <= g.N && (g.ldo & 7) == 0 && (((uintptr_t)g.out) & 15) == 0;
                auto epilogue_rows = [&](auto has_bias) {
                    constexpr bool HAS_BIAS = decltype(has_bias)::value;
                    struct Batch {
                        float4 bq[4];
                        int nn[4], mm;
                        bool nok[4], mok;
                    };
                    auto load_batch = [&](int bi, Batch& B) {          // batch bi = m tile bi >> 1, n tiles 4 (bi & 1) .. + 3
                        const int m = m0 + wr * 128 + (bi >> 1) * 16 + l15;
                        B.mok = m < g.M;
                        B.mm = B.mok ? m : g.M - 1;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int n = n0 + wc * 128 + ((bi & 1) * 4 + t) * 16 + l4;
                            B.nok[t] = n < g.N;
                            B.nn[t] = B.nok[t] ? n : 0;
                            if constexpr (HAS_BIAS) B.bq[t] = *reinterpret_cast<const float4*>(g.bias + B.nn[t]);
                            else B.bq[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    };
                    auto store_batch = [&](int bi, const Batch& B) {
                        unsigned pk[4][2];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 a = acc[bi >> 1][(bi & 1) * 4 + q];
                            f32x4 v = {a[0] + B.bq[q].x, a[1] + B.bq[q].y, a[2] + B.bq[q].z, a[3] + B.bq[q].w};
                            if constexpr (EPI == WAN_EPI_GELU_BF16) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f32(v[r]);
                            }
                            // 16-byte stores (CDNA4 guide T21): the tiles (q, q + 1) of a pair trade halves between lanes 16 apart --
                            // v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of its second
                            // -- after which lane groups 0 / 2 own columns 0..7 / 8..15 of tile q and groups 1 / 3 those of tile
                            // q + 1: a store instruction then covers 16 rows x 64 contiguous bytes instead of 16 x 32, and there
                            // are half as many of them
                            pk[q][0] = pack_bf16x2(v[0], v[1]); pk[q][1] = pack_bf16x2(v[2], v[3]);
                            if (q & 1) {
                                if (wide) {
                                    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(pk[q - 1][0]), "+v"(pk[q][0]));
                                    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(pk[q - 1][1]), "+v"(pk[q][1]));
                                    // this lane now holds 8 columns of tile q - 1 + (kg & 1): columns 8 (kg >> 1) .. + 7 of it
                                    const int col = n0 + wc * 128 + ((bi & 1) * 4 + q - 1 + (kg & 1)) * 16 + (kg >> 1) * 8;
                                    const u32x4 o = {pk[q - 1][0], pk[q - 1][1], pk[q][0], pk[q][1]};
                                    if (B.mok) *reinterpret_cast<u32x4*>((bf16_t*)g.out + (int64_t)B.mm * g.ldo + col) = o;
                                } else {
#pragma unroll
                                    for (int qq = q - 1; qq <= q; ++qq) {
                                        if (!(B.mok && B.nok[qq])) continue;
                                        const u32x2 o = {pk[qq][0], pk[qq][1]};
                                        *reinterpret_cast<u32x2*>((bf16_t*)g.out + (int64_t)B.mm * g.ldo + B.nn[qq]) = o;
                                    }
                                }
                            }
                        }
                    };
                    // two batches in flight: the loads of batch b + 1 are issued before batch b is converted and stored
                    Batch B[2];
                    load_batch(0, B[0]);
#pragma unroll
                    for (int bi = 0; bi < 16; ++bi) {
                        if (bi + 1 < 16) load_batch(bi + 1, B[(bi + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        store_batch(bi, B[bi & 1]);
                    }
                };
                if (g.bias) epilogue_rows(std::true_type{});
                else epilogue_rows(std::false_type{});
            } else if constexpr (kRowMajorOut) {
                // fp32 epilogues (plain and read-modify-write) from the UNswapped product: lane (l15, kg) holds column n = .. + l15 and
                // rows m = .. + 4 kg + r of every tile, so register r of tile (i, j) is, across the wave, 4 rows x 64 contiguous bytes.
                // Measured on the swapped form (profiles/r04/gemm_pk_resid_epilogue_loads_vs_stores.log): a lone workgroup moved its
                // epilogue's bytes at ~16 B per clock of its CU whatever the prefetch depth -- one 16-byte request per LANE per clock,
                // because consecutive lanes held consecutive rows.  Here 16 lanes share a request.
                // A batch = one m tile (8 n tiles x 4 rows = 32 dwords per lane); the residual rows of batch i + 1 are requested before
                // batch i is combined and stored.  bias / gate are per column = per lane: 8 + 8 registers for the whole tile.
                auto epilogue_f32 = [&](auto has_bias, auto has_gate, auto seam) {
                    constexpr bool HAS_BIAS = decltype(has_bias)::value, HAS_GATE = decltype(has_gate)::value, SEAM = decltype(seam)::value;
                    constexpr bool RESID = EPI == WAN_EPI_RESID_F32;
                    const int mw = m0 + wr * 128 + l4, nb = n0 + wc * 128 + l15;
                    float bq[8], gq[8];
                    int nn[8];
                    bool nok[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int n = nb + j * 16;
                        nok[j] = n < g.N;
                        nn[j] = nok[j] ? n : 0;
                        bq[j] = HAS_BIAS ? g.bias[nn[j]] : 0.f;
                        gq[j] = (HAS_GATE && !SEAM) ? g.gate[(int64_t)seam_b_lo * g.N + nn[j]] : 1.f;
                    }
                    struct Rows { float x[8][4]; };
                    auto row_of = [&](int i, int r) { return min(mw + i * 16 + r, g.M - 1); };      // clamped: a valid address, not stored
                    auto load_rows = [&](int i, Rows& R) {
                        if constexpr (RESID) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float* xp = (const float*)g.out + (int64_t)row_of(i, r) * g.ldo;
#pragma unroll
                                for (int j = 0; j < 8; ++j) {
#if WAN_DEV_EXPERIMENTS
                                    if (g.exp & 128) { R.x[j][r] = 0.f; continue; }          // timing only: no residual loads
#endif
                                    R.x[j][r] = xp[nn[j]];
                                }
                            }
                        }
                    };
                    auto store_rows = [&](int i, const Rows& R) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = mw + i * 16 + r;
                            if (m >= g.M) continue;
                            float* op = (float*)g.out + (int64_t)m * g.ldo;
                            const float* gp = SEAM ? g.gate + (int64_t)(seam_b_lo + (m >= seam_next ? 1 : 0)) * g.N : nullptr;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                if (!nok[j]) continue;
                                const float v = acc[i][j][r] + bq[j];
#if WAN_DEV_EXPERIMENTS
                                if (g.exp & 256) { asm volatile("" :: "v"(R.x[j][r] + v * gq[j])); continue; }      // timing only: no stores
#endif
                                if constexpr (RESID) op[nn[j]] = R.x[j][r] + v * (SEAM ? gp[nn[j]] : gq[j]);
                                else op[nn[j]] = v;
                            }
                        }
                    };
                    Rows R[2];
                    load_rows(0, R[0]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (i + 1 < 8) load_rows(i + 1, R[(i + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        store_rows(i, R[i & 1]);
                    }
                };
                using T = std::true_type; using F = std::false_type;
                if constexpr (EPI == WAN_EPI_RESID_F32) {
                    if (g.gate) {
                        // the gate row of a token row is that of its sample, row / rows_per_batch: one (scalar) division per wave and tile;
                        // rows_per_batch >= 128 (the host sends anything else to the per-tile kernels), so the 128 rows of a wave meet at
                        // most one boundary: the wave that holds it runs the SEAM copy (gate fetched per row)
                        seam_b_lo = __builtin_amdgcn_readfirstlane(min(m0 + wr * 128, g.M - 1) / (int)g.rows_per_batch);
                        seam_next = (seam_b_lo + 1) * (int)g.rows_per_batch;
                        const bool one_sample = min(m0 + wr * 128 + 127, g.M - 1) < seam_next;       // wave-uniform
                        if (one_sample) { if (g.bias) epilogue_f32(T{}, T{}, F{}); else epilogue_f32(F{}, T{}, F{}); }
                        else { if (g.bias) epilogue_f32(T{}, T{}, T{}); else epilogue_f32(F{}, T{}, T{}); }
                    } else {
                        if (g.bias) epilogue_f32(T{}, F{}, F{}); else epilogue_f32(F{}, F{}, F{});
                    }
                } else {
                    if (g.bias) epilogue_f32(T{}, F{}, F{}); else epilogue_f32(F{}, F{}, F{});
                }
            } else {
                // Transposed store: D = mfma(A fragment, W fragment): lane (l15, kg) holds output column n = .. + l15 and the 4 consecutive
                // rows m = .. + 4 kg .. + 3 of every tile
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = n0 + wc * 128 + j * 16 + l15;
                    const bool nok = n < g.N;
                    const float bv = (g.bias && nok) ? g.bias[n] : 0.f;
                    if (!nok) continue;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int m = m0 + wr * 128 + i * 16 + l4;
                        if (m >= g.M) continue;
                        const f32x4 a = acc[i][j];
                        bf16_t* p = (bf16_t*)g.out + (int64_t)n * g.ldo + m;
                        const float v0 = a[0] + bv, v1 = a[1] + bv, v2 = a[2] + bv, v3 = a[3] + bv;
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

        GP_STAMP(t_epi);
#if WAN_DEV_EXPERIMENTS
        if (g.exp & 1024) {     // TIMING ONLY: wait for the epilogue's own stores (and loads) to complete -> t_drain
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GP_STAMP(t_drain);
        }
#endif
        // ---- advance the stream
        if (!nxt.valid) break;
        cur = nxt; pc = pn;
        if (!dp_done && g.dynamic) {           // hand the pending ticket round, draw the one after it
            if (tid == 0) lds_ticket[1] = pending;
            __syncthreads();
            const int tk = __builtin_amdgcn_readfirstlane(lds_ticket[1]);
            __syncthreads();                   // everybody has read it before thread 0 may overwrite it at the next switch
            nxt = uniform(produce(tk));
            if (tid == 0) pending = draw();     // (a draw past the end is harmless; testing dp_done here makes hipcc treat the whole segment state as divergent)
        } else {
            nxt = uniform(produce(0));
        }
        pn = panel(nxt);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // zero-length requests past the last segment still write LDS: let them finish
#if WAN_DEV_EXPERIMENTS
    if ((g.exp & 64) && tid == 0 && blockIdx.x < 64) {          // counters words [640, 1024) are free
        int* const o_ = g.counters + 640 + 5 * blockIdx.x;
        o_[0] = (int)(t_start >> 4); o_[1] = (int)(t_loop >> 4); o_[2] = (int)(t_epi >> 4); o_[3] = (int)(t_wait >> 4); o_[4] = (int)(t_drain >> 4);
    }
#endif
#undef GP_STAMP
#undef GP_MFMA
#undef GP_MFMA_A
#undef GP_MFMA_V
#undef GP_SB
}

template <int EPI, int FORM, bool FP8 = false>
wan_status_t launch_pk(const PkArgs& g, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};
    const wan_status_t st = wan_once_per_device(attr_done, +[]() -> wan_status_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pk_kernel<EPI, FORM, FP8>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes + 64);
        if (e != hipSuccess) {
            wan_set_error("wan_gemm_bf16_ws: cannot reserve %d B of LDS: %s", kLdsBytes + 64, hipGetErrorString(e));
            return WAN_ERR_LAUNCH;
        }
        return WAN_OK;
    });
    if (st != WAN_OK) return st;
    if (hipMemsetAsync(g.counters, 0, kCounterBytes, s) != hipSuccess) {
        wan_set_error("wan_gemm_bf16_ws: cannot clear the arrival counters: %s", hipGetErrorString(hipGetLastError()));
        return WAN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((gemm_pk_kernel<EPI, FORM, FP8>), dim3((unsigned)g.nworkers), dim3(kThreads), kLdsBytes + 64, s, g);
    WAN_CHECK_LAUNCH("wan_gemm_bf16_ws");
    return WAN_OK;
}

}  // namespace

// grid of the persistent form: one workgroup per CU, a multiple of 8 (one lane set per XCD), never more than the tile count / 1
int wan_gemm_pk_workers(int M, int N) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    int w = wan_cu_count() & ~7;
    if (const int t = wan_tune(WAN_TUNE_GEMM_PK_WORKERS); t > 0) w = t & ~7;
    (void)tiles;
    // the 4 KiB counter header holds the arrival counters in words [0, workers) and the per-XCD ticket counters from word 512 on
    return w < 8 ? 8 : (w > kMaxWorkers ? kMaxWorkers : w);
}

int64_t wan_gemm_pk_workspace_bytes(int M, int N) {
    return kCounterBytes + (int64_t)wan_gemm_pk_workers(M, N) * 2 * kSlotBytes;
}

static void pk_plan_args(PkArgs& g, int M, int N, int K, int ktile = BK) {
    g.M = M; g.N = N; g.K = K; g.ktile = ktile;
    g.sa = g.sw = nullptr;
    g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
    // M tiles per rasterisation group = the shape of the 32 tiles in flight on an XCD (GM x 32 / GM): GM + 32 / GM distinct operand panels feed
    // 64 panel reads, so 4 x 8 is the minimum.  Round-6 sweep on the 14B shapes (profiles/r06/gemm_gm_sweep.log, gemm_gm_fetch.log): the L2 ->
    // fabric reads follow that count exactly (ffn.0: GM 2 22.8 GB, 4 15.5, 6 15.1, 16 22.1 per launch) and the TIME does not (GM 2 / 3 / 4 / 6
    // within 0.6 % on every shape; 8 loses 2-3 %, 16 loses 8 %): the Linears are not waiting for the fabric.  4 for the wide outputs (q|k +1.1 %,
    // ffn.0 +0.1 %, a third less fabric traffic), 3 otherwise (o / v / ffn.2 are at their best there).
    g.gm = g.tiles_n >= 40 ? 4 : 3;
    if (const int gm = wan_tune(WAN_TUNE_GEMM_GM); gm > 0) g.gm = gm;
    g.nworkers = wan_gemm_pk_workers(M, N);
    // smallest range worth a worker: a quarter of a tile's K range (a tile is then cut into at most ~5 pieces), at least one unit
    g.min_units = (K / ktile / 2 + 3) / 4;
    // ... unless cutting costs more than it saves.  The S leftover tiles of an XCD's slab (S = slab mod W lanes) either go WHOLE to S lanes
    // while W - S lanes idle for one tile -- (W - S) / W x n2 units of a lane's time (n2 = units per tile) -- or are cut stream-K fashion, whose
    // fix-up (the partial tiles' 256 KiB round trips through the workspace, the arrival counter, the last arriver's serial combine ahead of
    // its epilogue) was measured at ~ 9 + 0.36 n2 units (profiles/r06/gemm_split.log: whole tiles +7...+10 % on the 660 tiles of the 8-way
    // Ulysses shard's N = K = 5 120 Linears, +4 % at its K = 13 824, +4...+5 % on the 1.3B model's K = 1 536 ones, -4 % where a slab leaves
    // 5 of 32 lanes' worth).  Whole tiles (min_units = n2: range j is exactly tile j) when the idle time is the smaller price -- for launches of
    // at most 16 rounds only: behind 20+ rounds of tiles drawn by ticket the lanes no longer end together and either form is within +-1 %
    // (the 14B Linears at M = 67 080 stay as they were).
    {
        const int n2 = K / ktile / 2, W = g.nworkers >> 3;
        const int slab = g.tiles_m * g.tiles_n / 8, S = slab % W;
        if (S > 0 && n2 > 0 && slab / W <= 16 && (int64_t)(W - S) * n2 * 100 < (int64_t)W * (900 + 36 * n2)) g.min_units = n2;
    }
    if (const int mu = wan_tune(WAN_TUNE_GEMM_PK_MIN_UNITS); mu > 0) g.min_units = mu;
    g.dynamic = wan_tune(WAN_TUNE_GEMM_PK_ORDER) != 1;
    g.form = wan_tune(WAN_TUNE_GEMM_PK_FORM);      // bit e: epilogue e (WAN_EPI_*) runs from the row-permuted tile
    g.exp = wan_tune(WAN_TUNE_GEMM_EXP);
}

// Host arithmetic only: segment `index` of worker `worker` of the persistent GEMM's plan for this shape -- the SAME functions the
// kernel evaluates (tests: every (tile, K tile) is covered exactly once, pieces and slots are consistent).
// out[11] = tm, tn, kb, ke, partial, slot, counter, j_lo, j_hi, me, tau; returns 1 when the segment exists, 0 past the end.
extern "C" int wan_gemm_pk_segment(int M, int N, int K, int worker, int index, int* out) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 128 != 0 || worker < 0 || index < 0 || out == nullptr) return 0;
    PkArgs g;
    pk_plan_args(g, M, N, K);
    if (worker >= g.nworkers) return 0;
    const Slab slab = make_slab(g, worker, g.ktile);
    const Seg e = get_seg(g, slab, index);
    const int v[11] = {e.tm, e.tn, e.kb, e.ke, e.partial, e.slot, e.cnt, e.j_lo, e.j_hi, e.me, e.tau};
    for (int i = 0; i < 11; ++i) out[i] = v[i];
    return e.valid;
}

extern "C" int wan_gemm_pk_grid(int M, int N) { return wan_gemm_pk_workers(M, N); }

// called by wan_gemm_bf16_ws (gemm_bf16.hip) for shapes the 4-wave 256^2 kernel takes; arguments already validated there
wan_status_t wan_gemm_bf16_pk(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                              void* out, int64_t ldo, int M, int N, int K, int epilogue,
                              const float* gate, int64_t rows_per_batch, void* workspace, hipStream_t s) {
    PkArgs g;
    g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.bias = bias;
    g.out = out; g.ldo = ldo; g.gate = gate; g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    pk_plan_args(g, M, N, K);
    WAN_REQUIRE(g.nworkers <= kMaxWorkers && g.nworkers % 8 == 0, WAN_ERR_INVALID, "wan_gemm_bf16_ws: %d workers (at most %d, a multiple of 8)", g.nworkers, kMaxWorkers);
    g.counters = (int*)workspace;
    g.slots = (char*)workspace + kCounterBytes;
#define WAN_PK(E) (((g.form >> (E)) & 1) ? launch_pk<E, 1>(g, s) : launch_pk<E, 0>(g, s))
    switch (epilogue) {
        case WAN_EPI_BF16: return WAN_PK(WAN_EPI_BF16);
        case WAN_EPI_GELU_BF16: return WAN_PK(WAN_EPI_GELU_BF16);
        case WAN_EPI_F32: return WAN_PK(WAN_EPI_F32);
        case WAN_EPI_RESID_F32: return WAN_PK(WAN_EPI_RESID_F32);
        case WAN_EPI_BF16_T: return WAN_PK(WAN_EPI_BF16_T);
        default: wan_set_error("wan_gemm_bf16_ws: unknown epilogue %d", epilogue); return WAN_ERR_INVALID;
    }
#undef WAN_PK
}

// the e4m3 instantiation (wan_gemm_fp8_ws, gemm_bf16_256.hip): same plan with K tiles of 128 elements; arguments already validated there
wan_status_t wan_gemm_fp8_pk(const void* A, int64_t lda, const float* a_row_scale, const void* W, int64_t ldw, const float* w_row_scale,
                             const float* bias, void* out, int64_t ldo, int M, int N, int K, int epilogue,
                             const float* gate, int64_t rows_per_batch, void* workspace, hipStream_t s) {
    PkArgs g;
    g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.bias = bias;
    g.out = out; g.ldo = ldo; g.gate = gate; g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    pk_plan_args(g, M, N, K, 2 * BK);
    g.sa = a_row_scale; g.sw = w_row_scale;
    WAN_REQUIRE(g.nworkers <= kMaxWorkers && g.nworkers % 8 == 0, WAN_ERR_INVALID, "wan_gemm_fp8_ws: %d workers (at most %d, a multiple of 8)", g.nworkers, kMaxWorkers);
    g.counters = (int*)workspace;
    g.slots = (char*)workspace + kCounterBytes;
    switch (epilogue) {
        case WAN_EPI_BF16: return launch_pk<WAN_EPI_BF16, 1, true>(g, s);
        case WAN_EPI_GELU_BF16: return launch_pk<WAN_EPI_GELU_BF16, 1, true>(g, s);
        case WAN_EPI_F32: return launch_pk<WAN_EPI_F32, 1, true>(g, s);
        case WAN_EPI_RESID_F32: return launch_pk<WAN_EPI_RESID_F32, 1, true>(g, s);
        case WAN_EPI_BF16_T: return launch_pk<WAN_EPI_BF16_T, 1, true>(g, s);
        default: wan_set_error("wan_gemm_fp8_ws: unknown epilogue %d", epilogue); return WAN_ERR_INVALID;
    }
}

// segment `index` of worker `worker` of the e4m3 plan (K tiles of 128 elements): see wan_gemm_pk_segment
extern "C" int wan_gemm_fp8_pk_segment(int M, int N, int K, int worker, int index, int* out) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 256 != 0 || worker < 0 || index < 0 || out == nullptr) return 0;
    PkArgs g;
    pk_plan_args(g, M, N, K, 2 * BK);
    if (worker >= g.nworkers) return 0;
    const Slab slab = make_slab(g, worker, g.ktile);
    const Seg e = get_seg(g, slab, index);
    const int v[11] = {e.tm, e.tn, e.kb, e.ke, e.partial, e.slot, e.cnt, e.j_lo, e.j_hi, e.me, e.tau};
    for (int i = 0; i < 11; ++i) out[i] = v[i];
    return e.valid;
}
