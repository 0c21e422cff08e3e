// Flash-style non-causal attention for head_dim 128 on gfx950 (the kernel that replaces the
// flash_attn / sageattention wheels behind attention(), attention_utils.py:152-211).
//
// Workgroup = 4 waves x 64 query rows (two 32-query blocks per wave) = 256 queries of one (batch, head), one wave per SIMD
// with the whole 512-register file; KV tiles of 64 keys are streamed through a double-buffered LDS ring by LDS-DMA.
// Per wave, query block and KV tile:
//
//   S^T[key, q]  = K[key, :] . Q[q, :]        16 x v_mfma_f32_32x32x16_bf16  ("swapped" QK^T)
//   online softmax in registers: lane (q = l&31, hi = l>>5) owns 32 of the 64 scores of its query
//   O^T[d, q]   += V^T[d, key] . P^T[key, q]   16 x v_mfma_f32_32x32x16_bf16
//
// With the swapped product every lane owns ONE query column of S^T and of O^T, so the running
// max / sum and the rescale of O are lane-local; the only cross-lane traffic per tile is one
// exchange of the row max between lanes l and l^32 (one v_permlane32_swap, no LDS round trip).
//
// Key order: the MFMA leaves score row i = (r&3) + 8*(r>>2) + 4*hi in register r.  K rows are
// fetched from LDS through the permutation pi (swap bits 2 and 3 of the row index), which makes
// registers 8t..8t+7 of a lane hold the 8 CONSECUTIVE keys 16t + 8*hi .. +7 -- exactly the B-operand
// fragment of the P.V product -- so P never moves between lanes.
//
// V is consumed transposed (V^T[d][key], produced by the V-projection GEMM epilogue), so both
// MFMA operands of both products are 16-byte ds_read_b128 of k-contiguous data.
// LDS images (LDS-DMA lands lane-linear; the swizzle is applied on the source address and mirrored
// on the read):  K tile [64 keys][128 d]: 256-B rows, chunk' = chunk ^ (row & 15)
//                V^T tile [128 d][64 keys]: 128-B rows, chunk' = chunk ^ ((row >> 1) & 7)
#include <stdlib.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <type_traits>

#include "common.hpp"

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int kD = 128;          // head dim
constexpr int kQPerWave = 64;    // two 32-query blocks per wave
constexpr int kWavesPerWG = 4;   // one wave per SIMD
constexpr int kQPerWG = kQPerWave * kWavesPerWG;   // 256
constexpr int kKV = 64;          // keys per tile
constexpr int kThreads = kWavesPerWG * 64;
constexpr int kKTileBytes = kKV * kD * 2;    // 16 KiB
constexpr int kVTileBytes = kD * kKV * 2;    // 16 KiB

struct AttnArgs {
    const bf16_t* q; int64_t ldq, q_bs;
    const bf16_t* k; int64_t ldk, k_bs;
    const bf16_t* vt; int64_t ldvt, vt_bs;
    bf16_t* o; int64_t ldo, o_bs;
    int Lq, Lk, H;
    float scale_log2e;    // softmax_scale * log2(e)
    // split-KV tail launch (see the launcher): query blocks start at qblk0; blockIdx.z = batch * nsplit + split
    int qblk0, nsplit, tiles_per_split, row0, rows_tail;
    float* ws_o;          // [batch][nsplit][H][rows_tail][128] un-normalised partial outputs
    float* ws_ml;         // [batch][nsplit][H][rows_tail][2]   running max (log2 units), sum
    int* flags;           // scratch words behind the 16-byte header: one per workgroup of the main launch, != 0 -> the fix-up launch redoes it
    // 1-D grid decode: workgroup w -> (query block, head, z) with z = batch (or batch * nsplit + split).
    // xcd_map != 0 (needs nbh % 8 == 0): workgroups are dispatched round-robin over the 8 XCDs, so w & 7 IS the XCD;
    // (head, z) pair bh = 8 * ((w >> 3) / nqb) + (w & 7) pins every head to one XCD, whose 32 CUs walk that head's query
    // blocks together: one head's K / V^T (34 MB at L = 67 080) streams through ONE 4 MB L2 instead of all eight.
    int nqb, nbh, xcd_map;
    int nwg;              // nqb * nbh: the blocks a PERSIST launch walks (its grid is one workgroup per CU)
    // QK8 form (wan_attention_fwd_qk8): q, k as OCP e4m3 bytes [rows][H*128], pre-multiplied by powers of two; the MFMA
    // applies the inverse (q8_scale / k8_scale: the E8M0 byte 127 - exponent, replicated into all four bytes)
    const unsigned char* q8; int64_t ldq8, q8_bs;
    const unsigned char* k8; int64_t ldk8, k8_bs;
    unsigned q8_scale, k8_scale;
    // fp8 P.V form (wan_attention_fwd_f8): V^T as MX e4m3 [B][H*128][ldv8 bytes] (keys of a 64-tile in register order) and its
    // scale bytes [B][H][tiles][64 lanes][4 d-blocks] (vs8_hs = bytes per head), both written by wan_vt_quantize_mx
    const unsigned char* v8; int64_t ldv8, v8_bs;
    const unsigned char* vs8; int64_t vs8_hs;
    // packed ragged batches (wan_attention_fwd_varlen): keys of batch b end at klens[b] (device memory, clamped to [1, Lk]); NULL =
    // every batch attends Lk keys.  The reference packs such batches behind cu_seqlens (attention_utils.py:95-146)
    const int* klens;
};

// VARIANT (template parameter of the kernels below) only names the instantiation so profiles separate the two
// call sites:  0 = attn_self (long KV stream: self-attention, Lk ~ 1e4..1e5),
//              1 = attn_cross (short KV: the 512 text tokens of WanT2VCrossAttention).
// History: rounds 1-3 also carried an 8-wave kernel here (8 waves x 32 query rows, two waves per SIMD, running max per tile;
// 1.18-1.28 PFLOP/s) -- the product kernel of round 1 and from round 3 on only the developer A/B partner behind "attn_w4" = 0.  It was
// retired in round 5 (no product path reached it); what it taught is kept in DESIGN.md section 4.1 and in the A/B logs under
// profiles/r01 - r03.  Its workgroup geometry survives in the constants above (a workgroup is still 256 queries of one head,
// KV tiles are still 64 keys, the LDS images and the key permutation are the ones described at the top of this file).


// 32 scores -> 1.  Written as max(max(m, a), b) so that the DAG combiner has exactly one way to fuse each
// step into v_max3_f32 (15 v_max3 + 1 v_max); max(m, max(a, b)) makes it fuse the wrong pair and emit 31 ops.
// No inline asm here: an asm v_max3 reading MFMA results is invisible to the hazard recogniser (measured:
// missing wait states, run-to-run different maxima).
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// max(x, x of lane ^ 32) with one v_permlane32_swap (gfx950) instead of a ds_bpermute round trip through the LDS
// pipe at the head of every tile.  Inline asm because hipcc folds __builtin_amdgcn_permlane32_swap(x, x) to x;
// both operands come from plain VALU results (the v_max3 chain), the leading s_nop covers the VALU-write ->
// permlane-read wait states the compiler inserts for its own uses of the instruction.
__device__ __forceinline__ float max_with_lane_xor32(float x) {
    float y = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return fmaxf(x, y);
}
__device__ __forceinline__ float rowmax32(const f32x16 (&s)[2]) {
    float m0 = max3f(s[0][0], s[0][1], s[0][2]), m1 = max3f(s[1][0], s[1][1], s[1][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) {
        m0 = max3f(m0, s[0][r], s[0][r + 1]);
        m1 = max3f(m1, s[1][r], s[1][r + 1]);
    }
    return max3f(max3f(m0, s[0][15], s[1][15]), m1, m1);
}

// ====================================================================================================
// 4-wave kernel: every product launch of wan_attention_fwd (round 2: the max-free main launch only).
//
// Same workgroup (256 queries of one (batch, head)), same LDS images, same products and key permutation as the 8-wave
// kernel above, but FOUR waves of 64 query rows each -- one wave per SIMD with the whole 512-register file:
//   * O^T accumulators (2 query blocks x 4 d-blocks x 16) and the Q fragments (2 x 8 x 4) live in AGPRs
//     (192 of 256); the MFMAs are inline asm so that A/B/C operands can be named in either file -- hipcc's own
//     allocation of this shape spills (576 B/lane) and copies every S tile through v_accvgpr_read;
//   * every K / V^T fragment read from LDS feeds TWO MFMAs (one per query block): 32 ds_read_b128 per 64 MFMAs instead
//     of 32 per 32, and fragments are requested 3 fragments (6+ MFMA slots) ahead of use -- the 8-wave kernel reads each
//     fragment right before its MFMA (no registers left) and its waves sit in s_waitcnt lgkmcnt 31 % of the time;
//   * with a single instruction stream per SIMD the order of the stream IS the schedule: one interval is 64 slots of
//     `MFMA ; <= 5 fillers ; sched_barrier(0)` (attn_w4_sched.inc, emitted by tools/gen_attn_w4_sched.py).
// Hazards that hipcc does not see for an asm MFMA are handled by construction (CDNA4 guide, section 5.7):
//   * VALU-written P fragment -> MFMA B operand: every P fragment is complete at least one slot (an MFMA and a
//     sched_barrier) before the first PV slot that reads it (asserted by the generator);
//   * MFMA-written S -> VALU read: S(t+1) is produced in interval t and first read in interval t+1, behind the
//     s_waitcnt + s_barrier of the fence; the prologue's S(0) and the final O are followed by an explicit s_nop;
//   * accumulate chains (same D as C) have 4 (S) / 8 (O) independent MFMAs between links, far above the 43-cycle cliff.
// Three forms of the softmax reference (template parameter REF, below): max-free with a checked window, and two lazy forms.
// ====================================================================================================
constexpr int kW4Threads = 256;

__device__ __forceinline__ u32x4 lds_read16_at(unsigned lds_byte_address) {
    return *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)lds_byte_address;
}

// Pipeline: K(t+2) / V(t+1) are requested (LDS-DMA) during the S segment of interval t and waited for (vmcnt(0) + barrier)
// at its end; rings K 2 x 16 KiB + V 2 x 16 KiB, as in the 8-wave kernel.  Measured alternative that did not pay
// (profiles/r02/attn_w4_ab.log): requesting K(t+3) / V(t+2) in the PV segment, waiting for them at ONE barrier between the two
// segments of the next interval (V ring of 3) and fetching the first K fragments of an interval before the previous one
// ends -- no LDS latency exposed at the seam -- 68.7 vs 68.5 ms: the seam is not where the time goes.
// Also without effect: moving one softmax micro-op between the two slots of a fragment pair (gaps of 3|4 instead of 2|5
// fillers): 66.9 vs 66.8 ms.  At 85 % matrix-pipe occupancy the kernel sits on the chip's power limit (sustained clock
// 1.66 GHz, profiles/r02/attn_w4_pmc.json); what is left is energy per tile, not issue slots.
constexpr int kLdsBytesW4 = 2 * kKTileBytes + 2 * kVTileBytes;

// MAXFREE (REF = 0): softmax reference 0 -- p = exp2(S) directly; bf16 P / fp32 sums have head-room for log2-domain scores in
// about [-90, 100].  The assumption is CHECKED: a row whose sum leaves [2^-90, 2^100] (or is not finite) flags its workgroup
// in the caller's scratch, and the FIX launch right behind (the lazy form below, same grid) redoes exactly the flagged
// workgroups; a sticky word in the scratch turns the attempt off once more than 1/8 of a launch had to be redone.
// MAXFREE = false: the SAME loop with a LAZY softmax reference, which needs no second launch and has no input-dependent
// cliff (1.36 vs 1.40 PFLOP/s for the max-free attempt at L = 67 080 x 40 heads, profiles/r03/attn_lazy_ab.log):
//   * the reference m of a query row starts as the exact row max of tile 0 and rides in the MFMA accumulator (every S chain
//     starts from a 16-register splat of -m, as in the 8-wave PRE form), so p = exp2(S') costs no extra VALU;
//   * softmax does not care WHICH reference is used as long as nothing overflows or vanishes, and bf16 P / fp32 sums have
//     ~2^127 of head-room: instead of a row max per tile (32 v_max3 + a lane exchange + a branch), the loop tests the row
//     SUMS it computes anyway -- one v_max + one v_cmp per tile at MIDCHECK, where all four P fragments of the tile exist
//     and none has been consumed.  While every lane's tile sum stays <= 2^40 nothing happens (m <= true max always holds, so
//     no term that matters can vanish);
//   * a sum above 2^40 (or Inf/NaN: a score more than 127 above m) takes the wave-uniform repair path: exact row max of the
//     tile, O / l / the already accumulated next-tile scores rescaled to the new reference, the tile's P recomputed.  Every
//     repair raises m by > 2^34, so even adversarial inputs repair a handful of times per row, not per tile.
// SPLIT: the split-KV tail round (un-normalised O, m, l to the caller's scratch; see plan_tail).
// Plain (not pre-scaled) q runs REF = 2, whose packed fma applies softmax_scale * log2(e) to the fp32 scores exactly (folding it
// into the bf16 Q fragments instead was measured: one more rounding of q moves scores of magnitude ~500 by ~1, rel-L2 1.6e-2).
constexpr float kW4Trigger = 0x1p40f;

// REF: 0 = max-free, 1 = lazy reference riding in the accumulator (-m splats, 32 VGPRs, no extra VALU),
//      2 = lazy reference subtracted from the scores two at a time (v_pk_add_f32: 4 VGPRs, +32 VALU per tile).
// FIX: the launch right behind a max-free attempt on the same grid -- only flagged workgroups run (all others exit at once).
// QK8: S^T = K.Q^T on the fp8 matrix pipe (v_mfma_scale_f32_32x32x64_f8f6f4: 2 x the bf16 rate; 8 x 16-pass MFMAs per tile
//      instead of 32 x 8-pass), from e4m3 copies of q and k that the RMSNorm+RoPE kernel writes (wan_rmsnorm_rope_fp8); softmax,
//      P and the P.V product are the bf16 / fp32 ones of the REF = 1 form.  LOSSY (3 mantissa bits on q and k), opt-in.
//      K tile image: [64 keys][128 B], chunk' = chunk ^ ((row >> 1) & 7) (the V^T image's swizzle), 2 DMA pieces per wave; a lane's
//      A fragment (kt, dh) = the 32 bytes d = 64 dh + 32 hi .. + 31 of key pi(lane & 31) + 32 kt, read as two ds_read_b128 (which k of
//      the MFMA a byte lands on does not matter here: q and k use the same order and the operand scales are uniform).
// PERSIST (round 6; the cross-attention launch, VARIANT 1): ONE resident workgroup per CU walks query blocks w = blockIdx, blockIdx +
// grid, ... instead of one workgroup per block.  A cross-attention block is only 8 KV tiles (512 text rows): per block the one-shot
// form spends as long in what surrounds the tiles -- dispatch, the Q fragments' trip from HBM, K(0) / V(0) landing, 32 output stores
// per lane draining -- as in the tiles themselves (0.99 ms per launch = 0.28 of peak).  Here the NEXT block's K(0) / V(0) / K(1)
// requests and its Q fragment loads are issued before the current block's output stores (Q's registers are dead by then; vector memory
// operations complete in issue order, so `s_waitcnt vmcnt(16)` at the top of the next block waits for exactly those requests and
// leaves the 16 stores in flight), and the stores go through a per-block buffer descriptor whose range check drops rows >= Lq -- every
// lane always issues all 16 (16 bytes each, after the two lanes of a row have traded halves), which is what makes the count exact.
template <int VARIANT, bool SPLIT, int REF, bool FIX = false, bool QK8 = false, bool PERSIST = false>
__global__ __launch_bounds__(kW4Threads) __attribute__((amdgpu_waves_per_eu(1, 1)))
void attn_fwd_w4_kernel(AttnArgs a) {
    constexpr bool MAXFREE = REF == 0, SPLAT = REF == 1, PKSUB = REF == 2;
    static_assert(!PERSIST || (!SPLIT && !FIX && !QK8 && !MAXFREE), "the persistent form is the plain lazy-reference kernel");
    static_assert(!QK8 || (SPLAT && VARIANT == 0), "the fp8 QK^T form is the self-attention lazy-reference kernel");
    static_assert(!(SPLIT && MAXFREE), "the split tail round runs the lazy-reference form");
    static_assert(!FIX || (!SPLIT && !MAXFREE), "the fix-up launch is the unsplit lazy-reference form");
    const int wg_linear = blockIdx.x;
    if constexpr (FIX) {
        if (a.flags[wg_linear] == 0) return;          // workgroup-uniform
        if (threadIdx.x == 0) {
            // if more than 1/8 of a launch had to be redone, later calls on this scratch skip the max-free attempt
            int* const hdr = a.flags - 4;
            const int redone = atomicAdd(&hdr[1], 1) + 1;
            if (redone * 8 > (int)gridDim.x) hdr[0] = 1;
        }
    }
    if constexpr (MAXFREE) {
        int* const hdr = a.flags - 4;
        if (wg_linear == 0 && threadIdx.x == 0) hdr[1] = 0;
        if (hdr[0] != 0) {                                       // sticky "fast path off": hand everything to the fix-up launch
            if (threadIdx.x == 0) a.flags[wg_linear] = 1;
            return;
        }
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    // ---- the query block this workgroup works on (PERSIST: re-evaluated for every block it walks).  Everything below that depends on
    // the block -- operand origins, key count, the Q fragments -- is state that `setup_block` / `load_q` rewrite.
    int qblk, batch, head, split, t0, Lk, nkv;
    const bf16_t *Q, *K, *VT;
    const unsigned char* K8;
    bf16_t* O;
    auto setup_block = [&](int w) __attribute__((always_inline)) {
        int bh;
        if (a.xcd_map) {
            const int s_ = w >> 3, g_ = s_ / a.nqb;
            qblk = s_ - g_ * a.nqb;
            bh = g_ * 8 + (w & 7);
        } else {
            bh = w / a.nqb;
            qblk = w - bh * a.nqb;
        }
        const int bz = bh / a.H;
        head = bh - bz * a.H;
        batch = SPLIT ? bz / a.nsplit : bz;
        split = SPLIT ? bz - batch * a.nsplit : 0;
        t0 = split * a.tiles_per_split;                     // first KV tile of this split
        if constexpr (SPLIT) qblk += a.qblk0;
        if constexpr (SPLIT) Lk = min(a.Lk - t0 * kKV, a.tiles_per_split * kKV);
        else Lk = a.klens != nullptr ? max(1, min(a.klens[batch], a.Lk)) : a.Lk;        // scalar: one s_load per workgroup
        nkv = (Lk + kKV - 1) / kKV;
        Q = a.q + batch * a.q_bs + head * kD;
        K = a.k + batch * a.k_bs + head * kD + (int64_t)t0 * kKV * a.ldk;
        K8 = QK8 ? a.k8 + batch * a.k8_bs + head * kD + (int64_t)t0 * kKV * a.ldk8 : nullptr;
        VT = a.vt + batch * a.vt_bs + (int64_t)head * kD * a.ldvt + t0 * kKV;
        O = a.o + batch * a.o_bs + head * kD;
    };
    setup_block(wg_linear);

    // ---- Q fragments of the wave's two query blocks (B operands of S^T = K.Q^T), kept in AGPRs by the "a" constraints
    int qrow[2];
    u32x4 qf[2][8];
    i32x8 qf8[2][2];            // QK8: B operands, 32 e4m3 per lane and d half (d = 64 dh + 32 hi .. + 31)
    auto load_q = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            qrow[qb] = qblk * kQPerWG + wid * 64 + qb * 32 + l31;
            if constexpr (QK8) {
                const unsigned char* qp = a.q8 + batch * a.q8_bs + head * kD + (int64_t)min(qrow[qb], a.Lq - 1) * a.ldq8 + hi * 32;
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    const u32x4 lo = *reinterpret_cast<const u32x4*>(qp + dh * 64), up = *reinterpret_cast<const u32x4*>(qp + dh * 64 + 16);
                    qf8[qb][dh] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)up[0], (int)up[1], (int)up[2], (int)up[3]};
                }
            } else {
                const bf16_t* qp = Q + (int64_t)min(qrow[qb], a.Lq - 1) * a.ldq + hi * 8;
                if constexpr (PERSIST) {
                    // asm loads, invisible to hipcc's waitcnt bookkeeping ON PURPOSE: tracked, the first use of every fragment gets a
                    // wait sized for the worst predecessor (the first block, where only the K / V requests follow the loads) and the
                    // S(0) product of every later block then waits for most of the previous block's output stores.  The loads are
                    // older than those stores; the `s_waitcnt vmcnt(16)` (vmcnt(0) for the first block) at the top of the block covers them.
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(qf[qb][ks]) : "v"(qp), "i"(ks * 32) : "memory");
                } else {
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) qf[qb][ks] = *reinterpret_cast<const u32x4*>(qp + ks * 16);
                }
            }
        }
    };
    load_q();
    char* const kring = smem;
    // ---- staging: this wave copies pieces 4*wid .. 4*wid+3 (1 KiB each) of every K and V^T tile
    // K / V^T tiles are fetched with `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer descriptor): the tile origin
    // is SCALAR state (descriptor base advanced per tile by SALU), each lane keeps one constant 32-bit byte offset per
    // piece, and the descriptor's range check returns zeros for K rows >= Lk (they are masked in the peeled last tile)
    // -- no per-lane address arithmetic and no clamp in the loop.  Piece j of this wave = K rows 16 wid + 4 j + lane/16
    // (256-B rows, chunk' = chunk ^ (row & 15)) and V^T rows 32 wid + 8 j + lane/8 (128-B rows, chunk' = chunk ^ ((row>>1)&7)).
    int k_voff[4], v_voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (QK8) {        // 8 pieces of 8 rows x 128 B; this wave stages pieces 2 wid, 2 wid + 1 (j < 2)
            const int kr = (wid * 2 + (j & 1)) * 8 + (lane >> 3);
            k_voff[j] = (int)(kr * a.ldk8 + ((lane & 7) ^ ((kr >> 1) & 7)) * 16);
        } else {
            const int kr = wid * 16 + 4 * j + (lane >> 4);
            k_voff[j] = (int)((kr * a.ldk + ((lane & 15) ^ (kr & 15)) * 8) * 2);
        }
        const int vr = (wid * 4 + j) * 8 + (lane >> 3);
        v_voff[j] = (int)((vr * a.ldvt + ((lane & 7) ^ ((vr >> 1) & 7)) * 8) * 2);
    }
    const int64_t k_row_bytes = QK8 ? a.ldk8 : a.ldk * 2;
    const int64_t k_tile_bytes = (int64_t)kKV * k_row_bytes;
    auto k_rsrc = [&](int t) {          // tile min(t, nkv-1): a request past the end re-stages the last tile into a dead slot
        const int tc = min(t, nkv - 1);
        const int64_t left = (int64_t)(Lk - tc * kKV) * k_row_bytes;             // bytes from the tile origin to the end of row Lk-1
        const char* const k_origin = QK8 ? (const char*)K8 : (const char*)K;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(k_origin + tc * k_tile_bytes), 0,
                                                 (int)min(left, (int64_t)0x7fffffff), 0x00020000);
    };
    auto v_rsrc = [&](int t) {          // tile min(t, nkv-1) (t <= nkv - 1 on every call); V^T pad columns exist up to roundup(Lk, 64)
        return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)VT + (int64_t)min(t, nkv - 1) * kKV * 2), 0, 0x7fffffff, 0x00020000);
    };
    auto stage_k_piece = [&](__amdgpu_buffer_rsrc_t r, int kslot, int j) {
        if constexpr (QK8) {
            if (j < 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(kring + kslot * kKTileBytes + (wid * 2 + j) * 1024),
                                                         16, k_voff[j], 0, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(kring + kslot * kKTileBytes + (wid * 4 + j) * 1024),
                                                     16, k_voff[j], 0, 0, 0);
        }
    };
    auto stage_v_piece = [&](__amdgpu_buffer_rsrc_t r, int vslot, int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + 2 * kKTileBytes + vslot * kVTileBytes + (wid * 4 + j) * 1024),
                                                 16, v_voff[j], 0, 0, 0);
    };

    const int pi = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int k_rowoff = pi * 256, k_sw = pi & 15;
    // absolute LDS byte addresses (ds_read operands of both the C++ reads and the inline-asm reads below)
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned k_adr[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) k_adr[ks] = lds_base + k_rowoff + (((2 * ks + hi) ^ k_sw) << 4);
    unsigned k8_adr[4];         // QK8: chunk 4 dh + 2 hi + x of row pi, x = 0, 1 (index 2 dh + x)
#pragma unroll
    for (int i = 0; i < 4; ++i) k8_adr[i] = lds_base + pi * 128 + (((4 * (i >> 1) + 2 * hi + (i & 1)) ^ ((pi >> 1) & 7)) << 4);
    unsigned sc_k, sc_q;        // the MFMA's E8M0 operand scales (VGPR operands; asm so that they are set once, not per MFMA)
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(sc_k), "=v"(sc_q) : "s"(a.k8_scale), "s"(a.q8_scale));
    const int v_rowoff = l31 * 128, v_sw = (l31 >> 1) & 7;
    unsigned v_adr[4];          // carries the V ring base
#pragma unroll
    for (int t = 0; t < 4; ++t) v_adr[t] = lds_base + 2 * kKTileBytes + v_rowoff + (((2 * t + hi) ^ v_sw) << 4);

    auto fence = [&]() {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    };
    // ---- prologue requests: K(0), V(0), K(1) of the block `setup_block` last described
    auto stage_prologue = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { stage_k_piece(k_rsrc(0), 0, j); stage_v_piece(v_rsrc(0), 0, j); }
        if (nkv > 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) stage_k_piece(k_rsrc(1), 1, j);
        }
    };
    stage_prologue();
    int w_cur = wg_linear;              // PERSIST: the block being worked on
    bool first_block = true;            // PERSIST: no output stores of a previous block are in flight
  for (;;) {                            // PERSIST: one iteration per query block of this workgroup; otherwise a single pass
    f32x16 o[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][i][r] = 0.f;
    float l_run[2] = {0.f, 0.f};

    // ---- prologue: K(0), V(0), K(1) in flight; S(0)
    if constexpr (PERSIST) {
        // the requests of THIS block were issued before the previous block's 16 output stores: wait for them, not for the stores
        // (a bare s_barrier: __syncthreads() is a workgroup release fence as well, i.e. `s_waitcnt vmcnt(0)` -- it would wait for the stores)
        if (first_block) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else {
        fence();
    }
    f32x16 s0[2][2], s1[2][2];
// "=&v": an 8-pass MFMA reads A/B over several passes, so D must not share registers with them (hipcc marks its own
// MFMAs early-clobber for the same reason)
#define W4_MFMA0(D, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(D) : "v"(A), "a"(B))
// one accumulator register *= a VGPR factor, through one scratch VGPR (the repair path of the lazy reference): written as
// asm because hipcc's own AGPR <-> VGPR traffic for `o *= alpha` at a point where ~225 VGPRs are live ends in spills
#define W4_SCALE_ACC(ACC, F) do { float t_; asm volatile("v_accvgpr_read_b32 %0, %1\n\tv_mul_f32 %0, %0, %2\n\tv_accvgpr_write_b32 %1, %0" : "=&v"(t_), "+a"(ACC) : "v"(F)); } while (0)
// W4A_*: the A operand (K / V^T fragment) in an AGPR, the lazy-reference form's steady state
#define W4A_MFMA_C(D, A, B, C) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(D) : "a"(A), "a"(B), "v"(C))
#define W4A_MFMA0(D, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(D) : "a"(A), "a"(B))
#define W4A_MFMA_S(D, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(D) : "a"(A), "a"(B))
#define W4A_MFMA_O(D, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(D) : "a"(A), "v"(B))
#define W4_MFMA_S(D, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(D) : "v"(A), "a"(B))
#define W4_MFMA_O(D, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(D) : "v"(A), "v"(B))
// QK8: one 16-pass fp8 MFMA = 64 of the 128 d; the E8M0 scale bytes undo the power-of-two pre-scaling of q8 / k8
#define W8_MFMA0(D, A, B) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel_hi:[0,0,0]" : "=&v"(D) : "v"(A), "a"(B), "v"(sc_k), "v"(sc_q))
#define W8_MFMA_SV(D, A, B) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(D) : "v"(A), "a"(B), "v"(sc_k), "v"(sc_q))
// The loop's K fragments live in FIXED registers a[224 + 8 f .. 231 + 8 f] (f = 2 dh + kt): an 8-register MFMA operand filled by
// two ds_read_b128 has no expression as asm operands (no sub-register syntax; built from two 4-register operands hipcc copies the
// halves through VGPRs -- and does so before the untracked LDS data has arrived).  Every statement that touches them names all
// 32 as clobbered, so hipcc keeps its own AGPR values (O, Q, the V^T ring) out of that range.
#define W8_KCLOB "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", \
                 "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
#define W8_VCLOB "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", \
                 "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223"
#define W8K_MFMA_C(D, F, B, C) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[%5:%6], %1, %2, %3, %4 op_sel_hi:[0,0,0]" : "=&v"(D) : "a"(B), "v"(C), "v"(sc_k), "v"(sc_q), "i"(224 + 8 * (F)), "i"(231 + 8 * (F)) : W8_KCLOB)
#define W8K_MFMA_S(D, F, B) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[%4:%5], %1, %0, %2, %3 op_sel_hi:[0,0,0]" : "+v"(D) : "a"(B), "v"(sc_k), "v"(sc_q), "i"(224 + 8 * (F)), "i"(231 + 8 * (F)) : W8_KCLOB)
    if constexpr (QK8) {
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const u32x4 lo = lds_read16_at(k8_adr[2 * dh] + kt * 32 * 128), up = lds_read16_at(k8_adr[2 * dh + 1] + kt * 32 * 128);
                const i32x8 kf0 = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)up[0], (int)up[1], (int)up[2], (int)up[3]};
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    if (dh == 0) W8_MFMA0(s0[qb][kt], kf0, qf8[qb][0]);
                    else W8_MFMA_SV(s0[qb][kt], kf0, qf8[qb][1]);
                }
            }
    } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const u32x4 kf0 = lds_read16_at(k_adr[ks] + kt * 32 * 256);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    if (ks == 0) W4_MFMA0(s0[qb][kt], kf0, qf[qb][0]);
                    else W4_MFMA_S(s0[qb][kt], kf0, qf[qb][ks]);
                }
            }
    }
    __syncthreads();            // everyone is done with K slot 0 before tile 2 lands in it
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // S(0) is read by VALU code below: cover the MFMA -> VALU wait states
    if constexpr (QK8) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // ... of a 16-pass MFMA

    // ---- lazy reference (see the kernel header): -m of each query block as a 16-register splat = the C operand that starts
    // every S chain.  It starts as the exact row max of tile 0 (of its valid keys when tile 0 is also the last tile).
    f32x16 negm[2];             // SPLAT only
    float nm[2] = {0.f, 0.f};   // -m of the wave's two query blocks (log2 units)
    f32x2 nmp[2];               // PKSUB: (-m, -m), the v_pk_add_f32 operand
    if constexpr (!MAXFREE) {
        if (nkv == 1 && Lk < kKV) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (32 * kt + 16 * (r >> 3) + 8 * hi + (r & 7) >= Lk) s0[qb][kt][r] = -INFINITY;
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float mx = max_with_lane_xor32(rowmax32(s0[qb]));
            nm[qb] = PKSUB ? -mx * a.scale_log2e : -mx;        // softmax_scale > 0 (checked by the launcher): max commutes with the scale
            nmp[qb] = f32x2{nm[qb], nm[qb]};
            if constexpr (SPLAT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[qb][r] = -mx;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s0[qb][kt][r] -= mx;
            }
        }
    }
    // a score as the exponent sees it: SPLAT scores are already relative to m (it rode in the accumulator)
    auto rel = [&](float x, int qb) -> float { return PKSUB ? __builtin_fmaf(x, a.scale_log2e, nm[qb]) : x; };
    const f32x2 cpk = {a.scale_log2e, a.scale_log2e};

    // ---- softmax bookkeeping.  The 160 micro-ops of a tile run as one continuous stream of 2.5 per MFMA slot that starts in
    // the PV segment of the PREVIOUS interval (P fragments tt = 0, 1 -> `pn`, their row sums -> `carry`) and ends in the S
    // segment of the tile's own interval (tt = 2, 3 -> `pf23`).  pa / pb ping-pong between "consumed now" and "produced for
    // the next tile"; tile 0's first half is produced here.
    // QK8: segment B is only 8 (double-length) MFMA slots, so THREE of a tile's four P fragments are produced one interval ahead
    // (tt = 0, 1, 2 -> pn[.][0..2], in the 32 slots of segment C, which have room) and only tt = 3 in the tile's own segment B:
    // the stream is split 120 : 40 like the MFMA time of the two segments (1 024 : 512 cycles) instead of 80 : 80.
    constexpr int kAhead = QK8 ? 3 : 2;         // P fragments of a tile produced in the previous interval
    u32x4 pa[2][3], pb[2][3], pf23[2][2];       // ([.][2] and pf23[.][0]: one of the two is dead, depending on QK8)
    float carry[2] = {0.f, 0.f};
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int tt = 0; tt < kAhead; ++tt) {
            float p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { p[j] = __builtin_amdgcn_exp2f(rel(s0[qb][tt >> 1][8 * (tt & 1) + j], qb)); carry[qb] += p[j]; }
            u32x4 w = {pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]), pack_bf16x2(p[6], p[7])};
            pa[qb][tt] = w;
        }

    // ---- the repair path of the lazy reference (wave-uniform, rare): called at MIDCHECK of the interval of tile t when a
    // row sum of that tile left the window.  `sc` = S(t) (both key halves intact), `sn` = S(t+1) (complete, accumulated on the
    // OLD reference), `pc` / pf23 = the four P fragments of tile t, not yet consumed.
    auto repair = [&](f32x16 (&sc)[2][2], f32x16 (&sn)[2][2], u32x4 (&pc)[2][3], float& ps0, float& ps1) __attribute__((always_inline)) {
        // straight-line and cut into small steps by sched_barriers: at this point ~225 VGPRs are live, and a scheduler that
        // overlaps the steps for latency (as it would by default) spills
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // sn's accumulate chains ended in the last MFMA slots
        if constexpr (QK8) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // (16-pass MFMAs there)
        if (a.flags != nullptr && lane == 0) atomicAdd(a.flags - 2, 1);       // scratch header word [2]: repair events (statistics)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float delta = fmaxf(rel(max_with_lane_xor32(rowmax32(sc[qb])), qb), 0.f);       // >= 0: the reference only rises
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            l_run[qb] *= alpha;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) W4_SCALE_ACC(o[qb][dt][r], alpha);
            __builtin_amdgcn_sched_barrier(0);
            nm[qb] -= delta;
            const float sh = PKSUB ? nm[qb] : -delta;               // what the exponent adds to a (scaled) score of `sc` from now on
            nmp[qb] = f32x2{nm[qb], nm[qb]};
            if constexpr (SPLAT) {
                // in-place (asm "+v"): written as plain assignments these values get new registers on this path and the
                // common path pays for it with 16-register copies at the join
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("v_mov_b32 %0, %1" : "+v"(negm[qb][r]) : "v"(nm[qb]));
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(sn[qb][kt][r]) : "v"(delta));
            }
            float psum = 0.f;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                float p[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { p[j] = __builtin_amdgcn_exp2f(PKSUB ? __builtin_fmaf(sc[qb][tt >> 1][8 * (tt & 1) + j], a.scale_log2e, sh) : sc[qb][tt >> 1][8 * (tt & 1) + j] + sh); psum += p[j]; }
                const u32x4 w = {pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]), pack_bf16x2(p[6], p[7])};
                if (tt < kAhead) pc[qb][tt] = w; else pf23[qb][tt - 2] = w;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (qb == 0) ps0 = psum; else ps1 = psum;
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7" ::: "memory");                  // VALU-written P fragments / O -> MFMA operands
    };

    // MAXFREE = false: the K / V^T fragments are read by inline-asm ds_read_b128 straight into AGPRs (they are MFMA A operands
    // only), which frees the 32 VGPRs the two -m splats need; hipcc does not track asm LDS reads, so the generated schedule
    // carries the s_waitcnt lgkmcnt count of every first use (LDS reads return in order).
    auto interval = [&](f32x16 (&sc)[2][2], f32x16 (&sn)[2][2], u32x4 (&pc)[2][3], u32x4 (&pn)[2][3], auto kslot_c, auto vslot_c,
                        int t) __attribute__((always_inline)) {
        constexpr int kslot_next = decltype(kslot_c)::value, vslot = decltype(vslot_c)::value;
        u32x4 kfr[4], vfr[4];          // MAXFREE (VGPRs, hipcc's own ds_reads)
        // lazy forms: ONE 5-slot AGPR ring for both operand streams -- K fragment f in slot f & 3 (segment B), V^T fragment f
        // in slot (f + 4) % 5 (segment C; its first four are requested in B28..B31, when K slots 0..2 and the spare slot 4 are
        // free, and K fragments 0..3 of the next interval are requested behind the fence)
        u32x4 fr[5];
        const __amdgpu_buffer_rsrc_t rk = k_rsrc(t + 2), rv = v_rsrc(t + 1);
        float e[64];
        f32x2 dd[32];                                         // PKSUB: shifted score pairs
        float ps0 = carry[0], ps1 = carry[1];                // this tile's row sums so far (first key half)
        float cn0 = 0.f, cn1 = 0.f;                           // the next tile's
#define SB() __builtin_amdgcn_sched_barrier(0)
#define W4_DSREAD_A(D, ADR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(D) : "v"(ADR), "i"(OFF))
#define W4_LGKM(w) do { if constexpr ((w) >= 0) asm volatile("s_waitcnt lgkmcnt(%0)" :: "i"((w) < 0 ? 0 : (w))); } while (0)
#define RDK(f) do { if constexpr (MAXFREE) kfr[(f) & 3] = lds_read16_at(k_adr[(f) >> 1] + kslot_next * kKTileBytes + ((f) & 1) * 32 * 256); \
                    else W4_DSREAD_A(fr[(f) & 3], k_adr[(f) >> 1], kslot_next * kKTileBytes + ((f) & 1) * 32 * 256); } while (0)
#define RDV(f) do { if constexpr (MAXFREE) vfr[(f) & 3] = lds_read16_at(v_adr[(f) >> 2] + vslot * kVTileBytes + ((f) & 3) * 32 * 128); \
                    else W4_DSREAD_A(fr[((f) + 4) % 5], v_adr[(f) >> 2], vslot * kVTileBytes + ((f) & 3) * 32 * 128); } while (0)
// the MFMA is pinned at the head of its slot (a sched_barrier on both sides): left free, hipcc sinks the fillers of every
// other slot above their MFMA, which pairs the MFMAs up (gap 0) and doubles the fillers of the next gap (8-10 > the ~5 that hide)
#define QK(qb, kt, ks, f, w) do { if constexpr (MAXFREE) { if ((ks) == 0) W4_MFMA0(sn[qb][kt], kfr[(f) & 3], qf[qb][0]); else W4_MFMA_S(sn[qb][kt], kfr[(f) & 3], qf[qb][ks]); } \
                                  else { W4_LGKM(w); if ((ks) == 0) { if constexpr (SPLAT) W4A_MFMA_C(sn[qb][kt], fr[(f) & 3], qf[qb][0], negm[qb]); else W4A_MFMA0(sn[qb][kt], fr[(f) & 3], qf[qb][0]); } \
                                         else W4A_MFMA_S(sn[qb][kt], fr[(f) & 3], qf[qb][ks]); } SB(); } while (0)
// QK8: fragment f = 2 dh + kt
#define RDK8(f) asm volatile("ds_read_b128 a[%2:%3], %0 offset:%6\n\tds_read_b128 a[%4:%5], %1 offset:%6" \
                             :: "v"(k8_adr[2 * ((f) >> 1)]), "v"(k8_adr[2 * ((f) >> 1) + 1]), "i"(224 + 8 * (f)), "i"(227 + 8 * (f)), "i"(228 + 8 * (f)), \
                                "i"(231 + 8 * (f)), "i"(kslot_next * kKTileBytes + ((f) & 1) * 32 * 128) : W8_KCLOB)
#define QK8(qb, kt, dh, f, w) do { W4_LGKM(w); if ((dh) == 0) W8K_MFMA_C(sn[qb][kt], f, qf8[qb][0], negm[qb]); else W8K_MFMA_S(sn[qb][kt], f, qf8[qb][1]); SB(); } while (0)
#define G8(j) do { if ((j) < 2) stage_k_piece(rk, 1 - kslot_next, (j)); else stage_v_piece(rv, 1 - vslot, (j) - 2); } while (0)
// (the always-true SCALAR test `a.nsplit >= 0` in front of the vector test is a code-generation workaround: it ends the basic block
// ahead of the wave-wide compare, and only then does hipcc keep the loop free of accumulator <-> VGPR copies -- without it 7
// v_accvgpr_read + 1 v_accvgpr_write appear per two tiles; tests/test_isa_static.py watches the loop's instruction histogram)
#define MIDCHECK() do { if constexpr (!MAXFREE) { if (a.nsplit >= 0) { if (__builtin_expect(!__all(fmaxf(ps0, ps1) <= kW4Trigger), 0)) repair(sc, sn, pc, ps0, ps1); } } } while (0)
#define PV(qb, dt, tt, f, w) do { if constexpr (MAXFREE) { if ((tt) < 2) W4_MFMA_O(o[qb][dt], vfr[(f) & 3], pc[qb][(tt) & 1]); else W4_MFMA_O(o[qb][dt], vfr[(f) & 3], pf23[qb][(tt) & 1]); } \
                                  else { W4_LGKM(w); if ((tt) < kAhead) W4A_MFMA_O(o[qb][dt], fr[((f) + 4) % 5], pc[qb][tt]); else W4A_MFMA_O(o[qb][dt], fr[((f) + 4) % 5], pf23[qb][(tt) & 1]); } SB(); } while (0)
#define G(j) do { if ((j) < 4) stage_k_piece(rk, 1 - kslot_next, (j)); else stage_v_piece(rv, 1 - vslot, (j) - 4); } while (0)
// score i = 32 qb + 8 tt + j: tt >= kAhead reads the current tile (key half kt = 1), tt < kAhead the next tile
#define TT_(i) (((i) >> 3) & 3)
#define SRC(i) (TT_(i) >= kAhead ? sc[(i) >> 5][1][(i) & 15] : sn[(i) >> 5][TT_(i) >> 1][(i) & 15])
// PKSUB: scores leave the accumulator raw; D(p) turns the pair (2p, 2p + 1) into exponent arguments s * c - m with one v_pk_fma_f32
// (c = softmax_scale * log2(e) for plain q, exactly 1 for pre-scaled q), scheduled >= 1 op ahead of the first exp that reads it
#define D(p) do { const f32x2 s2_ = {SRC(2 * (p)), SRC(2 * (p) + 1)}; dd[p] = __builtin_elementwise_fma(s2_, cpk, nmp[(p) >> 4]); } while (0)
#define E(i) do { if constexpr (PKSUB) e[i] = __builtin_amdgcn_exp2f(dd[(i) >> 1][(i) & 1]); else e[i] = __builtin_amdgcn_exp2f(SRC(i)); } while (0)
#define A(i) do { if (TT_(i) >= kAhead) { if ((i) < 32) ps0 += e[i]; else ps1 += e[i]; } else { if ((i) < 32) cn0 += e[i]; else cn1 += e[i]; } } while (0)
#define C(w) do { const unsigned pk_ = pack_bf16x2(e[((w) >> 4) * 32 + ((w) & 15) * 2], e[((w) >> 4) * 32 + ((w) & 15) * 2 + 1]); \
                  if ((((w) >> 2) & 3) >= kAhead) pf23[(w) >> 4][((w) >> 2) & 1][(w) & 3] = pk_; \
                  else pn[(w) >> 4][((w) >> 2) & 3][(w) & 3] = pk_; } while (0)
        if constexpr (QK8) {
#include "attn_w4_sched_q8.inc"
        } else if constexpr (PKSUB) {
#include "attn_w4_sched_pk.inc"
        } else {
#ifdef WAN_ATTN_SCHED_ALT       // developer A/B builds only (a second library next to the product one); never defined by the Makefile
#include WAN_ATTN_SCHED_ALT
#else
#include "attn_w4_sched.inc"
#endif
        }
#undef RDK
#undef RDK8
#undef QK8
#undef G8
#undef RDV
#undef QK
#undef MIDCHECK
#undef W4_DSREAD_A
#undef W4_LGKM
#undef PV
#undef G
#undef E
#undef D
#undef SRC
#undef TT_
#undef A
#undef C
        l_run[0] += ps0;
        l_run[1] += ps1;
        carry[0] = cn0;
        carry[1] = cn1;
    };

    const int nfull = nkv - 1;  // tiles handled by the steady-state intervals; the last tile is peeled
    int it = 0;
    bool last_in_s1 = false;
    for (; it + 2 <= nfull; it += 2) {          // `it` is even here: K(it+1) sits in slot 1, V(it) in slot 0
        interval(s0, s1, pa, pb, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, it);
        fence();
        interval(s1, s0, pb, pa, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, it + 1);
        fence();
    }
    if (it < nfull) {
        interval(s0, s1, pa, pb, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, it);
        fence();
        ++it;
        last_in_s1 = true;
    }
    // PERSIST: what the peeled tile and the epilogue need of THIS block, before `setup_block` describes the next one.  With an even
    // tile count the whole ring but V slot 1 (the peeled tile's) is free from here on -- K(nkv-1) was consumed by the S product of the
    // last interval, behind its fence -- and the Q fragments are dead: the next block's K(0) / V(0) / K(1) requests and its Q loads go
    // out NOW and travel under the peeled tile and the epilogue (~5k cycles) instead of under the epilogue's 16 stores alone.
    const int Lk_blk = Lk, qblk_blk = qblk;
    bf16_t* const O_blk = O;
    const int w_next = w_cur + (int)gridDim.x;
    const bool more = PERSIST && w_next < a.nwg;
    bool staged = false;
    if constexpr (PERSIST) {
        if (more && (nkv & 1) == 0) {
            setup_block(w_next);
            stage_prologue();
            load_q();
            staged = true;
        }
    }
    // ---- peeled last tile (it == nkv - 1): mask keys >= Lk, no staging, no next S.  All four P fragments are recomputed
    // here with the mask (the first two that the last interval produced ahead of time, and `carry`, are dropped).
    {
        u32x4 pf[2][4];
        const int kv0 = it * kKV;
        const unsigned vb = (it & 1) * kVTileBytes;
        if (last_in_s1) {                                  // value copies (a select of array lvalues would pin both in memory)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) s0[qb][kt] = s1[qb][kt];
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float psum = 0.f;
            f32x16 sl[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                sl[kt] = s0[qb][kt];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + 32 * kt + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (key >= Lk_blk) sl[kt][r] = -INFINITY;
                }
            }
            float delta = 0.f;
            if constexpr (!MAXFREE) {
                // the last tile takes the classical step: exact row max of its valid keys, rescale if it exceeds the reference
                delta = fmaxf(rel(max_with_lane_xor32(rowmax32(sl)), qb), 0.f);
                if (!__all(delta <= 0.f)) {                   // wave-uniform
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    l_run[qb] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) W4_SCALE_ACC(o[qb][dt][r], alpha);
                }
                nm[qb] -= delta;
                delta = PKSUB ? -nm[qb] : delta;              // PKSUB scores are raw: scale, then subtract the whole (new) reference
            }
            const float cl = PKSUB ? a.scale_log2e : 1.0f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    float p[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { p[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(sl[kt][8 * t2 + j], cl, -delta)); psum += p[j]; }
                    u32x4 w = {pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]), pack_bf16x2(p[6], p[7])};
                    pf[qb][2 * kt + t2] = w;
                }
            }
            l_run[qb] += psum;
        }
        SB();
        asm volatile("s_nop 7" ::: "memory");             // VALU-written P fragments (and a rescaled O) -> MFMA operands
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const u32x4 vf0 = lds_read16_at(v_adr[tt] + vb + dt * 32 * 128);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) W4_MFMA_O(o[qb][dt], vf0, pf[qb][tt]);
            }
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // the last MFMAs' D -> the accumulator reads below
    }
#undef SB
#undef W8_MFMA0
#undef W8_MFMA_SV
#undef W8K_MFMA_C
#undef W8K_MFMA_S
#undef W4_MFMA0
#undef W4A_MFMA_C
#undef W4_SCALE_ACC
#undef W4A_MFMA_S
#undef W4A_MFMA0
#undef W4A_MFMA_O
#undef W4_MFMA_S
#undef W4_MFMA_O

    if constexpr (SPLIT) {
        // partial result of this KV range: un-normalised O, its reference (log2 units) and its sum
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
            if (qrow[qb] >= a.Lq) continue;
            const int64_t r = ((int64_t)(batch * a.nsplit + split) * a.H + head) * a.rows_tail + (qrow[qb] - a.row0);
            float* wo = a.ws_o + r * kD + 4 * hi;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(wo + 32 * dt + 8 * g) =
                        make_float4(o[qb][dt][4 * g + 0], o[qb][dt][4 * g + 1], o[qb][dt][4 * g + 2], o[qb][dt][4 * g + 3]);
            if (hi == 0) {
                a.ws_ml[2 * r] = -nm[qb];
                a.ws_ml[2 * r + 1] = l_tot;
            }
        }
    } else {
        bool ok = true;
        float inv[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
            ok = ok && ((l_tot >= 0x1p-90f && l_tot <= 0x1p100f) || qrow[qb] >= a.Lq);     // NaN fails both comparisons
            inv[qb] = 1.0f / l_tot;
        }
        if constexpr (MAXFREE) {
            const int bad = __syncthreads_or(!ok);
            if (tid == 0) a.flags[wg_linear] = bad;
        }
        if constexpr (PERSIST) {
            // this block's output window: rows [256 qblk, min(Lq, 256 qblk + 256)) of this head, as a buffer whose range check drops
            // the rows past Lq -- every lane issues all 16 stores whatever its row (the count the next block's vmcnt(16) relies on)
            const int rows_here = min(a.Lq - qblk_blk * kQPerWG, kQPerWG);
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(O_blk + (int64_t)qblk_blk * kQPerWG * a.ldo), 0, (int)((((int64_t)rows_here - 1) * a.ldo + kD) * 2), 0x00020000);
            int ovoff[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) ovoff[qb] = (int)((((int64_t)(wid * 64 + qb * 32 + l31)) * a.ldo + 8 * hi) * 2);
            if (more && !staged) {                        // odd tile count: V slot 0 was the peeled tile's -- the requests go out here,
                __syncthreads();                          // once every wave has read its last fragments, still BEFORE this block's stores
                setup_block(w_next);
                stage_prologue();
                load_q();
            }
            asm volatile("" ::: "memory");
            // 16-byte stores: a lane (row q, half hi) holds d = 32 dt + 8 g + 4 hi .. + 3; the two lanes of a row trade one 8-byte
            // piece per g pair (v_permlane32_swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second), after which lane
            // hi owns the 8 consecutive d = 32 dt + 16 m + 8 hi .. + 7 -- 16 stores of 16 bytes per lane instead of 32 of 8
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        unsigned w0 = pack_bf16x2(o[qb][dt][8 * m + 0] * inv[qb], o[qb][dt][8 * m + 1] * inv[qb]);
                        unsigned w1 = pack_bf16x2(o[qb][dt][8 * m + 2] * inv[qb], o[qb][dt][8 * m + 3] * inv[qb]);
                        unsigned w2 = pack_bf16x2(o[qb][dt][8 * m + 4] * inv[qb], o[qb][dt][8 * m + 5] * inv[qb]);
                        unsigned w3 = pack_bf16x2(o[qb][dt][8 * m + 6] * inv[qb], o[qb][dt][8 * m + 7] * inv[qb]);
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));
                        const u32x4 w = {w0, w1, w2, w3};
                        __builtin_amdgcn_raw_buffer_store_b128(w, orsrc, ovoff[qb] + (32 * dt + 16 * m) * 2, 0, 0);
                    }
            if (!more) break;
            w_cur = w_next;
            first_block = false;
            continue;
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (qrow[qb] >= a.Lq) continue;
            bf16_t* op = O + (int64_t)qrow[qb] * a.ldo + 4 * hi;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 w = {pack_bf16x2(o[qb][dt][4 * g + 0] * inv[qb], o[qb][dt][4 * g + 1] * inv[qb]),
                               pack_bf16x2(o[qb][dt][4 * g + 2] * inv[qb], o[qb][dt][4 * g + 3] * inv[qb])};
                    *reinterpret_cast<u32x2*>(op + 32 * dt + 8 * g) = w;
                }
        }
    }
    break;
  }
}

// ====================================================================================================
// fp8 QK^T AND fp8 P.V (opt-in, lossy; wan_attention_fwd_f8): the 4-wave structure with BOTH products on the fp8 matrix pipe --
// 16 x v_mfma_scale_f32_32x32x64_f8f6f4 per tile (256 matrix-pipe passes; the fp8-QK^T form: 384, bf16: 512).
//   * S = K.Q^T as in the QK8 form (e4m3 q / k with static power-of-two scales undone by the MFMA's operand scales);
//   * operand layout of the 32x32x64 fp8 MFMA (tools/probe/mx_block_probe.hip): byte b of lane (row, hi) is k = 32 (b >> 4) + 16 hi + (b & 15),
//     and the MX block kb (k in [32 kb, 32 kb + 32)) -- bytes [16 kb, 16 kb + 16) of BOTH lanes of a row -- takes its scale from lane hi = kb.
//     With the key order of the S registers (byte 16 kt + 8 g + j of lane hi = key 32 kt + 16 g + 8 hi + j) block kb is simply the key
//     half kt = kb of the tile: 32 consecutive keys;
//   * P is quantised as MX blocks: the 32 probabilities of one query row and key half (16 in each of the row's two lanes) share ONE
//     power-of-two scale taken from their fp32 sum s (one v_permlane32_swap + add per query block; scale = 2^(floor(log2 s) - 7): every
//     p / scale < 256, the largest >= 4), applied by v_cvt_scalef32_pk_fp8_f32 on the way to e4m3 and undone exactly by the MFMA's
//     E8M0 scale operand.  No bound on p is needed and none on the softmax reference: the error is relative to the block;
//   * V^T arrives as MX e4m3 from wan_vt_quantize_mx: per channel row and per 32 consecutive keys an E8M0 scale, the 64 keys of a tile
//     stored in the order the P registers hold them (position 32 hi + 16 kt + 8 g + j <- key 32 kt + 16 g + 8 hi + j);
//     tile image in LDS [128 d][64 B], chunk' = chunk ^ ((row >> 2) & 3); the tile's 256 scale bytes ([lane][dt]) ride in a 4-byte DMA;
//   * max-free softmax (reference 0: p = exp2(S), checked at the end like the bf16 max-free form; a flagged workgroup is redone by
//     the fp8-QK^T lazy-reference kernel launched right behind): with no -m splats and half-size P registers the 32 exponentials of a
//     block fit next to both S sets;
//   * the softmax stream runs in blocks (attn_f8_sched.inc): segment B finishes tile t (query block 1), segment C starts tile t+1
//     (query block 0), 10-11 micro-ops under each 64-cycle MFMA.
// K / V^T fragments live in fixed registers a[224:255] / a[192:223] (see W8_KCLOB above).
// ====================================================================================================
typedef short i16x2 __attribute__((ext_vector_type(2)));
constexpr int kF8KTile = kKV * kD;               // 8 KiB: 64 keys x 128 B
constexpr int kF8VTile = kD * kKV;               // 8 KiB: 128 d x 64 B
constexpr int kF8VOff = 2 * kF8KTile, kF8SOff = kF8VOff + 2 * kF8VTile;
constexpr int kLdsBytesF8 = kF8SOff + 2 * 256;

__global__ __launch_bounds__(kW4Threads) __attribute__((amdgpu_waves_per_eu(1, 1)))
void attn_fwd_f8_kernel(AttnArgs a) {
    const int wg_linear = blockIdx.x;
    {
        int* const hdr = a.flags - 4;
        if (wg_linear == 0 && threadIdx.x == 0) hdr[1] = 0;
        if (hdr[0] != 0) {                                       // sticky "attempt off": hand everything to the fix-up launch
            if (threadIdx.x == 0) a.flags[wg_linear] = 1;
            return;
        }
    }
    int qblk, bh;
    if (a.xcd_map) {
        const int s_ = wg_linear >> 3, g_ = s_ / a.nqb;
        qblk = s_ - g_ * a.nqb;
        bh = g_ * 8 + (wg_linear & 7);
    } else {
        bh = wg_linear / a.nqb;
        qblk = wg_linear - bh * a.nqb;
    }
    const int batch = bh / a.H, head = bh - batch * a.H;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int Lk = a.Lk, nkv = (Lk + kKV - 1) / kKV;
    const unsigned char* K8 = a.k8 + batch * a.k8_bs + head * kD;
    const unsigned char* V8 = a.v8 + batch * a.v8_bs + (int64_t)head * kD * a.ldv8;
    const unsigned char* VS = a.vs8 + ((int64_t)batch * a.H + head) * a.vs8_hs;
    bf16_t* O = a.o + batch * a.o_bs + head * kD;

    int qrow[2];
    i32x8 qf8[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qrow[qb] = qblk * kQPerWG + wid * 64 + qb * 32 + l31;
        const unsigned char* qp = a.q8 + batch * a.q8_bs + head * kD + (int64_t)min(qrow[qb], a.Lq - 1) * a.ldq8 + hi * 32;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
            const u32x4 lo = *reinterpret_cast<const u32x4*>(qp + dh * 64), up = *reinterpret_cast<const u32x4*>(qp + dh * 64 + 16);
            qf8[qb][dh] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)up[0], (int)up[1], (int)up[2], (int)up[3]};
        }
    }
    // ---- staging: per tile this wave copies K pieces 2 wid, 2 wid + 1 (8 rows x 128 B), V^T pieces 2 wid, 2 wid + 1 (16 rows x 64 B);
    // wave 0 also the tile's 256 scale bytes
    int k_voff[2], v_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kr = (wid * 2 + j) * 8 + (lane >> 3);
        k_voff[j] = (int)(kr * a.ldk8 + ((lane & 7) ^ ((kr >> 1) & 7)) * 16);
        const int vr = (wid * 2 + j) * 16 + (lane >> 2);
        v_voff[j] = (int)(vr * a.ldv8 + ((lane & 3) ^ ((lane >> 4) & 3)) * 16);      // (vr >> 2) & 3 == (lane >> 4) & 3
    }
    const int64_t k_tile_bytes = (int64_t)kKV * a.ldk8;
    auto k_rsrc = [&](int t) {
        const int tc = min(t, nkv - 1);
        const int64_t left = (int64_t)(Lk - tc * kKV) * a.ldk8;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(K8 + tc * k_tile_bytes), 0, (int)min(left, (int64_t)0x7fffffff), 0x00020000);
    };
    auto v_rsrc = [&](int t) { return __builtin_amdgcn_make_buffer_rsrc((void*)(V8 + (int64_t)min(t, nkv - 1) * kKV), 0, 0x7fffffff, 0x00020000); };
    auto s_rsrc = [&](int t) { return __builtin_amdgcn_make_buffer_rsrc((void*)(VS + (int64_t)min(t, nkv - 1) * 256), 0, 256, 0x00020000); };
    auto stage_k = [&](__amdgpu_buffer_rsrc_t r, int slot, int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + slot * kF8KTile + (wid * 2 + j) * 1024), 16, k_voff[j], 0, 0, 0);
    };
    auto stage_v = [&](__amdgpu_buffer_rsrc_t r, int slot, int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + kF8VOff + slot * kF8VTile + (wid * 2 + j) * 1024), 16, v_voff[j], 0, 0, 0);
    };
    auto stage_s = [&](__amdgpu_buffer_rsrc_t r, int slot) {
        if (wid == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + kF8SOff + slot * 256), 4, lane * 4, 0, 0, 0);
    };
    const int pi = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned k8_adr[4];         // chunk 4 dh + 2 hi + x of K row pi (index 2 dh + x)
#pragma unroll
    for (int i = 0; i < 4; ++i) k8_adr[i] = lds_base + pi * 128 + (((4 * (i >> 1) + 2 * hi + (i & 1)) ^ ((pi >> 1) & 7)) << 4);
    unsigned v8_adr[2];         // chunk 2 hi + x of V^T row l31 (+ 32 dt rows = + 2048 dt bytes)
#pragma unroll
    for (int x = 0; x < 2; ++x) v8_adr[x] = lds_base + kF8VOff + l31 * 64 + (((2 * hi + x) ^ ((l31 >> 2) & 3)) << 4);
    const unsigned vs_adr = lds_base + kF8SOff + lane * 4;
    unsigned sc_k, sc_q;
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(sc_k), "=v"(sc_q) : "s"(a.k8_scale), "s"(a.q8_scale));

    f32x16 o[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][i][r] = 0.f;
    float l_run[2] = {0.f, 0.f};
    auto fence = [&]() {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    };
    // one MX block: 32 probabilities -> e4m3 / their sum's scale; the E8M0 byte of the scale lands in bits 0..7 of `sbyte`
    // (P blocks are kept as eight separate 2 x 16-bit registers: __builtin_bit_cast of an ELEMENT of an ext-vector is folded to
    // element 0 by this hipcc -- every conversion's "old" operand became register 0 of the block)
    // sk0 / sk1: this lane's sums of its 16 probabilities of key half 0 / 1.  The two lanes of a query row exchange them (one
    // v_permlane32_swap): lane hi ends up with the total of key half kt = hi -- the block whose scale it owes the MFMA -- and a second
    // swap hands both lanes both scales for their conversions.
    auto block_scales = [&](float sk0, float sk1, float& sc0, float& sc1, unsigned& sbyte) {
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(sk0), "+v"(sk1));
        const float sc = __uint_as_float(__float_as_uint((sk0 + sk1) * 0x1p-7f) & 0x7f800000u);      // an exact power of two
        sbyte = __float_as_uint(sc) >> 23;
        sc0 = sc; sc1 = sc;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(sc0), "+v"(sc1));
    };
    auto quantise_block = [&](const float (&e)[32], float sk0, float sk1, i16x2 (&p8)[8], unsigned& sbyte) {
        float sc0, sc1;
        block_scales(sk0, sk1, sc0, sc1, sbyte);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float sc = r < 4 ? sc0 : sc1;
            i16x2 h = {0, 0};
            h = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(h, e[4 * r], e[4 * r + 1], sc, false);
            h = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(h, e[4 * r + 2], e[4 * r + 3], sc, true);
            p8[r] = h;
        }
    };
#define F8_PBLOCK(P) (i32x8{__builtin_bit_cast(int, (P)[0]), __builtin_bit_cast(int, (P)[1]), __builtin_bit_cast(int, (P)[2]), __builtin_bit_cast(int, (P)[3]), \
                            __builtin_bit_cast(int, (P)[4]), __builtin_bit_cast(int, (P)[5]), __builtin_bit_cast(int, (P)[6]), __builtin_bit_cast(int, (P)[7])})

    // ---- prologue: K(0), V(0) + scales, K(1) in flight; S(0); the first block of P(0)
#pragma unroll
    for (int j = 0; j < 2; ++j) { stage_k(k_rsrc(0), 0, j); stage_v(v_rsrc(0), 0, j); }
    stage_s(s_rsrc(0), 0);
    if (nkv > 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) stage_k(k_rsrc(1), 1, j);
    }
    fence();
    f32x16 s0[2][2], s1[2][2];
#define F8_MFMA0(D, A, B) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel_hi:[0,0,0]" : "=&v"(D) : "v"(A), "a"(B), "v"(sc_k), "v"(sc_q))
#define F8_MFMA_S(D, A, B) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(D) : "v"(A), "a"(B), "v"(sc_k), "v"(sc_q))
#define F8K_MFMA0(D, F, B) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[%4:%5], %1, 0, %2, %3 op_sel_hi:[0,0,0]" : "=&v"(D) : "a"(B), "v"(sc_k), "v"(sc_q), "i"(224 + 8 * (F)), "i"(231 + 8 * (F)) : W8_KCLOB)
#define F8K_MFMA_S(D, F, B) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[%4:%5], %1, %0, %2, %3 op_sel_hi:[0,0,0]" : "+v"(D) : "a"(B), "v"(sc_k), "v"(sc_q), "i"(224 + 8 * (F)), "i"(231 + 8 * (F)) : W8_KCLOB)
// O[qb][dt] (AGPR) += V^T fragment dt (fixed a[192 + 8 dt ..]) . P block (VGPR); scale_a = byte dt of the tile's scale dword, scale_b = byte 0
#define F8V_MFMA(ACC, DT, P, VSC, PSC) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, a[%4:%5], %1, %0, %2, %3 op_sel:[%6,0,0] op_sel_hi:[%7,0,0]" \
        : "+a"(ACC) : "v"(P), "v"(VSC), "v"(PSC), "i"(192 + 8 * (DT)), "i"(199 + 8 * (DT)), "i"((DT) & 1), "i"((DT) >> 1) : W8_VCLOB)
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const u32x4 lo = lds_read16_at(k8_adr[2 * dh] + kt * 32 * 128), up = lds_read16_at(k8_adr[2 * dh + 1] + kt * 32 * 128);
            const i32x8 kf0 = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)up[0], (int)up[1], (int)up[2], (int)up[3]};
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if (dh == 0) F8_MFMA0(s0[qb][kt], kf0, qf8[qb][0]);
                else F8_MFMA_S(s0[qb][kt], kf0, qf8[qb][1]);
            }
        }
    __syncthreads();            // everyone is done with K slot 0 before tile 2 lands in it
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");      // S(0) is read by VALU code below (16-pass MFMA results)

    i16x2 p8a0[8] = {}, p8a1[8] = {}, p8b[8] = {};     // P blocks: query block 0 ping-pongs (produced one interval ahead), query block 1
    unsigned psa0 = 0, psa1 = 0, psb = 0;     // their E8M0 scale bytes
    float suma = 0.f;                         // sum of the query-block-0 block that is waiting for its interval
    if (nkv > 1) {                            // (a single tile is the peeled tile: masked, computed there)
        float e[32], sk[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 32; ++i) { e[i] = __builtin_amdgcn_exp2f(s0[0][i >> 4][i & 15]); sk[i >> 4] += e[i]; }
        suma = sk[0] + sk[1];
        quantise_block(e, sk[0], sk[1], p8a0, psa0);
    }

    auto interval = [&](f32x16 (&sc)[2][2], f32x16 (&sn)[2][2], i16x2 (&pa_cur)[8], unsigned psa_cur, i16x2 (&pa_next)[8], unsigned& psa_next,
                        auto kslot_c, auto vslot_c, int t) __attribute__((always_inline)) {
        constexpr int kslot_next = decltype(kslot_c)::value, vslot = decltype(vslot_c)::value;
        const __amdgpu_buffer_rsrc_t rk = k_rsrc(t + 2), rv = v_rsrc(t + 1), rs = s_rsrc(t + 1);
        float e[32];
        // stream 1 = query block 1 of this tile, stream 0 = query block 0 of the next; per stream the lane's sums of key half 0 / 1,
        // their exchanged copies, the two block scales
        float s1k0 = 0.f, s1k1 = 0.f, s0k0 = 0.f, s0k1 = 0.f, x1a, x1b, x0a, x0b, sc1k0, sc1k1, sc0k0, sc0k1;
        unsigned vsc;
        l_run[0] += suma;                      // the query-block-0 block of this tile was summed one interval ago
#define SB() __builtin_amdgcn_sched_barrier(0)
#define W4_LGKM(w) do { if constexpr ((w) >= 0) asm volatile("s_waitcnt lgkmcnt(%0)" :: "i"((w) < 0 ? 0 : (w))); } while (0)
#define RDK8(f) asm volatile("ds_read_b128 a[%2:%3], %0 offset:%6\n\tds_read_b128 a[%4:%5], %1 offset:%6" \
                             :: "v"(k8_adr[2 * ((f) >> 1)]), "v"(k8_adr[2 * ((f) >> 1) + 1]), "i"(224 + 8 * (f)), "i"(227 + 8 * (f)), "i"(228 + 8 * (f)), \
                                "i"(231 + 8 * (f)), "i"(kslot_next * kF8KTile + ((f) & 1) * 32 * 128) : W8_KCLOB)
#define RDV8(dt) asm volatile("ds_read_b128 a[%2:%3], %0 offset:%6\n\tds_read_b128 a[%4:%5], %1 offset:%6" \
                              :: "v"(v8_adr[0]), "v"(v8_adr[1]), "i"(192 + 8 * (dt)), "i"(195 + 8 * (dt)), "i"(196 + 8 * (dt)), \
                                 "i"(199 + 8 * (dt)), "i"(vslot * kF8VTile + (dt) * 32 * 64) : W8_VCLOB)
#define RDVS() asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(vsc) : "v"(vs_adr), "i"(vslot * 256))
#define QK8F(qb, kt, dh, f, w) do { W4_LGKM(w); if ((dh) == 0) F8K_MFMA0(sn[qb][kt], f, qf8[qb][0]); else F8K_MFMA_S(sn[qb][kt], f, qf8[qb][1]); SB(); } while (0)
#define PV8(qb, dt, w) do { W4_LGKM(w); if ((qb) == 0) { const i32x8 p_ = F8_PBLOCK(pa_cur); F8V_MFMA(o[0][dt], dt, p_, vsc, psa_cur); } \
                            else { const i32x8 p_ = F8_PBLOCK(p8b); F8V_MFMA(o[1][dt], dt, p_, vsc, psb); } SB(); } while (0)
#define G8F(j) do { if ((j) < 2) stage_k(rk, 1 - kslot_next, (j)); else if ((j) < 4) stage_v(rv, 1 - vslot, (j) - 2); else stage_s(rs, 1 - vslot); } while (0)
// block 1 reads S(t) of query block 1, block 0 reads S(t+1) of query block 0; score i of a block = key-half kt = i >> 4, register i & 15
#ifndef F8_TIMING
#define F8_TIMING 0      // developer timing builds only (-DF8_TIMING=bits: 1 no row-sum adds, 2 no conversions, 4 no exponentials, 8 no scale chain); results are garbage
#endif
#define E8(blk, i) do { if constexpr ((F8_TIMING & 4) != 0) e[i] = (blk) ? sc[1][(i) >> 4][(i) & 15] : sn[0][(i) >> 4][(i) & 15]; \
                        else e[i] = __builtin_amdgcn_exp2f((blk) ? sc[1][(i) >> 4][(i) & 15] : sn[0][(i) >> 4][(i) & 15]); } while (0)
#define A8(blk, i) do { if constexpr ((F8_TIMING & 1) == 0 || (i) % 16 == 0) { if (blk) { if ((i) < 16) s1k0 += e[i]; else s1k1 += e[i]; } else { if ((i) < 16) s0k0 += e[i]; else s0k1 += e[i]; } } } while (0)
// the six steps of block_scales() as single instructions: exchange | total | * 2^-7 | keep the exponent | its byte for the MFMA | broadcast both
#define F8_SC(k, SK0, SK1, XA, XB, C0, C1, PS) do { \
        if ((k) == 0) { XA = SK0; XB = SK1; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(XA), "+v"(XB)); } \
        else if ((k) == 1) XA = XA + XB; \
        else if ((k) == 2) XA = XA * 0x1p-7f; \
        else if ((k) == 3) C0 = __uint_as_float(__float_as_uint(XA) & 0x7f800000u); \
        else if ((k) == 4) PS = __float_as_uint(C0) >> 23; \
        else { C1 = C0; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(C0), "+v"(C1)); } } while (0)
#define SC8(blk, k) do { if constexpr ((F8_TIMING & 8) != 0) { if ((k) == 0) { if (blk) { sc1k0 = s1k0; sc1k1 = s1k1; psb = 127; } else { sc0k0 = s0k0; sc0k1 = s0k1; psa_next = 127; } } } \
                         else if (blk) F8_SC(k, s1k0, s1k1, x1a, x1b, sc1k0, sc1k1, psb); else F8_SC(k, s0k0, s0k1, x0a, x0b, sc0k0, sc0k1, psa_next); } while (0)
#define C8(blk, w) do { if constexpr ((F8_TIMING & 2) != 0 && (w) % 8 != 0) break; if (blk) p8b[(w) >> 1] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(p8b[(w) >> 1], e[2 * (w)], e[2 * (w) + 1], (w) < 8 ? sc1k0 : sc1k1, ((w) & 1) != 0); \
                        else pa_next[(w) >> 1] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(pa_next[(w) >> 1], e[2 * (w)], e[2 * (w) + 1], (w) < 8 ? sc0k0 : sc0k1, ((w) & 1) != 0); } while (0)
#define MIDPOINT() do { l_run[1] += s1k0 + s1k1; } while (0)
#include "attn_f8_sched.inc"
#undef RDK8
#undef RDV8
#undef RDVS
#undef QK8F
#undef PV8
#undef G8F
#undef E8
#undef A8
#undef SC8
#undef F8_SC
#undef C8
#undef MIDPOINT
#undef W4_LGKM
        suma = s0k0 + s0k1;
    };

    const int nfull = nkv - 1;  // tiles handled by the steady-state intervals; the last tile is peeled
    int it = 0;
    bool last_in_s1 = false;
    for (; it + 2 <= nfull; it += 2) {          // `it` is even: K(it+1) sits in slot 1, V(it) in slot 0; P block of query block 0 in p8a0
        interval(s0, s1, p8a0, psa0, p8a1, psa1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, it);
        fence();
        interval(s1, s0, p8a1, psa1, p8a0, psa0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, it + 1);
        fence();
    }
    if (it < nfull) {
        interval(s0, s1, p8a0, psa0, p8a1, psa1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, it);
        fence();
        ++it;
        last_in_s1 = true;
    }
    // ---- peeled last tile (it == nkv - 1): keys >= Lk masked (p = 0); both P blocks are (re)computed here
    {
        const int kv0 = it * kKV;
        const unsigned vb = (it & 1) * kF8VTile;
        if (last_in_s1) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) s0[qb][kt] = s1[qb][kt];
        }
        i16x2 plh[2][8];
        i32x8 pl[2];
        unsigned psl[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float e[32], sk[2] = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int key = kv0 + 32 * (i >> 4) + 16 * ((i >> 3) & 1) + 8 * hi + (i & 7);
                e[i] = key < Lk ? __builtin_amdgcn_exp2f(s0[qb][i >> 4][i & 15]) : 0.f;
                sk[i >> 4] += e[i];
            }
            quantise_block(e, sk[0], sk[1], plh[qb], psl[qb]);
            pl[qb] = F8_PBLOCK(plh[qb]);
            l_run[qb] += sk[0] + sk[1];
        }
        const unsigned vsc = *(const __attribute__((address_space(3))) unsigned*)(uintptr_t)(vs_adr + (it & 1) * 256);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7" ::: "memory");             // VALU-written P blocks -> MFMA operands
#define F8_PEEL_PV(DT) do { \
            const u32x4 lo = lds_read16_at(v8_adr[0] + vb + (DT) * 32 * 64), up = lds_read16_at(v8_adr[1] + vb + (DT) * 32 * 64); \
            const i32x8 vf = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)up[0], (int)up[1], (int)up[2], (int)up[3]}; \
            asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[%5,0,0] op_sel_hi:[%6,0,0]" \
                         : "+a"(o[0][DT]) : "v"(vf), "v"(pl[0]), "v"(vsc), "v"(psl[0]), "i"((DT) & 1), "i"((DT) >> 1)); \
            asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[%5,0,0] op_sel_hi:[%6,0,0]" \
                         : "+a"(o[1][DT]) : "v"(vf), "v"(pl[1]), "v"(vsc), "v"(psl[1]), "i"((DT) & 1), "i"((DT) >> 1)); } while (0)
        F8_PEEL_PV(0); F8_PEEL_PV(1); F8_PEEL_PV(2); F8_PEEL_PV(3);
#undef F8_PEEL_PV
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // the last MFMAs' D -> the accumulator reads below
    }
#undef SB
#undef F8_MFMA0
#undef F8_MFMA_S
#undef F8K_MFMA0
#undef F8K_MFMA_S
#undef F8V_MFMA
#undef F8_PBLOCK

    bool ok = true;
    float inv[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        ok = ok && ((l_tot >= 0x1p-90f && l_tot <= 0x1p100f) || qrow[qb] >= a.Lq);     // NaN fails both comparisons
        inv[qb] = 1.0f / l_tot;
    }
    const int bad = __syncthreads_or(!ok);
    if (tid == 0) a.flags[wg_linear] = bad;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        if (qrow[qb] >= a.Lq) continue;
        bf16_t* op = O + (int64_t)qrow[qb] * a.ldo + 4 * hi;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w = {pack_bf16x2(o[qb][dt][4 * g + 0] * inv[qb], o[qb][dt][4 * g + 1] * inv[qb]),
                           pack_bf16x2(o[qb][dt][4 * g + 2] * inv[qb], o[qb][dt][4 * g + 3] * inv[qb])};
                *reinterpret_cast<u32x2*>(op + 32 * dt + 8 * g) = w;
            }
    }
}

// V^T bf16 [B][C][ldvt] -> MX e4m3 (see attn_fwd_f8_kernel): one wave per (channel row, 8 tiles); lane = (tile, 16-byte chunk m' of the
// tile's 64 keys): hi = m' & 1, group m = m' >> 1 = 2 kt + g; the 4 lanes of a (tile, key half kt = m' >> 2) hold one MX block of 32
// consecutive keys, whose scale byte goes where lane hi = kt of the attention kernel reads it
__global__ __launch_bounds__(64) void vt_quantize_mx_kernel(const bf16_t* __restrict__ vt, int64_t ldvt, int64_t vt_bs, int C, int H, int ntiles,
                                                            unsigned char* __restrict__ v8, int64_t ldv8, int64_t v8_bs,
                                                            unsigned char* __restrict__ vs8, int64_t vs8_hs) {
    const int lane = threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    const int tile = blockIdx.x * 8 + (lane >> 3), mp = lane & 7, hi = mp & 1, m = mp >> 1;
    if (tile >= ntiles) return;                     // whole (tile, *) lane groups leave together: the shuffles below stay inside a group of 8
    const u32x4 v = *reinterpret_cast<const u32x4*>(vt + b * vt_bs + (int64_t)c * ldvt + tile * 64 + mp * 8);
    float f[8], amax = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = bf16lo_to_f32(v[j]); f[2 * j + 1] = bf16hi_to_f32(v[j]); }
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    // scale = 2^(floor(log2 amax) - 7): amax / scale in [128, 256); an all-zero (or denormal) block takes the smallest normal scale
    const unsigned ebyte = max((int)(__float_as_uint(amax) >> 23) - 7, 1);
    const float scale = __uint_as_float(ebyte << 23);
    i16x2 h0 = {0, 0}, h1 = {0, 0};
    h0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(h0, f[0], f[1], scale, false);
    h0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(h0, f[2], f[3], scale, true);
    h1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(h1, f[4], f[5], scale, false);
    h1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(h1, f[6], f[7], scale, true);
    const u32x2 out = {__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
    *reinterpret_cast<u32x2*>(v8 + b * v8_bs + (int64_t)c * ldv8 + tile * 64 + 32 * hi + 8 * m) = out;
    if ((mp & 3) == 0) {
        const int head = c / kD, d = c - head * kD, kt = mp >> 2;
        vs8[((int64_t)b * H + head) * vs8_hs + (int64_t)tile * 256 + (32 * kt + (d & 31)) * 4 + (d >> 5)] = (unsigned char)ebyte;
    }
}

// merge the nsplit partial results of the tail rows: out = sum_s O_s 2^(m_s - M) / sum_s l_s 2^(m_s - M)
__global__ __launch_bounds__(kD) void attn_combine_kernel(AttnArgs a) {
    const int row = blockIdx.x, head = blockIdx.y, batch = blockIdx.z, d = threadIdx.x;
    float M = -INFINITY;
    for (int s = 0; s < a.nsplit; ++s) {
        const int64_t r = ((int64_t)(batch * a.nsplit + s) * a.H + head) * a.rows_tail + row;
        M = fmaxf(M, a.ws_ml[2 * r]);
    }
    float num = 0.f, den = 0.f;
    for (int s = 0; s < a.nsplit; ++s) {
        const int64_t r = ((int64_t)(batch * a.nsplit + s) * a.H + head) * a.rows_tail + row;
        const float w = __builtin_amdgcn_exp2f(a.ws_ml[2 * r] - M);
        num += a.ws_o[r * kD + d] * w;
        den += a.ws_ml[2 * r + 1] * w;
    }
    a.o[batch * a.o_bs + (int64_t)(a.row0 + row) * a.ldo + head * kD + d] = (bf16_t)(num / den);
}

// ------------------------------------------------------------------ [rows, cols] -> [cols, ldt]
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, int64_t ld,
                                                             bf16_t* __restrict__ out, int64_t ldt,
                                                             int64_t rows, int cols) {
    __shared__ unsigned short tile[64][66];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const unsigned short* src = reinterpret_cast<const unsigned short*>(in);
    unsigned short* dst = reinterpret_cast<unsigned short*>(out);
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        unsigned short v = 0;
        if (r0 + r < rows && c0 + c < cols) v = src[(r0 + r) * ld + c0 + c];
        tile[r][c] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (c0 + c < cols && r0 + r < ldt) dst[(int64_t)(c0 + c) * ldt + r0 + r] = tile[r][c];   // pad cols get 0
    }
}

}  // namespace

namespace {

// Tail balancing.  Every workgroup of a launch costs the same (all stream the whole K/V of their head) and one
// fits per CU, so W workgroups take ceil(W / CUs) rounds and the last round may be nearly empty: the 5 heads of
// an 8-way Ulysses shard at L = 67 080 give 1315 = 5 x 256 + 35 workgroups -> 6 rounds for 5.14 rounds of work
// (measured 1007 vs 1165 TFLOP/s).  When the remainder is small, the last `tq` query blocks of every
// (batch, head) leave the main launch; a second launch covers them with the SAME 8-wave kernel, each workgroup
// taking 1/nsplit of the keys (so that the tail fills the chip for 1/nsplit of a round), and a small kernel
// merges the partial (O, max, sum) triples.  Needs caller-provided workspace; without it the plain launch runs.
struct TailPlan { int tq = 0, nsplit = 1, tiles_per_split = 0, main_qb = 0, rows_tail = 0; int64_t ws_bytes = 0; };

TailPlan plan_tail(int batch, int Lq, int Lk, int num_heads) {
    TailPlan p;
    const int ncu = wan_cu_count();
    const int nqb = (Lq + kQPerWG - 1) / kQPerWG, nkv = (Lk + kKV - 1) / kKV;
    p.main_qb = nqb;
    const int64_t hb = (int64_t)num_heads * batch, items = hb * nqb;
    if (wan_tune(WAN_TUNE_ATTN_TAIL) == 0 || Lk <= 1024 || items <= ncu || items % ncu == 0) return p;
    const int64_t rem = items % ncu;
    const int cand = (int)((rem + hb - 1) / hb);        // query blocks per (batch, head) moved to the tail launch
    if (cand >= nqb) return p;
    const int64_t tail_items = hb * cand, main_items = hb * (nqb - cand);
    int nsplit = (int)std::min<int64_t>(std::min<int64_t>(ncu / tail_items, nkv / 8), 16);
    if (nsplit < 2) return p;
    const int tps = (nkv + nsplit - 1) / nsplit;
    nsplit = (nkv + tps - 1) / tps;                     // no empty split
    const double before = (double)((items + ncu - 1) / ncu);
    const double after = (double)((main_items + ncu - 1) / ncu) + 1.0 / nsplit + 0.05;
    if (nsplit < 2 || after > before - 0.2) return p;
    p.tq = cand; p.nsplit = nsplit; p.tiles_per_split = tps; p.main_qb = nqb - cand;
    p.rows_tail = Lq - p.main_qb * kQPerWG;
    p.ws_bytes = (int64_t)batch * nsplit * num_heads * p.rows_tail * (kD + 2) * (int64_t)sizeof(float);
    return p;
}

}  // namespace

namespace {
// Contract check behind the `debug_checks` switch: the V^T pad columns [Lk, roundup(Lk, 64)) of every row are read by
// the last KV tile with probability 0, and 0 * NaN is NaN in the MFMA, so they must be finite.  One pass over the pad
// columns, a device flag, a stream synchronise (developer / bring-up use only; the product path never synchronises).
__global__ void vt_pad_check_kernel(const bf16_t* vt, int64_t ldvt, int64_t vt_bs, int rows, int Lk, int lk_pad, int* flag) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x, batch = blockIdx.y;
    if (row >= rows) return;
    const unsigned short* p = reinterpret_cast<const unsigned short*>(vt + batch * vt_bs + (int64_t)row * ldvt);
    bool bad = false;
    for (int c = Lk; c < lk_pad; ++c) bad |= (p[c] & 0x7f80u) == 0x7f80u;       // exponent all ones: Inf / NaN
    if (bad) atomicOr(flag, 1);
}

wan_status_t check_vt_padding(const AttnArgs& a, int batch, int64_t lk_pad, hipStream_t st) {
    if (lk_pad == a.Lk) return WAN_OK;
    int* flag = nullptr;
    if (hipMalloc(&flag, sizeof(int)) != hipSuccess || hipMemsetAsync(flag, 0, sizeof(int), st) != hipSuccess) {
        wan_set_error("wan_attention_fwd: debug check could not allocate its flag");
        return WAN_ERR_LAUNCH;
    }
    const int rows = a.H * kD;
    hipLaunchKernelGGL(vt_pad_check_kernel, dim3((unsigned)((rows + 255) / 256), (unsigned)batch), dim3(256), 0, st,
                       a.vt, a.ldvt, a.vt_bs, rows, a.Lk, (int)lk_pad, flag);
    int host = 0;
    hipError_t e = hipMemcpyAsync(&host, flag, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(flag);
    if (e != hipSuccess) {
        wan_set_error("wan_attention_fwd: debug check failed to run: %s", hipGetErrorString(e));
        return WAN_ERR_LAUNCH;
    }
    WAN_REQUIRE(host == 0, WAN_ERR_INVALID,
                "wan_attention_fwd: V^T pad columns [%d, %lld) hold Inf/NaN (the caller must keep them finite, e.g. zero)",
                a.Lk, (long long)lk_pad);
    return WAN_OK;
}

// scratch layout: [16-byte header + one int per workgroup of the un-split grid, rounded up to 256 B][partials of the
// split tail round].  The header must be zero when the scratch is first used (it carries the sticky switch).
int64_t flag_bytes(int batch, int Lq, int num_heads) {
    const int64_t wgs = (int64_t)((Lq + kQPerWG - 1) / kQPerWG) * num_heads * batch;
    return (16 + wgs * (int64_t)sizeof(int) + 255) / 256 * 256;      // 16-byte header + one int per workgroup
}
}  // namespace

namespace {
// The dispatch decision of wan_attention_fwd as host arithmetic (shared by the launcher and wan_attention_plan).
struct AttnPlan { TailPlan tail; bool fast = false, ref2 = false, xcd = false; int variant = 0; };

AttnPlan plan_attention(int batch, int Lq, int Lk, int num_heads, bool pre, int64_t workspace_bytes, bool qk8 = false, bool pv8 = false) {
    AttnPlan p;
    const bool self = Lk > 1024;
    const int nqb_all = (Lq + kQPerWG - 1) / kQPerWG;
    // plain q always takes the packed-shift form (it applies softmax_scale exactly, in the same fma); pre-scaled q the
    // accumulator form unless the developer switch asks for the other
    p.ref2 = !pre || wan_tune(WAN_TUNE_ATTN_REF) == 2;
    const int64_t fb = flag_bytes(batch, Lq, num_heads);
    if (workspace_bytes >= fb) {
        // the attempt is worth its second launch (~5-15 us of workgroups that exit at once) only on long launches: self-attention
        // over >= 4 rounds of workgroups, or -- round 6 -- fewer rounds of LONG key streams (rounds x KV tiles >= 1024, i.e. >= ~1.5 ms of
        // launch at ~1.5 us per tile: the 2- and 3-head launches of an 8-way Ulysses rank at L = 67 080 are 2.05 / 3.08 rounds of 1 049
        // tiles and sat on the lazy form until `bench.py --emulate-sp 8` showed it); short launches (cross-attention's 8 KV tiles,
        // small grids) take the one-launch lazy form.  attn_fast = 2 forces the attempt whenever there is scratch (tests)
        const int fast_mode = wan_tune(WAN_TUNE_ATTN_FAST);
        const int64_t nwg_all = (int64_t)nqb_all * num_heads * batch, cus = wan_cu_count();
        const int64_t rounds = (nwg_all + cus - 1) / cus, kv_tiles = (Lk + kKV - 1) / kKV;
        const bool long_launch = self && (nwg_all >= 4 * cus || rounds * kv_tiles >= 1024);
        p.fast = pre && !qk8 && (fast_mode == 2 || (fast_mode == 1 && long_launch));
        p.tail = plan_tail(batch, Lq, Lk, num_heads);
        if (p.tail.tq > 0 && workspace_bytes - fb < p.tail.ws_bytes) p.tail = TailPlan();
    }
    // heads pinned to XCDs: only worth it (and only balanced) when the (batch, head) pairs split evenly over the 8 XCDs
    p.xcd = wan_tune(WAN_TUNE_ATTN_XCD_MAP) != 0 && self && (num_heads * batch) % 8 == 0;
    if (qk8) p.ref2 = false;
    p.variant = (qk8 && pv8 && workspace_bytes >= fb) ? WAN_ATTN_VARIANT_W4_F8 : qk8 ? WAN_ATTN_VARIANT_W4_LAZY_QK8
                    : (p.fast ? WAN_ATTN_VARIANT_W4_MAXFREE : WAN_ATTN_VARIANT_W4_LAZY);
    if (p.xcd) p.variant |= WAN_ATTN_VARIANT_XCD_PINNED;
    if (p.tail.tq > 0) p.variant |= WAN_ATTN_VARIANT_SPLIT_TAIL;
    return p;
}
}  // namespace

extern "C" int wan_attention_plan(int batch, int Lq, int Lk, int num_heads, int head_dim, int flags, int64_t workspace_bytes) {
    if (batch <= 0 || Lq <= 0 || Lk <= 0 || num_heads <= 0 || head_dim != kD) return 0;
    return plan_attention(batch, Lq, Lk, num_heads, (flags & WAN_ATTN_Q_PRESCALED) != 0, workspace_bytes, (flags & WAN_ATTN_QK_FP8) != 0,
                          (flags & WAN_ATTN_PV_FP8) != 0).variant;
}

extern "C" int64_t wan_attention_workspace_bytes(int batch, int Lq, int Lk, int num_heads, int head_dim) {
    if (batch <= 0 || Lq <= 0 || Lk <= 0 || num_heads <= 0 || head_dim != kD) return 0;
    return flag_bytes(batch, Lq, num_heads) + plan_tail(batch, Lq, Lk, num_heads).ws_bytes;
}

namespace {
struct Qk8Operands {            // q8 = e4m3(q * softmax_scale * log2(e) * 2^q_exp), k8 = e4m3(k * 2^k_exp)
    int q_exp, k_exp;
    // fp8 P.V as well (wan_attention_fwd_f8): the MX e4m3 V^T of wan_vt_quantize_mx; v8 = NULL: bf16 P.V
    const void* v8; int64_t ldv8, v8_bs; const void* vs8;
};
int64_t vt_mx_scale_bytes_per_head(int Lk) { return (int64_t)((Lk + kKV - 1) / kKV) * 256; }
}

// q / k are bf16 tensors, or -- with `qk8` -- e4m3 tensors whose strides count BYTES
static wan_status_t attention_fwd_impl(const void* q, int64_t ldq, int64_t q_bstride,
                                       const void* k, int64_t ldk, int64_t k_bstride,
                                       const void* vt, int64_t ldvt, int64_t vt_bstride,
                                       void* out, int64_t ldo, int64_t o_bstride,
                                       int batch, int Lq, int Lk, int num_heads, int head_dim,
                                       float softmax_scale, int flags, void* workspace, int64_t workspace_bytes,
                                       void* stream, const Qk8Operands* qk8, const int* klens = nullptr) {
    WAN_REQUIRE(q && k && vt && out, WAN_ERR_INVALID, "wan_attention_fwd: null tensor");
    WAN_REQUIRE((flags & ~WAN_ATTN_Q_PRESCALED) == 0, WAN_ERR_INVALID, "wan_attention_fwd: unknown flags 0x%x", flags);
    WAN_REQUIRE(head_dim == kD, WAN_ERR_UNSUPPORTED, "wan_attention_fwd: head_dim=%d (only 128 is built)", head_dim);
    WAN_REQUIRE(batch > 0 && Lq >= 0 && Lk > 0 && num_heads > 0, WAN_ERR_INVALID,
                "wan_attention_fwd: batch=%d Lq=%d Lk=%d heads=%d", batch, Lq, Lk, num_heads);
    const int64_t C = (int64_t)num_heads * kD;
    WAN_REQUIRE(ldq >= C && ldk >= C && ldo >= C && ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0, WAN_ERR_INVALID,
                "wan_attention_fwd: row strides (%lld,%lld,%lld) too small/misaligned for %d heads",
                (long long)ldq, (long long)ldk, (long long)ldo, num_heads);
    if (qk8) {
        WAN_REQUIRE(ldq % 16 == 0 && ldk % 16 == 0 && ((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && q_bstride % 16 == 0 &&
                        k_bstride % 16 == 0, WAN_ERR_INVALID, "wan_attention_fwd_qk8: e4m3 rows must be 16-byte aligned");
        WAN_REQUIRE(qk8->q_exp >= -100 && qk8->q_exp <= 100 && qk8->k_exp >= -100 && qk8->k_exp <= 100, WAN_ERR_INVALID,
                    "wan_attention_fwd_qk8: scale exponents (%d, %d) out of range", qk8->q_exp, qk8->k_exp);
    }
    const int64_t lk_pad = ((int64_t)Lk + kKV - 1) / kKV * kKV;
    WAN_REQUIRE(ldvt >= lk_pad && ldvt % 8 == 0, WAN_ERR_INVALID,
                "wan_attention_fwd: ldvt=%lld must be >= roundup(Lk,64)=%lld and a multiple of 8",
                (long long)ldvt, (long long)lk_pad);
    if (qk8 && qk8->v8) {
        WAN_REQUIRE(qk8->vs8 != nullptr && qk8->ldv8 >= lk_pad && qk8->ldv8 % 16 == 0 && qk8->v8_bs % 16 == 0 && ((uintptr_t)qk8->v8 & 15) == 0 &&
                        ((uintptr_t)qk8->vs8 & 3) == 0, WAN_ERR_INVALID,
                    "wan_attention_fwd_f8: v8 rows must be 16-byte aligned with ldv8=%lld >= roundup(Lk,64)=%lld, scales 4-byte aligned",
                    (long long)qk8->ldv8, (long long)lk_pad);
        WAN_REQUIRE(workspace != nullptr && workspace_bytes >= flag_bytes(batch, Lq, num_heads), WAN_ERR_INVALID,
                    "wan_attention_fwd_f8: needs the scratch of wan_attention_workspace_bytes (its softmax is the checked max-free form)");
    }
    if (Lq == 0) return WAN_OK;
    static std::atomic<uint64_t> attr_done{0};
    const wan_status_t ast = wan_once_per_device(attr_done, +[]() -> wan_status_t {
        const void* fns[] = {reinterpret_cast<const void*>(&attn_fwd_w4_kernel<0, false, 0>), reinterpret_cast<const void*>(&attn_fwd_w4_kernel<1, false, 0>),
                             reinterpret_cast<const void*>(&attn_fwd_w4_kernel<0, false, 1>), reinterpret_cast<const void*>(&attn_fwd_w4_kernel<1, false, 1>),
                             reinterpret_cast<const void*>(&attn_fwd_w4_kernel<0, false, 1, true>), reinterpret_cast<const void*>(&attn_fwd_w4_kernel<1, false, 1, true>),
                             reinterpret_cast<const void*>(&attn_fwd_w4_kernel<0, true, 1>),
                             reinterpret_cast<const void*>(&attn_fwd_w4_kernel<0, false, 2>), reinterpret_cast<const void*>(&attn_fwd_w4_kernel<1, false, 2>),
                             reinterpret_cast<const void*>(&attn_fwd_w4_kernel<0, true, 2>),
                             reinterpret_cast<const void*>(&attn_fwd_w4_kernel<0, false, 1, false, true>),
                             reinterpret_cast<const void*>(&attn_fwd_w4_kernel<0, true, 1, false, true>),
                             reinterpret_cast<const void*>(&attn_fwd_w4_kernel<0, false, 1, true, true>),
                             reinterpret_cast<const void*>(&attn_fwd_f8_kernel)};
        for (size_t i = 0; i < sizeof(fns) / sizeof(fns[0]); ++i) {
            hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesW4);
            if (e != hipSuccess) {
                wan_set_error("wan_attention_fwd: cannot reserve LDS: %s", hipGetErrorString(e));
                return WAN_ERR_LAUNCH;
            }
        }
        return WAN_OK;
    });
    if (ast != WAN_OK) return ast;
    AttnArgs a;
    a.q = (const bf16_t*)q; a.ldq = ldq; a.q_bs = q_bstride;
    a.k = (const bf16_t*)k; a.ldk = ldk; a.k_bs = k_bstride;
    a.q8 = nullptr; a.k8 = nullptr; a.ldq8 = a.ldk8 = a.q8_bs = a.k8_bs = 0; a.q8_scale = a.k8_scale = 0x7f7f7f7fu;
    if (qk8) {
        a.q8 = (const unsigned char*)q; a.ldq8 = ldq; a.q8_bs = q_bstride;
        a.k8 = (const unsigned char*)k; a.ldk8 = ldk; a.k8_bs = k_bstride;
        a.q8_scale = 0x01010101u * (unsigned)(127 - qk8->q_exp);
        a.k8_scale = 0x01010101u * (unsigned)(127 - qk8->k_exp);
    }
    a.v8 = nullptr; a.vs8 = nullptr; a.ldv8 = a.v8_bs = a.vs8_hs = 0;
    const bool pv8 = qk8 != nullptr && qk8->v8 != nullptr;
    if (pv8) {
        a.v8 = (const unsigned char*)qk8->v8; a.ldv8 = qk8->ldv8; a.v8_bs = qk8->v8_bs;
        a.vs8 = (const unsigned char*)qk8->vs8; a.vs8_hs = vt_mx_scale_bytes_per_head(Lk);
    }
    a.vt = (const bf16_t*)vt; a.ldvt = ldvt; a.vt_bs = vt_bstride;
    a.o = (bf16_t*)out; a.ldo = ldo; a.o_bs = o_bstride;
    a.Lq = Lq; a.Lk = Lk; a.H = num_heads; a.klens = klens;
    const bool pre = (flags & WAN_ATTN_Q_PRESCALED) != 0;
    WAN_REQUIRE(pre || (softmax_scale > 0.f && softmax_scale < 1e30f), WAN_ERR_INVALID,
                "wan_attention_fwd: softmax_scale=%g must be positive and finite", (double)softmax_scale);
    a.scale_log2e = pre ? 1.0f : softmax_scale * 1.4426950408889634f;
    a.qblk0 = 0; a.nsplit = 1; a.tiles_per_split = 0; a.row0 = 0; a.rows_tail = 0; a.ws_o = nullptr; a.ws_ml = nullptr; a.flags = nullptr;
    const int nqb_all = (Lq + kQPerWG - 1) / kQPerWG;
    hipStream_t st = (hipStream_t)stream;
    const bool self = Lk > 1024;
    if (wan_tune(WAN_TUNE_DEBUG_CHECKS) != 0 && klens == nullptr) {       // synchronising contract check, developer builds / bring-up only
        const wan_status_t cs = check_vt_padding(a, batch, lk_pad, st);
        if (cs != WAN_OK) return cs;
    }
    // With scratch memory: (1) pre-scaled q first runs the max-free form, followed by the FIX launch of the lazy-reference form
    // on the flagged workgroups only; (2) the last partial round of a long launch is split over the keys (plan_tail).
    // Without scratch (or attn_fast = 0): ONE launch of the lazy-reference form.
    // attn_fast / attn_tail (wan_set_tuning, or WAN_ATTN_FAST / WAN_ATTN_TAIL read once at load) are developer A/B switches.
    char* ws_tail = nullptr;
    int64_t ws_usable = 0;
    if (workspace != nullptr) {
        WAN_REQUIRE(((uintptr_t)workspace & 15) == 0, WAN_ERR_INVALID, "wan_attention_fwd: workspace must be 16-byte aligned");
        const int64_t fb = flag_bytes(batch, Lq, num_heads);
        if (workspace_bytes >= fb) {
            ws_usable = workspace_bytes;
            a.flags = (int*)workspace + 4;
            ws_tail = (char*)workspace + fb;
        }
    }
    AttnPlan plan = plan_attention(batch, Lq, Lk, num_heads, pre, ws_usable, qk8 != nullptr, qk8 != nullptr && qk8->v8 != nullptr);
    if (klens != nullptr && plan.tail.tq > 0) {        // ragged batches: the split-KV tail round divides ONE key count; every workgroup walks its own
        plan.tail = TailPlan();
        plan.variant &= ~WAN_ATTN_VARIANT_SPLIT_TAIL;
    }
    const TailPlan& tp = plan.tail;
    const bool fast = plan.fast;
    a.nqb = tp.tq > 0 ? tp.main_qb : nqb_all;
    a.nbh = num_heads * batch;
    a.xcd_map = plan.xcd ? 1 : 0;
    const int64_t nwg = (int64_t)a.nqb * a.nbh;
    WAN_REQUIRE(nwg < (int64_t)1 << 31, WAN_ERR_UNSUPPORTED, "wan_attention_fwd: grid too large");
    a.nwg = (int)nwg;
    dim3 grid((unsigned)nwg);
    const bool ref2 = plan.ref2;
    const dim3 block4(kW4Threads);
    int variant;
    if (pv8) {                       // fp8 QK^T and fp8 P.V (opt-in, lossy): checked max-free form, flagged workgroups redone by the fp8-QK^T lazy kernel
        variant = WAN_ATTN_VARIANT_W4_F8;
        hipLaunchKernelGGL(attn_fwd_f8_kernel, grid, block4, kLdsBytesF8, st, a);
        WAN_CHECK_LAUNCH("wan_attention_fwd_f8");
        hipLaunchKernelGGL((attn_fwd_w4_kernel<0, false, 1, true, true>), grid, block4, kLdsBytesW4, st, a);
    } else if (qk8) {                // fp8 QK^T (opt-in, lossy): the lazy-reference kernel with its S product on the fp8 pipe
        variant = WAN_ATTN_VARIANT_W4_LAZY_QK8;
        hipLaunchKernelGGL((attn_fwd_w4_kernel<0, false, 1, false, true>), grid, block4, kLdsBytesW4, st, a);
    } else if (!fast) {               // lazy-reference 4-wave kernel, one launch: any q form, scratch or not, no input-dependent path
        variant = WAN_ATTN_VARIANT_W4_LAZY;
        // short KV streams (cross-attention: 8 tiles per query block): ONE resident workgroup per CU walks the blocks (PERSIST, see the
        // kernel); the grid stays a multiple of 8 so that w & 7 -- the XCD a head is pinned to -- is the same for every block of a workgroup
        const bool persist = !self && wan_tune(WAN_TUNE_ATTN_PERSIST) != 0 && nwg > (wan_cu_count() & ~7) && (wan_cu_count() & ~7) >= 8;
        const dim3 pgrid(persist ? (unsigned)(wan_cu_count() & ~7) : (unsigned)nwg);
        if (ref2) {
            if (self) hipLaunchKernelGGL((attn_fwd_w4_kernel<0, false, 2>), grid, block4, kLdsBytesW4, st, a);
            else if (persist) hipLaunchKernelGGL((attn_fwd_w4_kernel<1, false, 2, false, false, true>), pgrid, block4, kLdsBytesW4, st, a);
            else hipLaunchKernelGGL((attn_fwd_w4_kernel<1, false, 2>), grid, block4, kLdsBytesW4, st, a);
        } else {
            if (self) hipLaunchKernelGGL((attn_fwd_w4_kernel<0, false, 1>), grid, block4, kLdsBytesW4, st, a);
            else if (persist) hipLaunchKernelGGL((attn_fwd_w4_kernel<1, false, 1, false, false, true>), pgrid, block4, kLdsBytesW4, st, a);
            else hipLaunchKernelGGL((attn_fwd_w4_kernel<1, false, 1>), grid, block4, kLdsBytesW4, st, a);
        }
    } else {                         // max-free attempt (2 % faster), then the lazy-reference kernel on the flagged workgroups only
        variant = WAN_ATTN_VARIANT_W4_MAXFREE;
        if (self) hipLaunchKernelGGL((attn_fwd_w4_kernel<0, false, 0>), grid, block4, kLdsBytesW4, st, a);
        else hipLaunchKernelGGL((attn_fwd_w4_kernel<1, false, 0>), grid, block4, kLdsBytesW4, st, a);
        WAN_CHECK_LAUNCH("wan_attention_fwd");
        if (self) hipLaunchKernelGGL((attn_fwd_w4_kernel<0, false, 1, true>), grid, block4, kLdsBytesW4, st, a);
        else hipLaunchKernelGGL((attn_fwd_w4_kernel<1, false, 1, true>), grid, block4, kLdsBytesW4, st, a);
    }
    if (a.xcd_map) variant |= WAN_ATTN_VARIANT_XCD_PINNED;
    if (tp.tq > 0) {
        WAN_CHECK_LAUNCH("wan_attention_fwd");
        variant |= WAN_ATTN_VARIANT_SPLIT_TAIL;
        a.qblk0 = tp.main_qb; a.nsplit = tp.nsplit; a.tiles_per_split = tp.tiles_per_split;
        a.row0 = tp.main_qb * kQPerWG; a.rows_tail = tp.rows_tail;
        a.ws_o = (float*)ws_tail;
        a.ws_ml = a.ws_o + (int64_t)batch * tp.nsplit * num_heads * tp.rows_tail * kD;
        a.nqb = tp.tq; a.nbh = num_heads * batch * tp.nsplit; a.xcd_map = 0;
        dim3 tgrid((unsigned)((int64_t)a.nqb * a.nbh));
        if (qk8) hipLaunchKernelGGL((attn_fwd_w4_kernel<0, true, 1, false, true>), tgrid, block4, kLdsBytesW4, st, a);
        else if (ref2) hipLaunchKernelGGL((attn_fwd_w4_kernel<0, true, 2>), tgrid, block4, kLdsBytesW4, st, a);
        else hipLaunchKernelGGL((attn_fwd_w4_kernel<0, true, 1>), tgrid, block4, kLdsBytesW4, st, a);
        WAN_CHECK_LAUNCH("wan_attention_fwd (tail)");
        hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)tp.rows_tail, (unsigned)num_heads, (unsigned)batch), dim3(kD), 0, st, a);
    }
    wan_note_attn_variant(variant);
    WAN_CHECK_LAUNCH("wan_attention_fwd");
    return WAN_OK;
}

extern "C" wan_status_t wan_attention_fwd(const void* q, int64_t ldq, int64_t q_bstride,
                                          const void* k, int64_t ldk, int64_t k_bstride,
                                          const void* vt, int64_t ldvt, int64_t vt_bstride,
                                          void* out, int64_t ldo, int64_t o_bstride,
                                          int batch, int Lq, int Lk, int num_heads, int head_dim,
                                          float softmax_scale, int flags, void* workspace, int64_t workspace_bytes,
                                          void* stream) {
    return attention_fwd_impl(q, ldq, q_bstride, k, ldk, k_bstride, vt, ldvt, vt_bstride, out, ldo, o_bstride, batch, Lq, Lk, num_heads,
                              head_dim, softmax_scale, flags, workspace, workspace_bytes, stream, nullptr);
}

extern "C" wan_status_t wan_attention_fwd_varlen(const void* q, int64_t ldq, int64_t q_bstride,
                                                 const void* k, int64_t ldk, int64_t k_bstride,
                                                 const void* vt, int64_t ldvt, int64_t vt_bstride,
                                                 void* out, int64_t ldo, int64_t o_bstride,
                                                 int batch, int Lq, int Lk, const int32_t* k_lens, int num_heads, int head_dim,
                                                 float softmax_scale, int flags, void* workspace, int64_t workspace_bytes,
                                                 void* stream) {
    WAN_REQUIRE(k_lens != nullptr && ((uintptr_t)k_lens & 3) == 0, WAN_ERR_INVALID, "wan_attention_fwd_varlen: k_lens must be a device array of batch int32");
    return attention_fwd_impl(q, ldq, q_bstride, k, ldk, k_bstride, vt, ldvt, vt_bstride, out, ldo, o_bstride, batch, Lq, Lk, num_heads,
                              head_dim, softmax_scale, flags, workspace, workspace_bytes, stream, nullptr, k_lens);
}

extern "C" wan_status_t wan_attention_fwd_qk8(const void* q8, int64_t ldq8, int64_t q8_bstride, int q_exp,
                                              const void* k8, int64_t ldk8, int64_t k8_bstride, int k_exp,
                                              const void* vt, int64_t ldvt, int64_t vt_bstride,
                                              void* out, int64_t ldo, int64_t o_bstride,
                                              int batch, int Lq, int Lk, int num_heads, int head_dim,
                                              void* workspace, int64_t workspace_bytes, void* stream) {
    const Qk8Operands ops = {q_exp, k_exp, nullptr, 0, 0, nullptr};
    return attention_fwd_impl(q8, ldq8, q8_bstride, k8, ldk8, k8_bstride, vt, ldvt, vt_bstride, out, ldo, o_bstride, batch, Lq, Lk,
                              num_heads, head_dim, 1.0f, WAN_ATTN_Q_PRESCALED, workspace, workspace_bytes, stream, &ops);
}

extern "C" wan_status_t wan_attention_fwd_f8(const void* q8, int64_t ldq8, int64_t q8_bstride, int q_exp,
                                             const void* k8, int64_t ldk8, int64_t k8_bstride, int k_exp,
                                             const void* v8, int64_t ldv8, int64_t v8_bstride, const void* v8_scales,
                                             const void* vt, int64_t ldvt, int64_t vt_bstride,
                                             void* out, int64_t ldo, int64_t o_bstride,
                                             int batch, int Lq, int Lk, int num_heads, int head_dim,
                                             void* workspace, int64_t workspace_bytes, void* stream) {
    WAN_REQUIRE(v8 && v8_scales, WAN_ERR_INVALID, "wan_attention_fwd_f8: null tensor");
    const Qk8Operands ops = {q_exp, k_exp, v8, ldv8, v8_bstride, v8_scales};
    return attention_fwd_impl(q8, ldq8, q8_bstride, k8, ldk8, k8_bstride, vt, ldvt, vt_bstride, out, ldo, o_bstride, batch, Lq, Lk,
                              num_heads, head_dim, 1.0f, WAN_ATTN_Q_PRESCALED, workspace, workspace_bytes, stream, &ops);
}

extern "C" int64_t wan_vt_mx_scale_bytes(int batch, int num_heads, int Lk) {
    return batch > 0 && num_heads > 0 && Lk > 0 ? (int64_t)batch * num_heads * vt_mx_scale_bytes_per_head(Lk) : 0;
}

extern "C" wan_status_t wan_vt_quantize_mx(const void* vt_bf16, int64_t ldvt, int64_t vt_bstride, int batch, int num_heads, int Lk,
                                           void* v8, int64_t ldv8, int64_t v8_bstride, void* v8_scales, void* stream) {
    WAN_REQUIRE(vt_bf16 && v8 && v8_scales, WAN_ERR_INVALID, "wan_vt_quantize_mx: null tensor");
    WAN_REQUIRE(batch > 0 && num_heads > 0 && Lk > 0, WAN_ERR_INVALID, "wan_vt_quantize_mx: batch=%d heads=%d Lk=%d", batch, num_heads, Lk);
    const int64_t lk_pad = ((int64_t)Lk + kKV - 1) / kKV * kKV;
    WAN_REQUIRE(ldvt >= lk_pad && ldvt % 8 == 0 && ldv8 >= lk_pad && ldv8 % 16 == 0 && ((uintptr_t)vt_bf16 & 15) == 0 && ((uintptr_t)v8 & 15) == 0,
                WAN_ERR_INVALID, "wan_vt_quantize_mx: ldvt=%lld / ldv8=%lld must be >= roundup(Lk,64)=%lld (multiples of 8 / 16), 16-byte aligned",
                (long long)ldvt, (long long)ldv8, (long long)lk_pad);
    const int ntiles = (int)(lk_pad / kKV), C = num_heads * kD;
    hipLaunchKernelGGL(vt_quantize_mx_kernel, dim3((unsigned)((ntiles + 7) / 8), (unsigned)C, (unsigned)batch), dim3(64), 0, (hipStream_t)stream,
                       (const bf16_t*)vt_bf16, ldvt, vt_bstride, C, num_heads, ntiles, (unsigned char*)v8, ldv8, v8_bstride,
                       (unsigned char*)v8_scales, vt_mx_scale_bytes_per_head(Lk));
    WAN_CHECK_LAUNCH("wan_vt_quantize_mx");
    return WAN_OK;
}

extern "C" wan_status_t wan_transpose_bf16(const void* in, int64_t ld, void* out_t, int64_t ldt,
                                           int64_t rows, int cols, void* stream) {
    WAN_REQUIRE(in && out_t, WAN_ERR_INVALID, "wan_transpose_bf16: null tensor");
    WAN_REQUIRE(rows >= 0 && cols > 0 && ld >= cols && ldt >= rows, WAN_ERR_INVALID,
                "wan_transpose_bf16: rows=%lld cols=%d ld=%lld ldt=%lld", (long long)rows, cols, (long long)ld, (long long)ldt);
    if (ldt == 0) return WAN_OK;
    dim3 grid((unsigned)((ldt + 63) / 64), (unsigned)((cols + 63) / 64)), block(256);
    hipLaunchKernelGGL(transpose_bf16_kernel, grid, block, 0, (hipStream_t)stream, (const bf16_t*)in, ld,
                       (bf16_t*)out_t, ldt, rows, cols);
    WAN_CHECK_LAUNCH("wan_transpose_bf16");
    return WAN_OK;
}
