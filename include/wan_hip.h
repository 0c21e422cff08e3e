/*
 * libwan_hip.so -- C ABI of the MI355X (gfx950) kernels behind the Wan2.1-DiT
 * video denoising path of VideoCoF.
 *
 * The reference is pure Python; it has no FFI.  Its seams are the duck-typed
 * calls listed in SURVEY.md section 8b -- WanTransformer3DModel.forward,
 * attention(), WanRMSNorm.forward, rope_apply_qk -- which the closed `paifuser`
 * package monkey-patches with native code (videox_fun/models/__init__.py:43-109).
 * Each entry point below replaces the arithmetic of one such seam and cites it
 * (file:line relative to the reference root).  INTEGRATION.md shows the ctypes
 * binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers;
 *   - the caller owns every buffer (PyTorch caching allocator in practice); the
 *     library keeps no state between calls except the last-error string;
 *   - every call enqueues on `stream` (a hipStream_t passed as void*) and
 *     returns without synchronising;
 *   - bf16 tensors are `void*` to 2-byte bfloat16, fp32 tensors are `float*`;
 *   - `ld*` are row strides in ELEMENTS; rows are contiguous in the last dim;
 *   - return value 0 = WAN_OK, otherwise see wan_last_error().
 */
#ifndef WAN_HIP_H
#define WAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WAN_ABI_VERSION 11     /* 11: wan_gemm_fp8_ws / wan_gemm_fp8_ws_plan / wan_gemm_fp8_pk_segment (the e4m3 Linear on the persistent stream-K kernel); 10: wan_box_probe (the benchmark's box fingerprint); 9: wan_attention_fwd_varlen (ragged batches in one launch); 8: wan_qk_quantize_fp8 takes either operand alone, wan_gemm_ws_splits + the split-K form of the 128^2 GEMM, tuning key gemm_pk_form replaces gemm_pk_sched (and attn_w4 is gone with the 8-wave attention kernel), the persistent GEMM from K >= 1024, the library communicator records collectives on a capturing stream; 7: the *_split wire entry points (Ulysses head groups); 6: wan_gemm_bf16_ws / wan_gemm_workspace_bytes / wan_gemm_ws_plan (persistent stream-K GEMM with a caller workspace); 5: the fp8 attention family */

typedef enum {
    WAN_OK = 0,
    WAN_ERR_INVALID = 1,     /* bad shape / alignment / null pointer (Python side raises ValueError) */
    WAN_ERR_UNSUPPORTED = 2, /* shape the kernels do not cover (RuntimeError) */
    WAN_ERR_LAUNCH = 3       /* HIP launch failure (RuntimeError) */
} wan_status_t;

int wan_abi_version(void);
const char* wan_last_error(void);

/* Developer switches (A/B harnesses, bring-up).  The matching environment variables (WAN_ATTN_TAIL, WAN_ATTN_FAST,
 * WAN_ATTN_XCD_MAP, WAN_ATTN_REF, WAN_GEMM_W4, WAN_GEMM_GM, WAN_GEMM_PHASES, WAN_GEMM_VARIANT, WAN_CONV_XCD, WAN_DEBUG_CHECKS, WAN_ATTN_PERSIST) are
 * read ONCE, at the first call into the library; the launch paths never call getenv().  Keys: "attn_tail",
 * "attn_fast", "attn_xcd_map", "attn_ref", "conv_head", "gemm_exp", "gemm_w4", "gemm_gm", "gemm_phases", "gemm_variant", "conv_xcd", "debug_checks",
 * "gemm_pk", "gemm_pk_form", "gemm_pk_workers", "gemm_pk_min_units", "gemm_pk_order", "gemm_splitk", "row_group", "sp_inline", "conv_patch", "attn_persist" (1: cross-attention on the persistent form of the 4-wave kernel), "conv_mfma" (0 = by the per-frame plane, 32 / 16 = force v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16).
 * "debug_checks" = 1 turns on SYNCHRONISING contract checks (V^T pad columns of wan_attention_fwd are finite).
 * wan_set_tuning is an atomic store: safe against concurrent launches, which see the old or the new value.
 * wan_get_tuning returns -1 for an unknown key.  No reference counterpart (the reference has no native code).
 * Read-only key "last_attn_variant": what the most recent wan_attention_fwd call of this process launched, as
 * WAN_ATTN_VARIANT_* (kernel family | WAN_ATTN_VARIANT_XCD_PINNED | WAN_ATTN_VARIANT_SPLIT_TAIL) -- so that a benchmark
 * reports the kernel the dispatcher picked instead of a literal.  Read-only key "dev_experiments": 1 if the library was built with
 * `make EXPERIMENTS=1` (timing-only kernel variants behind "gemm_exp" compiled in; never in the product build). */
wan_status_t wan_set_tuning(const char* key, int value);
int wan_get_tuning(const char* key);
#define WAN_ATTN_VARIANT_W4_LAZY 1          /* attn_fwd_w4_kernel<.,.,1|2>: 4 waves, lazy softmax reference, one launch */
#define WAN_ATTN_VARIANT_W4_MAXFREE 2       /* attn_fwd_w4_kernel<.,false,0> + its checked fix-up launch attn_fwd_w4_kernel<.,false,1,true> */
#define WAN_ATTN_VARIANT_W8_RUNNING_MAX 3   /* (rounds 1-4: the 8-wave running-max kernel; retired in round 5, never reported any more) */
#define WAN_ATTN_VARIANT_W4_LAZY_QK8 4      /* attn_fwd_w4_kernel<0,.,1,false,true>: the lazy form with QK^T on the fp8 matrix pipe (wan_attention_fwd_qk8) */
#define WAN_ATTN_VARIANT_W4_F8 5            /* attn_fwd_f8_kernel (fp8 QK^T and fp8 P.V, checked max-free softmax) + the fp8-QK^T lazy kernel on flagged workgroups (wan_attention_fwd_f8) */
#define WAN_ATTN_VARIANT_FAMILY_MASK 15
#define WAN_ATTN_VARIANT_XCD_PINNED 16      /* every (batch, head) pinned to one XCD */
#define WAN_ATTN_VARIANT_SPLIT_TAIL 32      /* the last partial round ran split over the keys (+ merge kernel) */

/* ---------------------------------------------------------------------------
 * a4  LayerNorm (no affine) + adaLN modulate, or LayerNorm with affine.
 *     out[r,c] = bf16( LN(x[r,:])[c] * (add_one + scale[b,c]) + shift[b,c] ),  b = r / rows_per_batch
 *     replaces: WanLayerNorm.forward + `norm(x) * (1 + e[1]) + e[0]` then `.to(dtype)`
 *               (wan_transformer3d.py:233-243, 495-496, 507-508, 547) with add_one=1,
 *               and norm3 = LayerNorm(affine) (wan_transformer3d.py:448-450, 504) with
 *               add_one=0, scale=weight, shift=bias, rows_per_batch=rows.
 *     x fp32 [rows, dim] contiguous; scale/shift fp32 [nbatch, dim]; dim % 4 == 0, dim <= 8192.
 * ------------------------------------------------------------------------- */
wan_status_t wan_ln_modulate(const float* x, const float* scale, const float* shift, int add_one,
                             void* out_bf16, int64_t rows, int dim, int64_t rows_per_batch,
                             float eps, void* stream);

/* ---------------------------------------------------------------------------
 * a5+a7  WanRMSNorm over the FULL channel dim, then 3-axis RoPE with the
 *     VideoCoF temporal position map, in place on bf16 rows.
 *     replaces: WanRMSNorm.forward (wan_transformer3d.py:214-230) followed by
 *               rope_apply_qk (wan_transformer3d.py:135-211); the paifuser patch points
 *               are models/__init__.py:78-80 (rms_norm_forward) and :85-109 (fast_rope_apply_qk).
 *     One launch handles two tensors (q and k); pass x1 = NULL for a single tensor.
 *     rope_cos/rope_sin: fp32 [max_pos, head_dim/2] tables (cos/sin of the angles of
 *     rope_params, wan_transformer3d.py:44-52, 692-699); NULL = RMSNorm only (cross-attention).
 *     Row r of the call is token  t = token_offset + (r % rows_per_batch)  of its sample;
 *     tokens >= F*Hp*Wp are normalised but not rotated (wan_transformer3d.py:202).
 *     mode 0: pos_t = f;  mode 1 (paired): f < f_src ? f : f - f_src;
 *     mode 2 (CoF): f < f_src ? f+1 : (f < ground_end ? 0 : f - ground_end + 1).
 *     x0_scale multiplies the x0 result in fp32 before its single bf16 rounding (x1 is not scaled).
 *     Pass WAN_ATTN_QSCALE(softmax_scale) to hand q to wan_attention_fwd(WAN_ATTN_Q_PRESCALED)
 *     -- the `q * softmax_scale` of flash attention folded into the norm -- or 1.0f for none.
 * ------------------------------------------------------------------------- */
typedef struct {
    int F, Hp, Wp;          /* patch grid (frames, rows, cols) */
    int mode;               /* 0 default, 1 paired, 2 CoF */
    int f_src;              /* frame_split_indices[b] */
    int ground_end;         /* ground_frame_indices[b][1] (mode 2) */
    int64_t token_offset;   /* first token of this shard (sequence parallel), else 0 */
    int64_t rows_per_batch; /* rows per sample in this call */
    int max_pos;            /* rows of the cos/sin tables (1024) */
} wan_rope_params;

wan_status_t wan_rmsnorm_rope(void* x0_bf16, const float* w0, void* x1_bf16, const float* w1,
                              int64_t ld, int64_t rows, int dim, int head_dim, float eps,
                              const float* rope_cos, const float* rope_sin,
                              const wan_rope_params* rp, float x0_scale, void* stream);

/* ---------------------------------------------------------------------------
 * a8/a10/a12  nn.Linear on the MFMA cores:  acc[m,n] = sum_k A[m,k] * W[n,k]   (fp32 accumulate)
 *     replaces: nn.Linear -> cuBLAS for q/k/v/o, ffn.0/ffn.2, text_embedding, patch_embedding
 *               (as im2col GEMM) and head.head (wan_transformer3d.py:264-267, 457-459, 662-666, 530).
 *     A bf16 [M,K] (row stride lda), W bf16 [N,K] (nn.Linear layout, row stride ldw),
 *     bias fp32 [N] or NULL.  K % 64 == 0, lda/ldw % 8 == 0.  M, N arbitrary (N % 4 == 0).
 *     Epilogues:
 *       WAN_EPI_BF16        out bf16 [M,N](ldo)  = acc + bias
 *       WAN_EPI_GELU_BF16   out bf16             = gelu_tanh(acc + bias)     (ffn.0 + nn.GELU('tanh'))
 *       WAN_EPI_F32         out fp32             = acc + bias
 *       WAN_EPI_RESID_F32   out fp32 (in place)  += (acc + bias) * gate[b,n]  (gate NULL = 1):
 *                           `x = x + y * e[2]`, `x = x + cross_attn(...)`, `x = x + y * e[5]`
 *                           (wan_transformer3d.py:499, 504, 511); gate fp32 [nbatch, N], b = m / rows_per_batch
 *       WAN_EPI_BF16_T      out bf16 [N, ldo] TRANSPOSED: out[n, m] = acc + bias  (V^T for the
 *                           attention kernel; ldo >= M, ldo % 8 == 0)
 * ------------------------------------------------------------------------- */
typedef enum {
    WAN_EPI_BF16 = 0,
    WAN_EPI_GELU_BF16 = 1,
    WAN_EPI_F32 = 2,
    WAN_EPI_RESID_F32 = 3,
    WAN_EPI_BF16_T = 4
} wan_epilogue_t;

wan_status_t wan_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                           void* out, int64_t ldo, int M, int N, int K, int epilogue,
                           const float* gate, int64_t rows_per_batch, void* stream);
/* Host arithmetic only: the kernel family wan_gemm_bf16 dispatches an [M, K] x [N, K]^T product to. */
int wan_gemm_plan(int M, int N, int K);
#define WAN_GEMM_VARIANT_128 0      /* gemm_bf16_kernel: 128 x 128 x 64 tile, 4 waves, two workgroups per CU (small / under-filled shapes) */
#define WAN_GEMM_VARIANT_256_W8 1   /* gemm256_kernel: 256 x 256 x 64 tile, 8 waves, phased K loop */
#define WAN_GEMM_VARIANT_256_W4 2   /* gemm_w4_kernel: the same tile, 4 waves of 128 x 128 outputs (K >= 4096 by default) */
#define WAN_GEMM_VARIANT_256_PK 3   /* gemm_pk_kernel: the 4-wave tile as ONE persistent workgroup per CU, stream-K remainder (wan_gemm_bf16_ws) */

/* The same product with a caller-provided workspace: every nn.Linear of the DiT blocks on the path
 * (wan_transformer3d.py:264-267 q/k/v/o, :457-459 ffn) runs through this entry in the Python host and in wan_dit_block_forward.
 *     With a workspace, shapes a 256^2 kernel would take (K % 128 == 0, K >= 1024) run on the PERSISTENT form: one resident workgroup per CU walks
 *     whole output tiles as one continuous K-tile stream (no per-tile pipeline fill) and the remainder tiles are cut stream-K
 *     fashion so that every CU does the same amount of work (no tail round); split tiles are combined in K order by the last
 *     arriver (bitwise reproducible, no spin waits).  Small shapes of the 128^2 kernel whose tiles do not fill the chip (M ~ 2 000
 *     tokens: BASELINE configs[0]) run that kernel SPLIT-K (ABI 8): the K range of every tile in wan_gemm_ws_splits(M, N, K) pieces,
 *     combined in split order by the last arriver -- the same guarantees.  Everything else -- and workspace == NULL -- is wan_gemm_bf16.
 *     workspace: >= wan_gemm_workspace_bytes(M, N, K) bytes (0 when the shape would not use one), 16-byte aligned, owned by the
 *     caller, not shared by launches that may run concurrently (one per stream); its contents need not survive between calls.
 *     wan_gemm_ws_plan: the kernel family wan_gemm_bf16_ws picks when given a workspace (host arithmetic only). */
wan_status_t wan_gemm_bf16_ws(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                              void* out, int64_t ldo, int M, int N, int K, int epilogue,
                              const float* gate, int64_t rows_per_batch, void* workspace, int64_t workspace_bytes, void* stream);
int64_t wan_gemm_workspace_bytes(int M, int N, int K);
int wan_gemm_ws_plan(int M, int N, int K);
int wan_gemm_ws_splits(int M, int N, int K);      /* pieces per tile of the split-K form (1: not split); host arithmetic only */
/* Host arithmetic only (tests, curious hosts): the persistent kernel's plan.  wan_gemm_pk_grid: its grid (workers = one per CU,
 * a multiple of 8).  wan_gemm_pk_segment: segment `index` of worker `worker` -- evaluated by the SAME functions the kernel runs;
 * out[11] = tile m, tile n, first K tile, end K tile, partial?, workspace slot, arrival counter, first / last lane holding a piece of
 * the tile, my lane, stream-K tile index; returns 1 while the segment exists. */
int wan_gemm_pk_grid(int M, int N);
int wan_gemm_pk_segment(int M, int N, int K, int worker, int index, int* out);

/* ---------------------------------------------------------------------------
 * 8f-4  FP8 (OCP e4m3) projections -- an explicit, lossy option of the host model (`enable_fp8_linear`); never the default.
 *     replaces: the reference keeps e4m3 only as a STORAGE format and up-casts every weight to bf16 for the matmul
 *               (convert_weight_dtype_wrapper, videox_fun/utils/fp8_optimization.py:19-57).  Here both operands stay
 *               e4m3 into the matrix pipe: out = (A_q . W_q^T) * a_row_scale[m] * w_row_scale[n]  (+ bias, epilogue),
 *               A_q = e4m3(A / a_row_scale) per token row, W_q = e4m3(W / w_row_scale) per output channel.
 *     wan_gemm_fp8: same tile kernel, epilogues and argument meaning as wan_gemm_bf16; A [M,K] and W [N,K] are e4m3 bytes
 *               (lda / ldw in bytes = elements, multiples of 16), K % 128 == 0; products accumulate in fp32 on
 *               v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (twice the bf16 MFMA rate).
 *     wan_quantize_rows_fp8: bf16 [rows, cols] -> e4m3 + one scale per row (max|x| / 448; round-to-nearest-even).
 *     wan_ln_modulate_fp8:  wan_ln_modulate with the quantisation fused into the row kernel (e4m3 out + row scales).
 * ------------------------------------------------------------------------- */
wan_status_t wan_gemm_fp8(const void* A_fp8, int64_t lda, const float* a_row_scale, const void* W_fp8, int64_t ldw,
                          const float* w_row_scale, const float* bias, void* out, int64_t ldo, int M, int N, int K,
                          int epilogue, const float* gate, int64_t rows_per_batch, void* stream);
/* wan_gemm_fp8_ws: wan_gemm_fp8 with a caller workspace (wan_gemm_fp8_workspace_bytes(M, N, K) bytes, 16-byte aligned, not shared
 * with another stream; w_row_scale 16-byte aligned).  Shapes whose bf16 product of the same TILE count would run on the persistent
 * stream-K kernel (wan_gemm_fp8_ws_plan == WAN_GEMM_VARIANT_256_PK: K % 256 == 0 and wan_gemm_ws_plan(M, N, K / 2) says so, or K >= 4096
 * and wan_gemm_ws_plan(M, N, K) does: the 8-way Ulysses shard) run its
 * e4m3 instantiation -- same segments, stream-K combine (bitwise reproducible) and epilogues, 128-element K tiles consumed by one
 * v_mfma_scale_f32_16x16x128_f8f6f4 per output tile in two phases ("schedule P", gemm_bf16_pk.hip); everything else, and a NULL
 * workspace, is wan_gemm_fp8.  wan_gemm_fp8_pk_segment: wan_gemm_pk_segment for that plan (K tiles of 128 elements). */
int wan_gemm_fp8_ws_plan(int M, int N, int K);
int64_t wan_gemm_fp8_workspace_bytes(int M, int N, int K);
wan_status_t wan_gemm_fp8_ws(const void* A_fp8, int64_t lda, const float* a_row_scale, const void* W_fp8, int64_t ldw,
                             const float* w_row_scale, const float* bias, void* out, int64_t ldo, int M, int N, int K,
                             int epilogue, const float* gate, int64_t rows_per_batch, void* workspace, int64_t workspace_bytes,
                             void* stream);
int wan_gemm_fp8_pk_segment(int M, int N, int K, int worker, int index, int* out);
wan_status_t wan_quantize_rows_fp8(const void* x_bf16, int64_t ldx, void* out_fp8, int64_t ldo, float* out_row_scale,
                                   int64_t rows, int cols, void* stream);
wan_status_t wan_ln_modulate_fp8(const float* x, const float* scale, const float* shift, int add_one, void* out_fp8,
                                 float* out_row_scale, int64_t rows, int dim, int64_t rows_per_batch, float eps, void* stream);

/* ---------------------------------------------------------------------------
 * a9  attention(): softmax(q k^T * scale) v, non-causal, head_dim 128, bf16 in/out,
 *     fp32 softmax and accumulation (flash-style, never materialises Lq x Lk).
 *     replaces: attention()/flash_attention() (attention_utils.py:43-211) as called from
 *               WanSelfAttention.forward (wan_transformer3d.py:294-299) and
 *               WanT2VCrossAttention.forward (wan_transformer3d.py:325-330).
 *     q   bf16 [B][Lq][H*128] (row stride ldq, sample stride q_bstride)
 *     k   bf16 [B][Lk][H*128]
 *     vt  bf16 [B][H*128][ldvt]  -- V TRANSPOSED: vt[h*128+d][key]; ldvt >= roundup(Lk,64),
 *         ldvt % 8 == 0, and columns [Lk, roundup(Lk,64)) must hold finite values (zeros)
 *     out bf16 [B][Lq][H*128]
 *     Keys >= Lk are masked (k_lens semantics of the flash-attn branch, attention_utils.py:95-100).
 *     flags: WAN_ATTN_Q_PRESCALED -- q already carries softmax_scale*log2(e) (written that way by
 *     wan_rmsnorm_rope's x0_scale); softmax_scale is then ignored and the softmax reference rides inside the MFMA
 *     accumulator (no per-score multiply-add).  0 = plain q: softmax_scale (> 0) is applied to the fp32 scores exactly.
 *     Numerics do not depend on the size of the scores: every form keeps a per-row softmax reference that is raised
 *     (O and the running sum rescaled) whenever a tile's row sums leave the fp32 / bf16 window.
 * ------------------------------------------------------------------------- */
wan_status_t wan_attention_fwd(const void* q, int64_t ldq, int64_t q_bstride,
                               const void* k, int64_t ldk, int64_t k_bstride,
                               const void* vt, int64_t ldvt, int64_t vt_bstride,
                               void* out, int64_t ldo, int64_t o_bstride,
                               int batch, int Lq, int Lk, int num_heads, int head_dim,
                               float softmax_scale, int flags, void* workspace, int64_t workspace_bytes,
                               void* stream);
/* The same product over a RAGGED batch in one launch: batch b attends keys [0, k_lens[b]) of its k / vt.
 *     replaces: the cu_seqlens packing of flash_attention() (attention_utils.py:95-146: k of every sample cut to k_lens[b],
 *               concatenated, one flash_attn_varlen_func call) -- here the samples stay where they are ([B][Lk][..], Lk = the
 *               padded length) and every workgroup reads ITS sample's key count: no packing copy, no host read of k_lens.
 *     k_lens  int32 [B] in DEVICE memory, read by the kernel (values are clamped to [1, Lk]; a caller that wants the
 *             reference's all-zero rows for an EMPTY sample zeroes them afterwards, as videocof_amd/attention_utils.py does)
 *     vt      columns [k_lens[b], roundup(k_lens[b], 64)) of sample b must be finite (zeros) -- the same contract per sample
 *     Everything else as wan_attention_fwd; the split-KV tail round is not used (it divides ONE key count), the max-free
 *     attempt and the XCD pinning are. */
wan_status_t wan_attention_fwd_varlen(const void* q, int64_t ldq, int64_t q_bstride,
                                      const void* k, int64_t ldk, int64_t k_bstride,
                                      const void* vt, int64_t ldvt, int64_t vt_bstride,
                                      void* out, int64_t ldo, int64_t o_bstride,
                                      int batch, int Lq, int Lk, const int32_t* k_lens, int num_heads, int head_dim,
                                      float softmax_scale, int flags, void* workspace, int64_t workspace_bytes,
                                      void* stream);
/* Optional scratch for wan_attention_fwd (16-byte aligned device memory whose first 16 bytes are ZERO when it is first used,
 * otherwise of irrelevant content; reusable across calls on one stream).  Header words (int32): [0] sticky "max-free attempt
 * off", [1] workgroups redone by the last call, [2] repair events of the lazy softmax reference since the caller last cleared
 * it (statistics).  With at least this many bytes (a) pre-scaled q first runs a max-free kernel (p = exp2(S) with reference 0;
 * rows whose sum leaves a checked window flag their workgroup in the scratch and are recomputed by the lazy-reference kernel
 * launched right behind -- same results for unflagged workgroups, ~2 % faster; once more than 1/8 of a launch had to be
 * recomputed word [0] turns the attempt off for later calls, which then cost one lazy-reference launch: 1.36 instead of
 * 1.40 PFLOP/s, no cliff), and (b) the last, partially filled round of workgroups of a long self-attention launch is split
 * over the key range so that it fills the chip (matters when few heads are local, e.g. the 5 heads per GPU of an 8-way
 * Ulysses shard: +15 %).  workspace = NULL is always valid (one lazy-reference launch). */
int64_t wan_attention_workspace_bytes(int batch, int Lq, int Lk, int num_heads, int head_dim);
/* Host arithmetic only (no GPU needed): what wan_attention_fwd will launch for this shape, flags and scratch size, as
 * WAN_ATTN_VARIANT_* bits (the value wan_get_tuning("last_attn_variant") reports after the call); 0 for an unsupported shape.
 * Lets a benchmark / test state the dispatched kernels without launching them.  No reference counterpart. */
int wan_attention_plan(int batch, int Lq, int Lk, int num_heads, int head_dim, int flags, int64_t workspace_bytes);
#define WAN_ATTN_Q_PRESCALED 1
#define WAN_ATTN_QK_FP8 2                   /* wan_attention_plan only: plan for wan_attention_fwd_qk8 */
#define WAN_ATTN_PV_FP8 4                   /* wan_attention_plan only, with WAN_ATTN_QK_FP8: plan for wan_attention_fwd_f8 */
#define WAN_ATTN_QSCALE(softmax_scale) ((softmax_scale) * 1.4426950408889634f)

/* a9' Self-attention with the QK^T product on the fp8 matrix pipe (LOSSY, opt-in; the role of the reference's
 *     `sageattn` branch, attention_utils.py:152-211 with attention_type = "SAGE_ATTENTION": 8-bit QK^T, 16-bit P.V).
 *       q8  e4m3 [B][Lq][H*128] = e4m3( q * softmax_scale * log2(e) * 2^q_exp )      (strides in BYTES, rows 16-byte aligned)
 *       k8  e4m3 [B][Lk][H*128] = e4m3( k * 2^k_exp )
 *     both written by wan_rmsnorm_rope_fp8 (below); the power-of-two factors move typical post-RMSNorm magnitudes into e4m3's
 *     normal range (2^-6 .. 448) and are undone exactly by the MFMA's E8M0 operand scales, so the fp32 scores differ from the
 *     bf16 kernel's only by the 3-bit mantissas of q8 / k8 (a relative error of <= 2^-4 per element; |q| * 2^q_exp or
 *     |k| * 2^k_exp above 448 saturates).  Everything behind the scores -- lazy softmax reference, bf16 P, P.V, fp32 sums,
 *     split tail, XCD pinning, the scratch layout -- is wan_attention_fwd's lazy-reference form; vt / out as there.
 *     Measured error and speed: tests/test_gpu_fp8.py, DESIGN.md section 13. */
wan_status_t wan_attention_fwd_qk8(const void* q8, int64_t ldq8, int64_t q8_bstride, int q_exp,
                                   const void* k8, int64_t ldk8, int64_t k8_bstride, int k_exp,
                                   const void* vt, int64_t ldvt, int64_t vt_bstride,
                                   void* out, int64_t ldo, int64_t o_bstride,
                                   int batch, int Lq, int Lk, int num_heads, int head_dim,
                                   void* workspace, int64_t workspace_bytes, void* stream);
/* a9'' Both attention products on the fp8 matrix pipe (LOSSY, opt-in; SageAttention-2's operating point: 8-bit QK^T, fp8 P.V).
 *     q8 / k8 / vt / out / workspace as wan_attention_fwd_qk8 (the scratch is REQUIRED here: the softmax is the checked max-free form,
 *     a flagged workgroup is redone by the fp8-QK^T lazy-reference kernel with the bf16 vt).  v8 / v8_scales: V^T as MX e4m3, written by
 *     wan_vt_quantize_mx from the bf16 vt -- per channel row and per block of 32 consecutive keys one E8M0 scale (block max / scale in [128, 256)),
 *     the 64 keys of a tile stored in the order the kernel's P registers hold them; v8 [B][H*128][ldv8 bytes], ldv8 >= roundup(Lk, 64),
 *     % 16; v8_scales: wan_vt_mx_scale_bytes(batch, heads, Lk) bytes.  P is quantised inside the kernel, also as MX blocks (32 keys
 *     per query row, scale from the block's fp32 sum), so neither operand needs a bounded range.  Error / speed: DESIGN.md section 13. */
wan_status_t wan_attention_fwd_f8(const void* q8, int64_t ldq8, int64_t q8_bstride, int q_exp,
                                  const void* k8, int64_t ldk8, int64_t k8_bstride, int k_exp,
                                  const void* v8, int64_t ldv8, int64_t v8_bstride, const void* v8_scales,
                                  const void* vt, int64_t ldvt, int64_t vt_bstride,
                                  void* out, int64_t ldo, int64_t o_bstride,
                                  int batch, int Lq, int Lk, int num_heads, int head_dim,
                                  void* workspace, int64_t workspace_bytes, void* stream);
int64_t wan_vt_mx_scale_bytes(int batch, int num_heads, int Lk);
wan_status_t wan_vt_quantize_mx(const void* vt_bf16, int64_t ldvt, int64_t vt_bstride, int batch, int num_heads, int Lk,
                                void* v8, int64_t ldv8, int64_t v8_bstride, void* v8_scales, void* stream);
/* wan_rmsnorm_rope that leaves x0 / x1 untouched and writes e4m3 copies of the results, dense [rows][dim] bytes:
 * out0 = e4m3(bf16(result0 * x0_scale)), out1 = e4m3(bf16(result1 * x1_scale)) -- the bf16 rounding first, so that with
 * power-of-two extra factors the e4m3 value is a quantisation of exactly the number the bf16 path would have used. */
wan_status_t wan_rmsnorm_rope_fp8(const void* x0_bf16, const float* w0, const void* x1_bf16, const float* w1,
                                  int64_t ld, int64_t rows, int dim, int head_dim, float eps,
                                  const float* rope_cos, const float* rope_sin, const wan_rope_params* rp,
                                  float x0_scale, float x1_scale, void* out0_fp8, void* out1_fp8, void* stream);

/* K smoothing for a9' (what `sageattn(..., smooth_k=True)`, the default of the wheel behind attention_utils.py:173-185, does):
 * softmax_j(q_i . k_j) is unchanged when one vector is subtracted from every k_j, so the e4m3 copy of k is taken of
 * k - mean_over_tokens(k) -- a channel with a large common offset would otherwise spend its 3 mantissa bits on the offset.
 *   wan_col_mean_bf16:   mean[b][c] = mean over rows [0, valid_rows) of sample b of x (bf16 [batch * rows_per_batch][ld]); two-stage,
 *                        fixed summation order (bit-reproducible); workspace of wan_col_mean_workspace_bytes(batch, dim) bytes.
 *   wan_qk_quantize_fp8: q8 = e4m3(q * q_scale), k8 = e4m3((k - k_mean[b]) * k_scale), dense [rows][dim] bytes; q, k are the
 *                        bf16 results of wan_rmsnorm_rope (q already carries softmax_scale * log2(e)); k_mean = NULL: no smoothing.
 *                        Either operand may be left out (q_bf16 = q8 = NULL or k_bf16 = k8 = NULL; ABI 8): under Ulysses k is
 *                        quantised once when it has arrived, every head group of q when its own exchange has completed. */
int64_t wan_col_mean_workspace_bytes(int batch, int dim);
wan_status_t wan_col_mean_bf16(const void* x_bf16, int64_t ld, int64_t rows_per_batch, int valid_rows, int batch, int dim,
                               void* workspace, float* mean, void* stream);
wan_status_t wan_qk_quantize_fp8(const void* q_bf16, const void* k_bf16, int64_t ld, int64_t rows, int dim, int64_t rows_per_batch,
                                 const float* k_mean, float q_scale, float k_scale, void* q8, void* k8, void* stream);

/* [rows, cols] bf16 (row stride ld) -> [cols, ldt] bf16 transposed; pad columns [rows, ldt) are zeroed.
 * Used when a caller hands attention() a row-major V (the reference's [B,L,N,D] layout). */
wan_status_t wan_transpose_bf16(const void* in, int64_t ld, void* out_t, int64_t ldt,
                                int64_t rows, int cols, void* stream);

/* ---------------------------------------------------------------------------
 * a1  patchify: latent [Cin,F,H,W] -> im2col tokens bf16 [L, Cin*pt*ph*pw], token order (f,h,w),
 *     K index (c,pt,ph,pw) c slowest = Conv3d(k=s=patch) weight.flatten(1) order
 *     (wan_transformer3d.py:662-663, 870-879).  in_dtype: 0 fp32, 1 bf16.
 * a14 unpatchify: head output fp32 [L, pt*ph*pw*Cout] (c fastest) -> [Cout, F*pt, H*ph, W*pw]
 *     (einsum 'fhwpqrc->cfphqwr', wan_transformer3d.py:1108-1131).  out_dtype: 0 fp32, 1 bf16.
 *     zero_frames: output frames [0, zero_frames) are written as 0 and their token rows are not read -- the
 *     VideoCoF mask `noise_pred[:, :, :condition_count] = 0` (pipeline_wan.py:736) folded into the store
 *     (exact also under classifier-free guidance: 0 + s * (0 - 0) = 0); 0 = plain unpatchify.
 * ------------------------------------------------------------------------- */
wan_status_t wan_patchify(const void* latent, int in_dtype, void* tokens_bf16, int64_t ldt,
                          int Cin, int F, int H, int W, int pt, int ph, int pw, void* stream);
wan_status_t wan_unpatchify(const float* tokens, int64_t ldt, void* out, int out_dtype,
                            int Cout, int F, int Hp, int Wp, int pt, int ph, int pw, int zero_frames, void* stream);

/* ---------------------------------------------------------------------------
 * a21  Ulysses sequence parallelism: the wire layouts of the head all-to-all (replaces usp_attn_forward's packing,
 *      videox_fun/dist/wan_xfuser.py:68-111, and yunchang's SeqAllToAll4D; the collective itself is an RCCL
 *      all_to_all_single with equal splits issued by the host, videocof_amd/dist.py).  One rank holds T tokens of B samples;
 *      P ranks; Cl = dim / P channels (H / P heads) per rank after the exchange.
 *        token-major wire   [P][T][B][Cl]  (q, k, and the attention output)
 *        channel-major wire [P][Cl][B][T]  (V^T: exactly what wan_gemm_bf16(WAN_EPI_BF16_T, ldo = B*T) writes per sample)
 *      After the exchange a token-major wire buffer reads as [P*T][B][Cl] -- all tokens, this rank's heads, row stride B*Cl,
 *      sample stride Cl -- which wan_attention_fwd consumes and produces in place (no unpacking of q, k or o).
 *      wan_rmsnorm_rope_sp: wan_rmsnorm_rope that reads x0 / x1 (not modified) and writes the results straight into
 *                           token-major wire buffers (slabs = P, batch = B; rows = B * rp->rows_per_batch): the fused form
 *                           of "norm, rotate, then pack for the all-to-all".
 *      wan_sp_pack_heads / wan_sp_unpack_heads: [B][T][ldx >= P*Cl] <-> token-major wire (the o projection's input).
 *      wan_sp_unpack_vt: arrived channel-major wire -> vt [B][Cl][ldvt], column s*T + t (ldvt >= P*T; pad columns untouched).
 *      Cl % 8 == 0, T % 8 == 0.
 *      *_split (head-group pipelining): the same kernels on a wire buffer cut into TWO head groups -- channels
 *                           [0, split) of every slab, then channels [split, Cl) -- each a complete token-major wire buffer of its
 *                           own ([P][T][B][split], then [P][T][B][Cl - split] right behind it), so that each group travels in its own
 *                           all-to-all: the exchange of group 1 runs under the attention of group 0, the inverse exchange of group 0
 *                           under the attention of group 1.  split % 8 == 0, 0 <= split < Cl; split = 0 is the one-group layout.
 * ------------------------------------------------------------------------- */
wan_status_t wan_rmsnorm_rope_sp(const void* x0_bf16, const float* w0, const void* x1_bf16, const float* w1,
                                 int64_t ld, int64_t rows, int dim, int head_dim, float eps,
                                 const float* rope_cos, const float* rope_sin, const wan_rope_params* rp,
                                 float x0_scale, void* wire0, void* wire1, int slabs, int batch, void* stream);
wan_status_t wan_sp_pack_heads(const void* x_bf16, int64_t ldx, void* wire, int P, int T, int B, int Cl, void* stream);
wan_status_t wan_sp_unpack_heads(const void* wire, void* x_bf16, int64_t ldx, int P, int T, int B, int Cl, void* stream);
wan_status_t wan_sp_unpack_vt(const void* wire, void* vt_bf16, int64_t ldvt, int P, int B, int Cl, int T, void* stream);
wan_status_t wan_rmsnorm_rope_sp_split(const void* x0_bf16, const float* w0, const void* x1_bf16, const float* w1,
                                       int64_t ld, int64_t rows, int dim, int head_dim, float eps,
                                       const float* rope_cos, const float* rope_sin, const wan_rope_params* rp,
                                       float x0_scale, void* wire0, void* wire1, int slabs, int batch, int split, void* stream);
wan_status_t wan_sp_pack_heads_split(const void* x_bf16, int64_t ldx, void* wire, int P, int T, int B, int Cl, int split, void* stream);
wan_status_t wan_sp_unpack_heads_split(const void* wire, void* x_bf16, int64_t ldx, int P, int T, int B, int Cl, int split, void* stream);

/* a21' The collective itself, owned by the library (for a host without torch.distributed; the Python host may use either):
 *      one communicator = one RCCL comm + ONE side HIP stream + a small ring of events (one per exchange in flight).
 *      replaces: set_multi_gpus_devices / init_distributed_environment (dist/fuser.py:35-54) and the head all-to-all inside
 *                xFuserLongContextAttention (dist/wan_xfuser.py:68-111).
 *      wan_sp_unique_id: rank 0 creates the 128-byte rendezvous token; the host distributes it (MPI, a file, torchrun's store).
 *      wan_sp_init: collective over all ranks (ncclCommInitRank) on the CURRENT HIP device; wan_sp_init_from_comm adopts an
 *                existing ncclComm_t (not destroyed by wan_sp_destroy).
 *      wan_sp_a2a_scatter_heads / wan_sp_a2a_gather_heads: the same operation on the wire layouts above -- slab d of `send_wire`
 *                (bytes_total / world_size bytes) goes to rank d, slab s of `recv_wire` arrives from rank s; the first name is
 *                for q / k / V^T (token shards -> head shards), the second for o (back).  ASYNCHRONOUS: enqueued on the side
 *                stream behind everything already enqueued on `compute_stream`; the caller goes on enqueueing the next
 *                projection and calls
 *      wan_sp_wait before the kernel that reads the receive buffers: `compute_stream` then waits (on the device) for every
 *                exchange started so far.  The host is never synchronised.  send / receive buffers must be distinct and stay
 *                untouched between start and wait (persistent pairs, as videocof_amd/wan_transformer3d.py keeps them).
 *      wan_sp_ticket / wan_sp_wait_for: the finer form -- wan_sp_ticket right after a start names that exchange, wan_sp_wait_for
 *                makes `compute_stream` wait for it (and, the side stream being in order, for every EARLIER one) while exchanges
 *                started later stay in flight: the k exchange can be consumed while the q exchange behind it still runs.  A
 *                ticket is good for a wait at any later time (a recycled event marks a later point of the side stream).  One
 *                host thread drives a communicator.
 *      wan_sp_all_gather: recv[r] <- send of rank r (the head output, wan_transformer3d.py:1085-1086); also asynchronous.
 *      Stream capture: while `compute_stream` is being captured into a hipGraph the collectives are recorded ON it (no side stream,
 *                no events; tuning key "sp_inline" forces that form).  Verified with ONE rank only (capture + bit-identical replay);
 *                several ranks replaying RCCL collectives from graphs in lockstep are untested -- the Python host refuses to capture
 *                a sequence-parallel forward with world_size > 1 unless WAN_SP_GRAPH_MULTI_RANK=1 (videocof_amd/graph.py).
 *      wan_sp_destroy: destroys the RCCL comm, the side stream and the events -- EXCEPT for a communicator that was ever captured:
 *                instantiated graphs hold its RCCL kernels, so its comm is deliberately leaked (it lives until the process exits).
 *      RCCL is bound at run time (dlopen "librccl.so.1"): WAN_ERR_UNSUPPORTED if it cannot be loaded. */
typedef struct wan_sp_comm wan_sp_comm;
#define WAN_SP_UNIQUE_ID_BYTES 128
wan_status_t wan_sp_unique_id(void* id128);
wan_status_t wan_sp_init(wan_sp_comm** comm, const void* id128, int rank, int world_size);
wan_status_t wan_sp_init_from_comm(wan_sp_comm** comm, void* nccl_comm, int rank, int world_size);
int wan_sp_rank(const wan_sp_comm* comm);
int wan_sp_world_size(const wan_sp_comm* comm);
wan_status_t wan_sp_a2a_scatter_heads(wan_sp_comm* comm, const void* send_wire, void* recv_wire, int64_t bytes_total, void* compute_stream);
wan_status_t wan_sp_a2a_gather_heads(wan_sp_comm* comm, const void* send_wire, void* recv_wire, int64_t bytes_total, void* compute_stream);
wan_status_t wan_sp_all_gather(wan_sp_comm* comm, const void* send, void* recv, int64_t bytes_per_rank, void* compute_stream);
wan_status_t wan_sp_wait(wan_sp_comm* comm, void* compute_stream);
int64_t wan_sp_ticket(const wan_sp_comm* comm);
wan_status_t wan_sp_wait_for(wan_sp_comm* comm, int64_t ticket, void* compute_stream);
wan_status_t wan_sp_destroy(wan_sp_comm* comm);

/* ---------------------------------------------------------------------------
 * a11  One WanAttentionBlock as a single call (WanAttentionBlock.forward, wan_transformer3d.py:464-515) -- the
 *      composite a non-Python host drives: LN-modulate -> q|k GEMM -> RMSNorm+RoPE (q pre-scaled) -> V^T GEMM ->
 *      self-attention -> o GEMM (+gate, +residual) -> LN-affine -> q GEMM -> RMSNorm -> cross-attention over the text
 *      K / V^T -> o GEMM (+residual) -> LN-modulate -> ffn.0 (+GELU) -> ffn.2 (+gate, +residual); the kernels above,
 *      enqueued in that order on `stream` (host code only, no extra arithmetic).  Single device (no sequence parallelism).
 *      x      fp32 [batch * rows_per_batch, dim]   residual stream, updated IN PLACE
 *      emod   fp32 [6][batch][dim]                 modulation + time projection (`(self.modulation + e).chunk(6)`, :494)
 *      ctx_k  bf16 [batch][text_len][dim]          RMS-normed text keys   (cross_attn.norm_k(k(context)), :321)
 *      ctx_vt bf16 [batch][dim][text_len]          text values, transposed (cross_attn.v(context), :322)
 *      valid_tokens = F*Hp*Wp <= rows_per_batch: keys beyond it (sequence padding) are masked, V^T covers them only.
 *      Workspaces are caller-owned (sizes: wan_dit_block_workspace_bytes); vt's pad columns [valid_tokens, ldvt) must be
 *      finite (zero them once); the two attention scratches follow wan_attention_workspace_bytes and may be NULL.
 * ------------------------------------------------------------------------- */
typedef struct {
    int dim, ffn_dim, num_heads, text_len;
    float eps;
    const void *w_qk, *w_v, *w_o, *w_cq, *w_co, *w_ffn0, *w_ffn2;      /* bf16 [out, in]; w_qk = q rows then k rows */
    const float *b_qk, *b_v, *b_o, *b_cq, *b_co, *b_ffn0, *b_ffn2;
    const float *norm_q, *norm_k, *norm_cq;                           /* WanRMSNorm gains */
    const float *norm3_w, *norm3_b;                                   /* LayerNorm (affine) before cross-attention */
} wan_block_weights;

typedef struct {
    void *h, *qk, *att, *cq, *ff, *vt;        /* bf16: [M,dim] [M,2 dim] [M,dim] [M,dim] [M,ffn] [batch][dim][ldvt] */
    int64_t ldvt;
    void* attn_ws_self; int64_t attn_ws_self_bytes;
    void* attn_ws_cross; int64_t attn_ws_cross_bytes;
    void* gemm_ws; int64_t gemm_ws_bytes;      /* wan_gemm_bf16_ws workspace shared by the block's Linears (wan_gemm_workspace_bytes of the
                                                  largest one; NULL = the one-workgroup-per-tile kernels) */
} wan_block_workspace;

wan_status_t wan_dit_block_forward(float* x, const float* emod, const void* ctx_k, const void* ctx_vt,
                                   const wan_block_weights* w, const wan_block_workspace* ws,
                                   const float* rope_cos, const float* rope_sin, const wan_rope_params* rp,
                                   int batch, int64_t rows_per_batch, int64_t valid_tokens, void* stream);
/* The token-local part of a block -- everything behind the self-attention output projection (wan_transformer3d.py:504-511:
 * norm3, cross-attention over the text tokens, FFN) -- as one call: the tail of wan_dit_block_forward, and what a
 * sequence-parallel host calls after its own self-attention part (it drives the exchanges itself; dist/wan_xfuser.py:68-111).
 * Uses ws->h, att, cq, ff, attn_ws_cross, gemm_ws only (qk / vt may be NULL).  ABI 8. */
wan_status_t wan_dit_block_tail_forward(float* x, const float* emod, const void* ctx_k, const void* ctx_vt,
                                        const wan_block_weights* w, const wan_block_workspace* ws,
                                        int batch, int64_t rows_per_batch, void* stream);
wan_status_t wan_dit_block_workspace_bytes(int dim, int ffn_dim, int batch, int64_t rows_per_batch, int64_t valid_tokens,
                                           int64_t* bytes /* [6]: h, qk, att, cq, ff, vt */, int64_t* ldvt);

/* ---------------------------------------------------------------------------
 * a11' The token path of WanTransformer3DModel.forward (wan_transformer3d.py:870-879, 1034-1083, 535-548, 1108-1131) as one
 *      call: patchify -> patch-embedding GEMM -> num_layers x wan_dit_block_forward -> head LN-modulate -> head GEMM ->
 *      unpatchify.  The embedding MLPs stay with the host and arrive as inputs:
 *      latent  [batch][in_dim][F][H][W] (latent_dtype 0 fp32 | 1 bf16); out the same shape with out_dim channels
 *      emod    fp32 [num_layers][6][batch][dim]   `modulation + time_projection(e)` of every block (:494, :899-901)
 *      ehead   fp32 [2][batch][dim]               `head.modulation + e` (shift, scale; :545)
 *      ctx_k / ctx_vt  host arrays of num_layers device pointers: the text K and V^T of each block (wan_dit_block_forward)
 *      rp      grid = (F/pt, H/ph, W/pw); rows_per_batch >= tokens (pad rows are zero on entry to the first block)
 *      zero_frames as wan_unpatchify.  Workspace: the block workspace plus x fp32 [batch*rows_per_batch, dim],
 *      tokens bf16 [tokens, in_dim*pt*ph*pw], head_out fp32 [batch*rows_per_batch, pt*ph*pw*out_dim].
 * ------------------------------------------------------------------------- */
typedef struct {
    int num_layers, in_dim, out_dim, pt, ph, pw;
    const wan_block_weights* blocks;                /* [num_layers] */
    const void* pe_w; const float* pe_b;            /* patch_embedding as a GEMM: bf16 [dim, in_dim*pt*ph*pw] */
    const void* head_w; const float* head_b;        /* head.head: bf16 [pt*ph*pw*out_dim, dim] */
} wan_dit_weights;

typedef struct {
    wan_block_workspace block;
    float* x;
    void* tokens;
    float* head_out;
} wan_dit_workspace;

wan_status_t wan_dit_forward(const void* latent, int latent_dtype, void* out, int out_dtype, const float* emod,
                             const float* ehead, const void* const* ctx_k, const void* const* ctx_vt,
                             const wan_dit_weights* w, const wan_dit_workspace* ws, const float* rope_cos,
                             const float* rope_sin, const wan_rope_params* rp, int batch, int F, int H, int W,
                             int64_t rows_per_batch, int zero_frames, void* stream);

/* a17  UniPC updates as one fused pass: out[i] = c0*x0[i] + c1*x1[i] + c2*x2[i] + c3*x3[i] (x1..x3 may be
 *      NULL), fp32 accumulate, all tensors of one dtype (0 fp32, 1 bf16).
 *      replaces: the elementwise chains of convert_model_output / multistep_uni_p_bh_update /
 *      multistep_uni_c_bh_update (fm_solvers_unipc.py:318-320, 458-470, 600-612); the scalar algebra
 *      stays on the host (float64). */
wan_status_t wan_lincomb(void* out, int dtype, const void* x0, const void* x1, const void* x2, const void* x3,
                         float c0, float c1, float c2, float c3, int64_t n, void* stream);

/* ===========================================================================
 * WanVAE (videox_fun/models/wan_vae.py).  Activations are CHANNELS-LAST bf16 [T, H, W, C].
 * ------------------------------------------------------------------------- */

/* a18/a19  every convolution of the VAE through one entry (two kernels behind it: the causal 3x3x3 / stride-1 convolutions with
 *     Cin % 32 == 0 and Cout % 96 == 0 keep their input patch in LDS; everything else is an implicit GEMM with a gathered A tile):
 *     out[(to,ho,wo), n] = bias[n] + sum_{kt,kh,kw,ci} in[to*st+kt-pt, ho*sh+kh-ph, wo*sw+kw-pw, ci] * w[n,(kt,kh,kw,ci)]
 *                          (+ resid[(to,ho,wo), n])
 *     replaces: CausalConv3d.forward incl. the cache_x halo (wan_vae.py:21-40: frames with negative
 *               time index are read from `hist`, the last `hist_frames` (<= 2) frames of the previous
 *               chunk, CACHE_T :18; missing history = the zero padding of :38);
 *               Resample's Conv2d: upsample2x=1 fuses Upsample(nearest-exact, 2x) (:81-83) into the
 *               gather, ZeroPad2d((0,1,0,1)) + stride 2 (:92-94) is ph=pw=0, sh=sw=2 with zero fill;
 *               upsample3d's time_conv + the channel->time interleave (:132-141): time_interleave=1
 *               writes channel n of frame t to frame 2t + n/(Cout/2), channel n%(Cout/2);
 *               ResidualBlock's `x + h` (:224) through `resid`; all 1x1 convs (:203, 238-239, 509-510).
 *     x    bf16 [T_in, H_in, W_in, Cin], Cin % 8 == 0
 *     w    bf16 [Cout, ldw] with k = ((kt*KH + kh)*KW + kw)*Cin + ci, zero padded to ldw >= roundup(K, 64)
 *     out  bf16 [T_out*H_out*W_out, ldo]  (time_interleave: [2*T_out*H_out*W_out, ldo], Cout/2 channels) */
typedef struct {
    int T_in, H_in, W_in, Cin;
    int T_out, H_out, W_out, Cout;
    int KT, KH, KW;          /* each 1 or 3 */
    int st, sh, sw;          /* strides */
    int pt, ph, pw;          /* leading pads (time pad = causal front pad) */
    int upsample2x;          /* gather from a virtual nearest-exact 2x upsample of (H_in, W_in) */
    int time_interleave;
} wan_conv_params;

wan_status_t wan_conv_cl(const void* x, const void* hist, int hist_frames, const void* w, int64_t ldw,
                         const float* bias, const void* resid, void* out, int64_t ldo,
                         const wan_conv_params* p, void* stream);

/* a19  RMS_norm (F.normalize over channels * sqrt(C) * gamma, wan_vae.py:43-58) + optional SiLU,
 *      per pixel on channels-last rows.  C % 8 == 0, C <= 512. */
wan_status_t wan_rmsnorm_silu_cl(const void* x_bf16, const float* gamma, void* out_bf16, int64_t rows, int C,
                                 int silu, void* stream);

/* a19  row softmax for AttentionBlock (single head of dim C over h*w, wan_vae.py:256-260):
 *      probs[r, i] = softmax_i(scale * scores[r, i]) for i < n, 0 for n <= i < npad. */
wan_status_t wan_softmax_rows(const float* scores, int64_t lds, void* probs_bf16, int64_t ldp, int64_t rows,
                              int n, int npad, float scale, void* stream);

/* boundary layout: planar [Cv, npix] (the reference's [C,T,H,W]) <-> channels-last [npix, Cpad] bf16.
 * dtype: 0 fp32, 1 bf16.  cl_to_video optionally clamps to [-1, 1] (wan_vae.py:669). */
wan_status_t wan_video_to_cl(const void* video, int in_dtype, void* out_bf16, int Cv, int Cpad, int64_t npix, void* stream);
wan_status_t wan_cl_to_video(const void* x_bf16, int64_t ld, void* out, int out_dtype, int Cv, int64_t npix,
                             int clamp, void* stream);

/* ===========================================================================
 * the umT5 text encoder (videox_fun/models/wan_text_encoder.py:256-304), the step
 * before the denoising path.  Its Linear layers are wan_gemm_bf16; the rest:
 * ------------------------------------------------------------------------- */

/* Strided-batched nn.Linear-style product: for z in [0, batch):
 *     out_z[m, n] = sum_k A_z[m, k] * W_z[n, k],  X_z = X + z * strideX (elements)
 * replaces: torch.einsum('binc,bjnc->bnij', q, k) and einsum('bnij,bjnc->binc', attn, v)
 *           of T5Attention.forward (wan_text_encoder.py:102-105), one problem per head.
 * epilogue: WAN_EPI_BF16 or WAN_EPI_F32, no bias.  K % 64 == 0, N % 4 == 0, strides of A/W % 8 == 0. */
wan_status_t wan_gemm_bf16_batched(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw,
                                   int64_t strideW, void* out, int64_t ldo, int64_t strideO,
                                   int M, int N, int K, int batch, int epilogue, void* stream);

/* nn.Embedding gather (wan_text_encoder.py:286-287): out fp32 [rows, dim] = table_bf16[ids[r], :].
 * ids are clamped into [0, vocab) for memory safety; the caller validates the range. */
wan_status_t wan_embedding_rows(const int64_t* ids, const void* table_bf16, int64_t vocab, float* out,
                                int64_t rows, int dim, void* stream);

/* T5LayerNorm (wan_text_encoder.py:48-60): out = x * rsqrt(mean(x^2) + eps) * w, x fp32 [rows, dim],
 * w fp32 [dim]; out_dtype 0 fp32, 1 bf16.  dim % 4 == 0, dim <= 8192. */
wan_status_t wan_rmsnorm_rows(const float* x, const float* w, void* out, int out_dtype, int64_t rows, int dim,
                              float eps, void* stream);

/* T5Attention score softmax (wan_text_encoder.py:93-104) with the relative-position bias of
 * T5RelativeEmbedding (:226-260) evaluated in place:
 *     probs[h, i, j] = softmax_j(scores[h, i, j] + bucket_table[bucket_lut[j - i + Lq - 1], h]),  j < k_len
 * and 0 for k_len <= j < npad (attention_mask == 0 keys, :98-99, and the K padding of the P.V product).
 * scores fp32 [H*Lq, lds]; bucket_table fp32 [num_buckets, H] (pos_embedding.embedding.weight);
 * bucket_lut int32 [Lq + Lk - 1] = _relative_position_bucket(j - i) (:245-264); probs bf16 [H*Lq, ldp]. */
wan_status_t wan_t5_softmax_bias(const float* scores, int64_t lds, const float* bucket_table,
                                 const int* bucket_lut, void* probs_bf16, int64_t ldp, int num_heads,
                                 int Lq, int Lk, int k_len, int npad, void* stream);

/* gated-GELU feed-forward product fc1(x) * gelu(gate(x)) (wan_text_encoder.py:125-126): out = a * b, bf16, n % 8 == 0. */
wan_status_t wan_mul_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);

/* ---------------------------------------------------------------------------
 * Box fingerprint for benchmarks (no reference counterpart: the reference has no benchmark harness).  A FIXED calibration
 * workload, independent of the product kernels: (a) ~target_ms of 32x32x16 bf16 MFMAs on random operands beside LDS fragment
 * reads and a softmax VALU stream (the instruction mix of a flash-attention tile), one 4-wave workgroup per CU -- what the
 * matrix pipes of THIS box hold at its power limit; (b) 40 passes of a 256 MiB -> 256 MiB copy.  Unlike every other entry
 * point this call SYNCHRONISES the stream (it reads HIP events) -- it is a measurement, never part of a timed region.
 * scratch: device memory of >= wan_box_probe_scratch_bytes(), caller-owned.  target_ms <= 0 selects 300.
 * bench.py runs it before and after the timed region (`box` object, `value_normalised`).
 * ------------------------------------------------------------------------- */
typedef struct {
    float mfma_mix_tflops;   /* issued MFMA FLOP / time of the last two of three equal launches */
    float copy_tbps;         /* (read + write) bytes / time */
    float mfma_ms, copy_ms;  /* the measured intervals */
} wan_box_probe_result;
int64_t wan_box_probe_scratch_bytes(void);
wan_status_t wan_box_probe(wan_box_probe_result* result, void* scratch, int64_t scratch_bytes, int target_ms, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WAN_HIP_H */
