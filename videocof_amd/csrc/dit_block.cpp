// Composite entry point: one WanAttentionBlock (videox_fun/models/wan_transformer3d.py:464-515) as ONE C call, so that a
// host in any language can drive the path without re-stating the launch sequence, and the Python host crosses the FFI
// once per block instead of 15 times.  Pure host code: it enqueues the kernels of this library on the caller's stream
// through their own C entry points, in exactly the order videocof_amd/wan_transformer3d.py documents; nothing is computed here.
#include "common.hpp"

namespace {
inline const char* at(const void* p, int64_t bytes) { return (const char*)p + bytes; }
inline char* at(void* p, int64_t bytes) { return (char*)p + bytes; }
}  // namespace

#define WAN_TRY(call)                   \
    do {                                \
        const wan_status_t st__ = (call); \
        if (st__ != WAN_OK) return st__; \
    } while (0)

extern "C" wan_status_t wan_dit_block_tail_forward(float* x, const float* emod, const void* ctx_k, const void* ctx_vt,
                                                   const wan_block_weights* w, const wan_block_workspace* ws,
                                                   int batch, int64_t rows_per_batch, void* stream);

extern "C" wan_status_t wan_dit_block_forward(float* x, const float* emod, const void* ctx_k, const void* ctx_vt,
                                              const wan_block_weights* w, const wan_block_workspace* ws,
                                              const float* rope_cos, const float* rope_sin, const wan_rope_params* rp,
                                              int batch, int64_t rows_per_batch, int64_t valid_tokens, void* stream) {
    WAN_REQUIRE(x && emod && ctx_k && ctx_vt && w && ws && rope_cos && rope_sin && rp, WAN_ERR_INVALID,
                "wan_dit_block_forward: null argument");
    WAN_REQUIRE(batch > 0 && rows_per_batch > 0 && valid_tokens > 0 && valid_tokens <= rows_per_batch, WAN_ERR_INVALID,
                "wan_dit_block_forward: batch=%d rows_per_batch=%lld valid_tokens=%lld", batch, (long long)rows_per_batch,
                (long long)valid_tokens);
    const int C = w->dim, F = w->ffn_dim, H = w->num_heads, T = w->text_len;
    WAN_REQUIRE(C > 0 && H > 0 && C == H * 128 && F > 0 && T > 0, WAN_ERR_UNSUPPORTED,
                "wan_dit_block_forward: dim=%d heads=%d (head_dim must be 128) ffn=%d text_len=%d", C, H, F, T);
    WAN_REQUIRE(ws->h && ws->qk && ws->att && ws->cq && ws->ff && ws->vt, WAN_ERR_INVALID, "wan_dit_block_forward: null workspace");
    void* const GWS = ws->gemm_ws;                                // every Linear of the block shares one GEMM workspace (they run back to back)
    const int64_t GWSB = ws->gemm_ws_bytes;
    const int64_t Ll = rows_per_batch, M = (int64_t)batch * Ll;
    WAN_REQUIRE(M <= 0x7fffffff && valid_tokens <= 0x7fffffff, WAN_ERR_UNSUPPORTED, "wan_dit_block_forward: too many rows");
    const int64_t lk_pad = (valid_tokens + 63) / 64 * 64;
    WAN_REQUIRE(ws->ldvt >= lk_pad && ws->ldvt % 8 == 0, WAN_ERR_INVALID,
                "wan_dit_block_forward: ldvt=%lld must be >= roundup(valid_tokens,64)=%lld and a multiple of 8", (long long)ws->ldvt,
                (long long)lk_pad);
    const float eps = w->eps;
    const float qs = (float)(0.08838834764831845 * 1.4426950408889634);   // 1/sqrt(128) * log2(e) (WAN_ATTN_QSCALE), folded into q by the norm kernel
    const int64_t bC = (int64_t)batch * C;                        // one modulation row block [B][C]
    const float* shift_msa = emod, *scale_msa = emod + bC, *gate_msa = emod + 2 * bC;
    void* qk = ws->qk;
    void* kpart = at(qk, (int64_t)C * 2);                         // k columns of the fused q|k rows (bf16)
    wan_rope_params rope = *rp;
    rope.rows_per_batch = Ll;

    // ---- self-attention (:495-499)
    WAN_TRY(wan_ln_modulate(x, scale_msa, shift_msa, 1, ws->h, M, C, Ll, eps, stream));
    WAN_TRY(wan_gemm_bf16_ws(ws->h, C, w->w_qk, C, w->b_qk, qk, 2 * C, (int)M, 2 * C, C, WAN_EPI_BF16, nullptr, 0, GWS, GWSB, stream));
    WAN_TRY(wan_rmsnorm_rope(qk, w->norm_q, kpart, w->norm_k, 2 * C, M, C, 128, eps, rope_cos, rope_sin, &rope, qs, stream));
    for (int b = 0; b < batch; ++b)
        WAN_TRY(wan_gemm_bf16_ws(at(ws->h, (int64_t)b * Ll * C * 2), C, w->w_v, C, w->b_v, at(ws->vt, (int64_t)b * C * ws->ldvt * 2),
                              ws->ldvt, (int)valid_tokens, C, C, WAN_EPI_BF16_T, nullptr, 0, GWS, GWSB, stream));
    WAN_TRY(wan_attention_fwd(qk, 2 * C, Ll * 2 * C, kpart, 2 * C, Ll * 2 * C, ws->vt, ws->ldvt, (int64_t)C * ws->ldvt, ws->att, C, Ll * C,
                              batch, (int)Ll, (int)valid_tokens, H, 128, 0.f, WAN_ATTN_Q_PRESCALED, ws->attn_ws_self,
                              ws->attn_ws_self_bytes, stream));
    WAN_TRY(wan_gemm_bf16_ws(ws->att, C, w->w_o, C, w->b_o, x, C, (int)M, C, C, WAN_EPI_RESID_F32, gate_msa, Ll, GWS, GWSB, stream));
    // ---- cross-attention (:504) and FFN (:507-511): the token-local part of the block
    return wan_dit_block_tail_forward(x, emod, ctx_k, ctx_vt, w, ws, batch, rows_per_batch, stream);
}

// The token-local two thirds of a WanAttentionBlock -- everything behind the self-attention's output projection
// (videox_fun/models/wan_transformer3d.py:504-511: norm3 -> cross-attention over the text tokens -> FFN) -- as ONE C call.  The tail of
// wan_dit_block_forward, and what a sequence-parallel host calls after its own self-attention part (whose exchanges it drives
// itself): an Ulysses layer is then ~20 FFI crossings instead of ~30.  Needs ws->h, att, cq, ff, attn_ws_cross, gemm_ws only.
extern "C" wan_status_t wan_dit_block_tail_forward(float* x, const float* emod, const void* ctx_k, const void* ctx_vt,
                                                   const wan_block_weights* w, const wan_block_workspace* ws,
                                                   int batch, int64_t rows_per_batch, void* stream) {
    WAN_REQUIRE(x && emod && ctx_k && ctx_vt && w && ws, WAN_ERR_INVALID, "wan_dit_block_tail_forward: null argument");
    WAN_REQUIRE(batch > 0 && rows_per_batch > 0, WAN_ERR_INVALID, "wan_dit_block_tail_forward: batch=%d rows_per_batch=%lld", batch,
                (long long)rows_per_batch);
    const int C = w->dim, F = w->ffn_dim, H = w->num_heads, T = w->text_len;
    WAN_REQUIRE(C > 0 && H > 0 && C == H * 128 && F > 0 && T > 0, WAN_ERR_UNSUPPORTED,
                "wan_dit_block_tail_forward: dim=%d heads=%d (head_dim must be 128) ffn=%d text_len=%d", C, H, F, T);
    WAN_REQUIRE(ws->h && ws->att && ws->cq && ws->ff, WAN_ERR_INVALID, "wan_dit_block_tail_forward: null workspace");
    void* const GWS = ws->gemm_ws;
    const int64_t GWSB = ws->gemm_ws_bytes;
    const int64_t Ll = rows_per_batch, M = (int64_t)batch * Ll;
    WAN_REQUIRE(M <= 0x7fffffff, WAN_ERR_UNSUPPORTED, "wan_dit_block_tail_forward: too many rows");
    const float eps = w->eps;
    const float qs = (float)(0.08838834764831845 * 1.4426950408889634);
    const int64_t bC = (int64_t)batch * C;
    const float* shift_mlp = emod + 3 * bC, *scale_mlp = emod + 4 * bC, *gate_mlp = emod + 5 * bC;
    // ---- cross-attention over the text tokens (:504; rows are not masked, context_lens = None)
    WAN_TRY(wan_ln_modulate(x, w->norm3_w, w->norm3_b, 0, ws->h, M, C, M, eps, stream));
    WAN_TRY(wan_gemm_bf16_ws(ws->h, C, w->w_cq, C, w->b_cq, ws->cq, C, (int)M, C, C, WAN_EPI_BF16, nullptr, 0, GWS, GWSB, stream));
    WAN_TRY(wan_rmsnorm_rope(ws->cq, w->norm_cq, nullptr, nullptr, C, M, C, 128, eps, nullptr, nullptr, nullptr, qs, stream));
    WAN_TRY(wan_attention_fwd(ws->cq, C, Ll * C, ctx_k, C, (int64_t)T * C, ctx_vt, T, (int64_t)C * T, ws->att, C, Ll * C, batch, (int)Ll,
                              T, H, 128, 0.f, WAN_ATTN_Q_PRESCALED, ws->attn_ws_cross, ws->attn_ws_cross_bytes, stream));
    WAN_TRY(wan_gemm_bf16_ws(ws->att, C, w->w_co, C, w->b_co, x, C, (int)M, C, C, WAN_EPI_RESID_F32, nullptr, 0, GWS, GWSB, stream));
    // ---- FFN (:507-511)
    WAN_TRY(wan_ln_modulate(x, scale_mlp, shift_mlp, 1, ws->h, M, C, Ll, eps, stream));
    WAN_TRY(wan_gemm_bf16_ws(ws->h, C, w->w_ffn0, C, w->b_ffn0, ws->ff, F, (int)M, F, C, WAN_EPI_GELU_BF16, nullptr, 0, GWS, GWSB, stream));
    WAN_TRY(wan_gemm_bf16_ws(ws->ff, F, w->w_ffn2, F, w->b_ffn2, x, C, (int)M, C, F, WAN_EPI_RESID_F32, gate_mlp, Ll, GWS, GWSB, stream));
    return WAN_OK;
}

extern "C" wan_status_t wan_dit_block_workspace_bytes(int dim, int ffn_dim, int batch, int64_t rows_per_batch, int64_t valid_tokens,
                                                      int64_t* bytes /* [6]: h, qk, att, cq, ff, vt */, int64_t* ldvt) {
    WAN_REQUIRE(bytes && ldvt && dim > 0 && ffn_dim > 0 && batch > 0 && rows_per_batch > 0 && valid_tokens > 0, WAN_ERR_INVALID,
                "wan_dit_block_workspace_bytes: bad argument");
    const int64_t M = (int64_t)batch * rows_per_batch;
    *ldvt = (valid_tokens + 63) / 64 * 64;
    bytes[0] = M * dim * 2; bytes[1] = M * 2 * dim * 2; bytes[2] = M * dim * 2; bytes[3] = M * dim * 2;
    bytes[4] = M * ffn_dim * 2; bytes[5] = (int64_t)batch * dim * *ldvt * 2;
    return WAN_OK;
}

// The token path of WanTransformer3DModel.forward (videox_fun/models/wan_transformer3d.py:870-879 patch embedding,
// :1034-1083 the block loop, :535-548 head, :1108-1131 unpatchify) as ONE C call: patchify -> patch-embedding GEMM into the
// fp32 residual stream -> num_layers x wan_dit_block_forward -> head LN-modulate -> head GEMM -> unpatchify (with the CoF
// mask).  The two embedding MLPs that feed it (time: :891-901, text: :915-919 -- a few GFLOP per call, fp32 / bf16 GEMMs of
// this library or the host's) stay with the host: their results arrive as `emod`, `ehead`, `ctx_k`, `ctx_vt`.
extern "C" wan_status_t wan_dit_forward(const void* latent, int latent_dtype, void* out, int out_dtype, const float* emod,
                                        const float* ehead, const void* const* ctx_k, const void* const* ctx_vt,
                                        const wan_dit_weights* w, const wan_dit_workspace* ws, const float* rope_cos,
                                        const float* rope_sin, const wan_rope_params* rp, int batch, int F, int H, int W,
                                        int64_t rows_per_batch, int zero_frames, void* stream) {
    WAN_REQUIRE(latent && out && emod && ehead && ctx_k && ctx_vt && w && ws && rp, WAN_ERR_INVALID, "wan_dit_forward: null argument");
    WAN_REQUIRE(w->num_layers > 0 && w->blocks && w->pe_w && w->pe_b && w->head_w && w->head_b, WAN_ERR_INVALID,
                "wan_dit_forward: incomplete weights");
    WAN_REQUIRE(ws->x && ws->tokens && ws->head_out, WAN_ERR_INVALID, "wan_dit_forward: null workspace");
    void* const GWS = ws->block.gemm_ws;
    const int64_t GWSB = ws->block.gemm_ws_bytes;
    const int pt = w->pt, ph = w->ph, pw = w->pw;
    WAN_REQUIRE(pt > 0 && ph > 0 && pw > 0 && F > 0 && H > 0 && W > 0 && F % pt == 0 && H % ph == 0 && W % pw == 0, WAN_ERR_INVALID,
                "wan_dit_forward: latent (%d,%d,%d) is not a multiple of the patch (%d,%d,%d)", F, H, W, pt, ph, pw);
    const int gf = F / pt, gh = H / ph, gw = W / pw;
    const int64_t L = (int64_t)gf * gh * gw, Ll = rows_per_batch;
    WAN_REQUIRE(batch > 0 && L <= Ll && (int64_t)batch * Ll <= 0x7fffffff, WAN_ERR_INVALID,
                "wan_dit_forward: batch=%d tokens=%lld rows_per_batch=%lld", batch, (long long)L, (long long)Ll);
    WAN_REQUIRE(rp->F == gf && rp->Hp == gh && rp->Wp == gw, WAN_ERR_INVALID, "wan_dit_forward: rope grid (%d,%d,%d) != latent grid (%d,%d,%d)",
                rp->F, rp->Hp, rp->Wp, gf, gh, gw);
    const int C = w->blocks[0].dim;
    const int pv = pt * ph * pw, Kpe = w->in_dim * pv, Nh = w->out_dim * pv;
    const int64_t M = (int64_t)batch * Ll;
    const int64_t lat_el = latent_dtype == 0 ? 4 : 2, out_el = out_dtype == 0 ? 4 : 2;
    WAN_REQUIRE((latent_dtype == 0 || latent_dtype == 1) && (out_dtype == 0 || out_dtype == 1), WAN_ERR_INVALID,
                "wan_dit_forward: dtypes must be 0 (fp32) or 1 (bf16)");

    // pad rows (token >= L) enter the stream as zeros, exactly as torch.cat([u, u.new_zeros(...)]) leaves them (:907-909)
    WAN_REQUIRE(hipMemsetAsync(ws->x, 0, (size_t)M * C * 4, (hipStream_t)stream) == hipSuccess, WAN_ERR_LAUNCH,
                "wan_dit_forward: hipMemsetAsync failed");
    for (int b = 0; b < batch; ++b) {
        WAN_TRY(wan_patchify(at(latent, (int64_t)b * w->in_dim * F * H * W * lat_el), latent_dtype, ws->tokens, Kpe, w->in_dim, F, H, W,
                             pt, ph, pw, stream));
        WAN_TRY(wan_gemm_bf16_ws(ws->tokens, Kpe, w->pe_w, Kpe, w->pe_b, ws->x + (int64_t)b * Ll * C, C, (int)L, C, Kpe, WAN_EPI_F32,
                              nullptr, 0, GWS, GWSB, stream));
    }
    const int64_t emod_layer = 6 * (int64_t)batch * C;
    for (int l = 0; l < w->num_layers; ++l) {
        WAN_REQUIRE(w->blocks[l].dim == C, WAN_ERR_INVALID, "wan_dit_forward: block %d has dim %d != %d", l, w->blocks[l].dim, C);
        WAN_TRY(wan_dit_block_forward(ws->x, emod + l * emod_layer, ctx_k[l], ctx_vt[l], &w->blocks[l], &ws->block, rope_cos, rope_sin, rp,
                                      batch, Ll, L, stream));
    }
    const int64_t bC = (int64_t)batch * C;                    // ehead = [shift][scale], each [batch][C] (Head.forward :545-547)
    WAN_TRY(wan_ln_modulate(ws->x, ehead + bC, ehead, 1, ws->block.h, M, C, Ll, w->blocks[0].eps, stream));
    WAN_TRY(wan_gemm_bf16_ws(ws->block.h, C, w->head_w, C, w->head_b, ws->head_out, Nh, (int)M, Nh, C, WAN_EPI_F32, nullptr, 0, GWS, GWSB, stream));
    for (int b = 0; b < batch; ++b)
        WAN_TRY(wan_unpatchify(ws->head_out + (int64_t)b * Ll * Nh, Nh, at(out, (int64_t)b * w->out_dim * F * H * W * out_el), out_dtype,
                               w->out_dim, gf, gh, gw, pt, ph, pw, zero_frames, stream));
    return WAN_OK;
}
