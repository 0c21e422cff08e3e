// Layout glue at the two ends of the DiT: patchify (im2col of the (1,2,2) Conv3d) and unpatchify.
// Pure data movement (HBM-bound, a few MB per forward).
#include <algorithm>

#include "common.hpp"

namespace {

template <typename TIn>
__global__ __launch_bounds__(256) void patchify_kernel(const TIn* __restrict__ x, bf16_t* __restrict__ tok,
                                                       int64_t ldt, int Cin, int F, int H, int W,
                                                       int pt, int ph, int pw) {
    // one thread per output element; consecutive threads walk the K index of one token then the next token
    const int Fp = F / pt, Hp = H / ph, Wp = W / pw;
    const int K = Cin * pt * ph * pw;
    const int64_t total = (int64_t)Fp * Hp * Wp * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / K;
        int kk = (int)(i - t * K);
        const int c = kk / (pt * ph * pw); kk -= c * pt * ph * pw;
        const int a = kk / (ph * pw); kk -= a * ph * pw;
        const int b = kk / pw, d = kk - b * pw;
        const int wq = (int)(t % Wp);
        const int hq = (int)((t / Wp) % Hp);
        const int fq = (int)(t / ((int64_t)Wp * Hp));
        const float v = (float)x[(((int64_t)c * F + fq * pt + a) * H + hq * ph + b) * W + wq * pw + d];
        tok[t * ldt + (i - t * K)] = (bf16_t)v;
    }
}

template <typename TOut>
__global__ __launch_bounds__(256) void unpatchify_kernel(const float* __restrict__ tok, int64_t ldt,
                                                         TOut* __restrict__ out, int Cout, int Fp, int Hp, int Wp,
                                                         int pt, int ph, int pw, int zero_frames) {
    // one thread per OUTPUT element (coalesced writes along W); 'fhwpqrc->cfphqwr'
    const int F = Fp * pt, H = Hp * ph, W = Wp * pw;
    const int64_t total = (int64_t)Cout * F * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); r /= H;
        const int f = (int)(r % F); r /= F;
        const int c = (int)r;
        const int fq = f / pt, a = f - fq * pt;
        const int hq = h / ph, b = h - hq * ph;
        const int wq = w / pw, d = w - wq * pw;
        const int64_t t = ((int64_t)fq * Hp + hq) * Wp + wq;
        const int col = ((a * ph + b) * pw + d) * Cout + c;
        out[i] = f < zero_frames ? (TOut)0.f : (TOut)tok[t * ldt + col];    // CoF mask (pipeline_wan.py:736), no token read
    }
}

}  // namespace

extern "C" wan_status_t wan_patchify(const void* latent, int in_dtype, void* tokens_bf16, int64_t ldt,
                                     int Cin, int F, int H, int W, int pt, int ph, int pw, void* stream) {
    WAN_REQUIRE(latent && tokens_bf16, WAN_ERR_INVALID, "wan_patchify: null tensor");
    WAN_REQUIRE(Cin > 0 && F > 0 && H > 0 && W > 0 && pt > 0 && ph > 0 && pw > 0, WAN_ERR_INVALID, "wan_patchify: bad shape");
    WAN_REQUIRE(F % pt == 0 && H % ph == 0 && W % pw == 0, WAN_ERR_INVALID,
                "wan_patchify: (%d,%d,%d) not divisible by patch (%d,%d,%d)", F, H, W, pt, ph, pw);
    WAN_REQUIRE(ldt >= (int64_t)Cin * pt * ph * pw, WAN_ERR_INVALID, "wan_patchify: ldt too small");
    WAN_REQUIRE(in_dtype == 0 || in_dtype == 1, WAN_ERR_INVALID, "wan_patchify: in_dtype=%d", in_dtype);
    const int64_t total = (int64_t)Cin * F * H * W;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == 0)
        hipLaunchKernelGGL(patchify_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)latent,
                           (bf16_t*)tokens_bf16, ldt, Cin, F, H, W, pt, ph, pw);
    else
        hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)latent,
                           (bf16_t*)tokens_bf16, ldt, Cin, F, H, W, pt, ph, pw);
    WAN_CHECK_LAUNCH("wan_patchify");
    return WAN_OK;
}

extern "C" wan_status_t wan_unpatchify(const float* tokens, int64_t ldt, void* out, int out_dtype,
                                       int Cout, int F, int Hp, int Wp, int pt, int ph, int pw, int zero_frames,
                                       void* stream) {
    WAN_REQUIRE(tokens && out, WAN_ERR_INVALID, "wan_unpatchify: null tensor");
    WAN_REQUIRE(Cout > 0 && F > 0 && Hp > 0 && Wp > 0 && pt > 0 && ph > 0 && pw > 0, WAN_ERR_INVALID, "wan_unpatchify: bad shape");
    WAN_REQUIRE(ldt >= (int64_t)Cout * pt * ph * pw, WAN_ERR_INVALID, "wan_unpatchify: ldt too small");
    WAN_REQUIRE(out_dtype == 0 || out_dtype == 1, WAN_ERR_INVALID, "wan_unpatchify: out_dtype=%d", out_dtype);
    WAN_REQUIRE(zero_frames >= 0, WAN_ERR_INVALID, "wan_unpatchify: zero_frames=%d", zero_frames);
    const int64_t total = (int64_t)Cout * F * pt * Hp * ph * Wp * pw;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == 0)
        hipLaunchKernelGGL(unpatchify_kernel<float>, dim3(blocks), dim3(256), 0, s, tokens, ldt, (float*)out,
                           Cout, F, Hp, Wp, pt, ph, pw, zero_frames);
    else
        hipLaunchKernelGGL(unpatchify_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, tokens, ldt, (bf16_t*)out,
                           Cout, F, Hp, Wp, pt, ph, pw, zero_frames);
    WAN_CHECK_LAUNCH("wan_unpatchify");
    return WAN_OK;
}

// ---------------------------------------------------------------------------------------------
// UniPC updates are linear combinations of <= 4 latent-sized tensors (host computes the float64
// coefficients): out = sum_i c_i * x_i in ONE pass, fp32 accumulate, instead of ~10 elementwise
// launches in the tensors' own dtype (fm_solvers_unipc.py:318-320, 458-470, 600-612).
// ---------------------------------------------------------------------------------------------
namespace {
template <typename T>
__global__ __launch_bounds__(256) void lincomb_kernel(T* __restrict__ out, const T* __restrict__ x0, const T* __restrict__ x1,
                                                      const T* __restrict__ x2, const T* __restrict__ x3, float c0, float c1,
                                                      float c2, float c3, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float acc = c0 * (float)x0[i];
        if (x1) acc += c1 * (float)x1[i];
        if (x2) acc += c2 * (float)x2[i];
        if (x3) acc += c3 * (float)x3[i];
        out[i] = (T)acc;
    }
}
}  // namespace

extern "C" wan_status_t wan_lincomb(void* out, int dtype, const void* x0, const void* x1, const void* x2, const void* x3,
                                    float c0, float c1, float c2, float c3, int64_t n, void* stream) {
    WAN_REQUIRE(out && x0, WAN_ERR_INVALID, "wan_lincomb: null tensor");
    WAN_REQUIRE(dtype == 0 || dtype == 1, WAN_ERR_INVALID, "wan_lincomb: dtype=%d (0 fp32, 1 bf16)", dtype);
    WAN_REQUIRE(n >= 0, WAN_ERR_INVALID, "wan_lincomb: n=%lld", (long long)n);
    if (n == 0) return WAN_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL(lincomb_kernel<float>, dim3(blocks), dim3(256), 0, s, (float*)out, (const float*)x0, (const float*)x1,
                           (const float*)x2, (const float*)x3, c0, c1, c2, c3, n);
    else
        hipLaunchKernelGGL(lincomb_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (bf16_t*)out, (const bf16_t*)x0,
                           (const bf16_t*)x1, (const bf16_t*)x2, (const bf16_t*)x3, c0, c1, c2, c3, n);
    WAN_CHECK_LAUNCH("wan_lincomb");
    return WAN_OK;
}

// ---------------------------------------------------------------------------------------------
// Ulysses (sequence-parallel) wire layouts, replacing videox_fun/dist/wan_xfuser.py:68-111 + yunchang's all-to-all packing.
// One rank holds T local tokens of B samples; P ranks; Cl = C / P channels (= H / P heads) per rank after the exchange.
//   token-major wire   [P][T][B][Cl]:  q / k on the way out (slab d goes to rank d; written directly by wan_rmsnorm_rope's
//                      `out` arguments, or by wan_sp_pack_heads) -- on arrival slab s holds the tokens of rank s, i.e. the
//                      buffer reads as [P*T][B][Cl] = all tokens, my heads, with UNIFORM strides (row B*Cl, sample Cl), so
//                      wan_attention_fwd consumes it in place; the attention output is written in the same form and is
//                      the send buffer of the inverse exchange as it stands.
//   channel-major wire [P][Cl][B][T]:  V^T on the way out = what the V-projection's transposed epilogue writes with
//                      ldo = B*T (no packing); on arrival wan_sp_unpack_vt lays it out as [B][Cl][ldvt] (column s*T + t).
// The collective itself stays an RCCL all_to_all_single issued by the host (videocof_amd/dist.py).
// All three kernels move 16-byte chunks: Cl % 8 == 0 and T % 8 == 0.
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void sp_pack_heads_kernel(const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ out,
                                                            int P, int T, int B, int Cl, bool unpack, int split) {
    // one 16-byte chunk per thread; grid-stride over B*T*P*Cl/8 chunks, local layout index fastest along channels
    const int cpr = Cl >> 3;                                  // chunks per (token, slab)
    const int64_t total = (int64_t)B * T * P * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int c8 = (int)(r % cpr); r /= cpr;
        const int p = (int)(r % P); r /= P;
        const int t = (int)(r % T);
        const int b = (int)(r / T);
        const u32x4* loc = reinterpret_cast<const u32x4*>(x + ((int64_t)b * T + t) * ldx + (int64_t)p * Cl) + c8;
        const int64_t cell = ((int64_t)p * T + t) * B + b;
        const int c = c8 << 3;
        // split > 0: channels [0, split) of every slab form head group 0, the rest group 1 -- two complete wire buffers, one behind the other
        const int64_t at = split <= 0 ? cell * Cl + c : c < split ? cell * split + c : (int64_t)P * T * B * split + cell * (Cl - split) + (c - split);
        u32x4* wire = reinterpret_cast<u32x4*>(out + at);
        if (unpack) *const_cast<u32x4*>(loc) = *wire;        // `x` is the destination [B][T][ldx], `out` the wire buffer
        else *wire = *loc;
    }
}

__global__ __launch_bounds__(256) void sp_unpack_vt_kernel(const bf16_t* __restrict__ wire, bf16_t* __restrict__ vt, int64_t ldvt,
                                                           int P, int B, int Cl, int T) {
    const int tpr = T >> 3;                                   // chunks per (source, channel, sample)
    const int64_t total = (int64_t)P * Cl * B * tpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int t8 = (int)(r % tpr); r /= tpr;
        const int b = (int)(r % B); r /= B;
        const int c = (int)(r % Cl);
        const int s = (int)(r / Cl);
        const u32x4 v = reinterpret_cast<const u32x4*>(wire + (((int64_t)s * Cl + c) * B + b) * T)[t8];
        reinterpret_cast<u32x4*>(vt + ((int64_t)b * Cl + c) * ldvt + (int64_t)s * T)[t8] = v;
    }
}
}  // namespace

static wan_status_t sp_check(const char* what, int P, int T, int B, int Cl) {
    WAN_REQUIRE(P > 0 && T > 0 && B > 0 && Cl > 0 && Cl % 8 == 0, WAN_ERR_INVALID, "%s: P=%d T=%d B=%d Cl=%d (Cl must be a multiple of 8)",
                what, P, T, B, Cl);
    return WAN_OK;
}

static wan_status_t sp_heads(const char* what, const void* x, int64_t ldx, void* wire, int P, int T, int B, int Cl, int split, bool unpack,
                             void* stream) {
    WAN_REQUIRE(x && wire, WAN_ERR_INVALID, "%s: null tensor", what);
    if (wan_status_t st = sp_check(what, P, T, B, Cl)) return st;
    WAN_REQUIRE(ldx >= (int64_t)P * Cl && ldx % 8 == 0, WAN_ERR_INVALID, "%s: ldx=%lld", what, (long long)ldx);
    WAN_REQUIRE(split >= 0 && split % 8 == 0 && split < Cl, WAN_ERR_INVALID, "%s: split=%d must be a multiple of 8 below Cl=%d (0 = one group)",
                what, split, Cl);
    const int64_t total = (int64_t)B * T * P * (Cl >> 3);
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(sp_pack_heads_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)wire, P, T, B, Cl,
                       unpack, split);
    WAN_CHECK_LAUNCH(what);
    return WAN_OK;
}

extern "C" wan_status_t wan_sp_pack_heads(const void* x, int64_t ldx, void* wire, int P, int T, int B, int Cl, void* stream) {
    return sp_heads("wan_sp_pack_heads", x, ldx, wire, P, T, B, Cl, 0, false, stream);
}

extern "C" wan_status_t wan_sp_unpack_heads(const void* wire, void* x, int64_t ldx, int P, int T, int B, int Cl, void* stream) {
    return sp_heads("wan_sp_unpack_heads", x, ldx, const_cast<void*>(wire), P, T, B, Cl, 0, true, stream);
}

extern "C" wan_status_t wan_sp_pack_heads_split(const void* x, int64_t ldx, void* wire, int P, int T, int B, int Cl, int split, void* stream) {
    return sp_heads("wan_sp_pack_heads_split", x, ldx, wire, P, T, B, Cl, split, false, stream);
}

extern "C" wan_status_t wan_sp_unpack_heads_split(const void* wire, void* x, int64_t ldx, int P, int T, int B, int Cl, int split,
                                                  void* stream) {
    return sp_heads("wan_sp_unpack_heads_split", x, ldx, const_cast<void*>(wire), P, T, B, Cl, split, true, stream);
}

extern "C" wan_status_t wan_sp_unpack_vt(const void* wire, void* vt, int64_t ldvt, int P, int B, int Cl, int T, void* stream) {
    WAN_REQUIRE(vt && wire, WAN_ERR_INVALID, "wan_sp_unpack_vt: null tensor");
    if (wan_status_t st = sp_check("wan_sp_unpack_vt", P, T, B, Cl)) return st;
    WAN_REQUIRE(T % 8 == 0 && ldvt >= (int64_t)P * T && ldvt % 8 == 0, WAN_ERR_INVALID, "wan_sp_unpack_vt: T=%d ldvt=%lld (T %% 8 == 0, ldvt >= P*T)",
                T, (long long)ldvt);
    const int64_t total = (int64_t)P * Cl * B * (T >> 3);
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(sp_unpack_vt_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)wire, (bf16_t*)vt, ldvt, P, B, Cl, T);
    WAN_CHECK_LAUNCH("wan_sp_unpack_vt");
    return WAN_OK;
}
