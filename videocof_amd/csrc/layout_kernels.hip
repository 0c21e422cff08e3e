// Layout glue at the two ends of the DiT: patchify (im2col of the (1,2,2) Conv3d) and unpatchify.
// Pure data movement (HBM-bound, a few MB per forward).
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
