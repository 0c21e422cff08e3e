// Row kernels of the umT5 text encoder (videox_fun/models/wan_text_encoder.py), the step before the
// denoising path.  All are HBM-bound, one 256-thread workgroup per row, 16-byte accesses.
//
//   wan_embedding_rows   : nn.Embedding gather, bf16 table -> fp32 residual stream          (:286-287)
//   wan_rmsnorm_rows     : T5LayerNorm  x * rsqrt(mean(x^2) + eps) * w   (no mean, no bias)   (:48-60)
//   wan_t5_softmax_bias  : softmax_j(S[h,i,j] + table[bucket(j - i), h]) with key masking      (:93-105, 226-260)
//   wan_mul_bf16         : fc1(x) * gelu(gate(x)) of the gated-GELU feed-forward               (:125-126)
#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

__global__ __launch_bounds__(kThreads) void embedding_rows_kernel(const int64_t* __restrict__ ids,
                                                                  const bf16_t* __restrict__ table, int64_t vocab,
                                                                  float* __restrict__ out, int dim) {
    int64_t id = ids[blockIdx.x];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);     // memory safety only; the host validates the range
    const u32x4* src = reinterpret_cast<const u32x4*>(table + id * dim);
    float4* dst = reinterpret_cast<float4*>(out + (int64_t)blockIdx.x * dim);
    for (int c = threadIdx.x; c < (dim >> 3); c += kThreads) {
        const u32x4 v = src[c];
        dst[2 * c] = make_float4(bf16lo_to_f32(v[0]), bf16hi_to_f32(v[0]), bf16lo_to_f32(v[1]), bf16hi_to_f32(v[1]));
        dst[2 * c + 1] = make_float4(bf16lo_to_f32(v[2]), bf16hi_to_f32(v[2]), bf16lo_to_f32(v[3]), bf16hi_to_f32(v[3]));
    }
}

template <int NV, bool OUT_F32>   // float4 chunks per thread
__global__ __launch_bounds__(kThreads) void rmsnorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                void* __restrict__ out, int dim, float eps) {
    __shared__ float red[kWaves];
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)blockIdx.x * dim);
    const int nchunk = dim >> 2;
    float4 v[NV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * kThreads;
        if (idx < nchunk) {
            v[i] = xr[idx];
            ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
    }
    const float rstd = rsqrtf(block_sum<kWaves>(ss, red) / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * kThreads;
        if (idx < nchunk) {
            const float4 a = reinterpret_cast<const float4*>(w)[idx];
            const float y0 = v[i].x * rstd * a.x, y1 = v[i].y * rstd * a.y, y2 = v[i].z * rstd * a.z, y3 = v[i].w * rstd * a.w;
            if constexpr (OUT_F32) {
                reinterpret_cast<float4*>((float*)out + (int64_t)blockIdx.x * dim)[idx] = make_float4(y0, y1, y2, y3);
            } else {
                u32x2 o = {pack_bf16x2(y0, y1), pack_bf16x2(y2, y3)};
                reinterpret_cast<u32x2*>((bf16_t*)out + (int64_t)blockIdx.x * dim)[idx] = o;
            }
        }
    }
}

// one workgroup per (head, query) row.  bias[h][i][j] = table[lut[j - i + Lq - 1] * H + h]; keys >= k_len masked.
__global__ __launch_bounds__(kThreads) void t5_softmax_bias_kernel(const float* __restrict__ s, int64_t lds_,
                                                                   const float* __restrict__ table,
                                                                   const int* __restrict__ lut, bf16_t* __restrict__ p,
                                                                   int64_t ldp, int H, int Lq, int k_len, int npad) {
    __shared__ float red[kWaves];
    const int h = blockIdx.x / Lq, i = blockIdx.x - h * Lq;
    const float* sr = s + (int64_t)blockIdx.x * lds_;
    bf16_t* pr = p + (int64_t)blockIdx.x * ldp;
    const int* lrow = lut + (Lq - 1 - i);
    // k_len <= 2048: at most 8 scores per thread, kept in registers
    float v[8];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int j = threadIdx.x + t * kThreads;
        v[t] = j < k_len ? sr[j] + table[lrow[j] * H + h] : -INFINITY;
        mx = fmaxf(mx, v[t]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        v[t] = __expf(v[t] - mx);       // exp(-inf) = 0 for masked / out-of-range keys
        sum += v[t];
    }
    const float inv = 1.f / block_sum<kWaves>(sum, red);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int j = threadIdx.x + t * kThreads;
        if (j < npad) pr[j] = (bf16_t)(v[t] * inv);
    }
}

__global__ __launch_bounds__(kThreads) void mul_bf16_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                                            u32x4* __restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n8; i += (int64_t)gridDim.x * kThreads) {
        const u32x4 x = a[i], y = b[i];
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = pack_bf16x2(bf16lo_to_f32(x[j]) * bf16lo_to_f32(y[j]), bf16hi_to_f32(x[j]) * bf16hi_to_f32(y[j]));
        out[i] = o;
    }
}

}  // namespace

extern "C" wan_status_t wan_embedding_rows(const int64_t* ids, const void* table_bf16, int64_t vocab, float* out,
                                           int64_t rows, int dim, void* stream) {
    WAN_REQUIRE(ids && table_bf16 && out, WAN_ERR_INVALID, "wan_embedding_rows: null tensor");
    WAN_REQUIRE(vocab > 0 && rows >= 0 && dim > 0 && dim % 8 == 0, WAN_ERR_INVALID,
                "wan_embedding_rows: vocab=%lld rows=%lld dim=%d (dim must be a multiple of 8)", (long long)vocab,
                (long long)rows, dim);
    if (rows == 0) return WAN_OK;
    hipLaunchKernelGGL(embedding_rows_kernel, dim3((unsigned)rows), dim3(kThreads), 0, (hipStream_t)stream, ids,
                       (const bf16_t*)table_bf16, vocab, out, dim);
    WAN_CHECK_LAUNCH("wan_embedding_rows");
    return WAN_OK;
}

extern "C" wan_status_t wan_rmsnorm_rows(const float* x, const float* w, void* out, int out_dtype, int64_t rows, int dim,
                                         float eps, void* stream) {
    WAN_REQUIRE(x && w && out, WAN_ERR_INVALID, "wan_rmsnorm_rows: null tensor");
    WAN_REQUIRE(dim > 0 && dim % 4 == 0 && rows >= 0, WAN_ERR_INVALID, "wan_rmsnorm_rows: rows=%lld dim=%d", (long long)rows, dim);
    WAN_REQUIRE(dim <= 8192, WAN_ERR_UNSUPPORTED, "wan_rmsnorm_rows: dim=%d > 8192", dim);
    WAN_REQUIRE(out_dtype == 0 || out_dtype == 1, WAN_ERR_INVALID, "wan_rmsnorm_rows: out_dtype=%d (0 fp32, 1 bf16)", out_dtype);
    if (rows == 0) return WAN_OK;
    hipStream_t s = (hipStream_t)stream;
    const int nv = (dim / 4 + kThreads - 1) / kThreads;
    dim3 grid((unsigned)rows), block(kThreads);
#define RN_CASE(N)                                                                                              \
    case N:                                                                                                     \
        if (out_dtype == 0) hipLaunchKernelGGL((rmsnorm_rows_kernel<N, true>), grid, block, 0, s, x, w, out, dim, eps); \
        else hipLaunchKernelGGL((rmsnorm_rows_kernel<N, false>), grid, block, 0, s, x, w, out, dim, eps);       \
        break;
    switch (nv) { RN_CASE(1) RN_CASE(2) RN_CASE(3) RN_CASE(4) RN_CASE(5) RN_CASE(6) RN_CASE(7) RN_CASE(8) }
#undef RN_CASE
    WAN_CHECK_LAUNCH("wan_rmsnorm_rows");
    return WAN_OK;
}

extern "C" wan_status_t wan_t5_softmax_bias(const float* scores, int64_t lds, const float* bucket_table,
                                            const int* bucket_lut, void* probs_bf16, int64_t ldp, int num_heads,
                                            int Lq, int Lk, int k_len, int npad, void* stream) {
    WAN_REQUIRE(scores && bucket_table && bucket_lut && probs_bf16, WAN_ERR_INVALID, "wan_t5_softmax_bias: null tensor");
    WAN_REQUIRE(num_heads > 0 && Lq > 0 && Lk > 0 && k_len > 0 && k_len <= Lk && npad >= Lk && lds >= Lk && ldp >= npad,
                WAN_ERR_INVALID, "wan_t5_softmax_bias: H=%d Lq=%d Lk=%d k_len=%d npad=%d", num_heads, Lq, Lk, k_len, npad);
    WAN_REQUIRE(npad <= 8 * kThreads, WAN_ERR_UNSUPPORTED, "wan_t5_softmax_bias: npad=%d > %d", npad, 8 * kThreads);
    WAN_REQUIRE(Lq == Lk, WAN_ERR_UNSUPPORTED, "wan_t5_softmax_bias: only self-attention (Lq == Lk) is built");
    hipLaunchKernelGGL(t5_softmax_bias_kernel, dim3((unsigned)(num_heads * Lq)), dim3(kThreads), 0, (hipStream_t)stream,
                       scores, lds, bucket_table, bucket_lut, (bf16_t*)probs_bf16, ldp, num_heads, Lq, k_len, npad);
    WAN_CHECK_LAUNCH("wan_t5_softmax_bias");
    return WAN_OK;
}

extern "C" wan_status_t wan_mul_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
    WAN_REQUIRE(a && b && out, WAN_ERR_INVALID, "wan_mul_bf16: null tensor");
    WAN_REQUIRE(n >= 0 && n % 8 == 0, WAN_ERR_INVALID, "wan_mul_bf16: n=%lld must be a multiple of 8", (long long)n);
    if (n == 0) return WAN_OK;
    const int64_t n8 = n / 8;
    const unsigned blocks = (unsigned)((n8 + kThreads - 1) / kThreads < 65536 ? (n8 + kThreads - 1) / kThreads : 65536);
    hipLaunchKernelGGL(mul_bf16_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream,
                       (const u32x4*)a, (const u32x4*)b, (u32x4*)out, n8);
    WAN_CHECK_LAUNCH("wan_mul_bf16");
    return WAN_OK;
}
