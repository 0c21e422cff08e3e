// HBM-bound row kernels of the DiT block:
//   wan_ln_modulate  : LayerNorm (fp32 in) -> * (add_one + scale) + shift -> bf16
//   wan_rmsnorm_rope : RMSNorm over the full channel dim + 3-axis RoPE (CoF positions), in place on bf16
// One 256-thread workgroup per token row; the row lives in registers between the
// reduction and the write, so each element is read once and written once
// (algorithmic bytes: 6*C per row for LN-modulate, 4*C per row per tensor for RMSNorm+RoPE).
#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

// ------------------------------------------------------------------ LN + modulate
// FP8 = true (wan_ln_modulate_fp8): the row is written as OCP e4m3 bytes with ONE scale per token row,
// q[c] = cvt(y[c] / s), s = max_c |y[c]| / 448 -- the activation operand of wan_gemm_fp8.  The row is in registers anyway, so the
// quantisation costs one more block reduction (max) and no memory pass.
template <int NV, bool FP8 = false>   // float4 per thread
__global__ __launch_bounds__(kThreads) void ln_modulate_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    float add_one, bf16_t* __restrict__ out, int dim, int64_t rows_per_batch, float eps, float* __restrict__ row_scale = nullptr) {
    __shared__ float red[kWaves];
    const int64_t row = blockIdx.x;
    const int64_t b = row / rows_per_batch;
    const int nvec = dim >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + row * (int64_t)dim);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * kThreads;
        if (idx < nvec) {
            v[i] = xr[idx];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float mean = block_sum<kWaves>(s, red) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * kThreads;
        if (idx < nvec) {
            const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + bb * bb) + (c * c + d * d);
        }
    }
    const float rstd = rsqrtf(block_sum<kWaves>(q, red) / (float)dim + eps);
    const float4* sc = scale ? reinterpret_cast<const float4*>(scale + b * dim) : nullptr;
    const float4* sh = shift ? reinterpret_cast<const float4*>(shift + b * dim) : nullptr;
    u32x2* orow = reinterpret_cast<u32x2*>(out + row * (int64_t)dim);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * kThreads;
        if (idx < nvec) {
            float4 a = make_float4(add_one, add_one, add_one, add_one);
            float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sc) { const float4 t = sc[idx]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
            else if (add_one == 0.f) { a = make_float4(1.f, 1.f, 1.f, 1.f); }
            if (sh) c = sh[idx];
            const float y0 = (v[i].x - mean) * rstd * a.x + c.x;
            const float y1 = (v[i].y - mean) * rstd * a.y + c.y;
            const float y2 = (v[i].z - mean) * rstd * a.z + c.z;
            const float y3 = (v[i].w - mean) * rstd * a.w + c.w;
            if constexpr (FP8) {
                v[i] = make_float4(y0, y1, y2, y3);
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(y0), fabsf(y1))), fmaxf(fabsf(y2), fabsf(y3)));
            } else {
                u32x2 o = {pack_bf16x2(y0, y1), pack_bf16x2(y2, y3)};
                orow[idx] = o;
            }
        }
    }
    if constexpr (FP8) {
        amax = fmaxf(block_max<kWaves>(amax, red), 1e-12f);
        const float s = amax * (1.0f / 448.0f), inv = 448.0f / amax;
        if (threadIdx.x == 0) row_scale[row] = s;
        unsigned int* qrow = reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(out) + row * (int64_t)dim);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * kThreads;
            if (idx < nvec) qrow[idx] = pack_fp8x4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
        }
    }
}


// ------------------------------------------------------------------ R rows per workgroup (round 4)
// What bounds the row kernels is not the row traffic but (a) the per-column parameters -- the modulation vectors (2 x 4C bytes) or the
// RMSNorm gains (4C bytes) came from L2 once PER ROW, against 4C / 2C bytes of HBM reads for the row itself: LN-modulate ran 4.9 TB/s
// with its vectors and 5.95 TB/s, the rate of a plain fp32 -> bf16 cast, without them (tools/probe/rows_probe.py) -- and (b) the bytes
// in flight per CU (a persistent form that kept the parameters in registers lost more to its lower occupancy than it saved,
// profiles/r04/row_kernels_persistent_variant_ab.log).  Here a workgroup owns R consecutive rows: a thread reads its columns'
// parameters ONCE for the R rows, the R row sums share one pair of barriers, and a resident workgroup keeps R rows in flight.
// R = 2 is the default (profiles/r04/row_kernels_rows_per_workgroup_ab.log: LN-modulate 4.85 -> 5.7 TB/s, 97 % of the cast rate in
// process; RMSNorm+RoPE 4.7 -> 5.0-5.3; R = 4 costs LN-modulate its occupancy: 192 registers).  Per-row arithmetic and summation order
// are those of the one-row kernels: results are bitwise equal.
template <int NW, int R>
__device__ __forceinline__ void block_sum_n(float (&v)[R], float (*red)[NW]) {
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = wave_sum(v[r]);
    const int wid = threadIdx.x >> 6;
    __syncthreads();   // protect `red` from the previous use
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) red[r][wid] = v[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) t += red[r][i];
        v[r] = t;
    }
}

template <int NV, int R>   // float4 per thread and row
__global__ __launch_bounds__(kThreads) void ln_modulate_rows_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    float add_one, bf16_t* __restrict__ out, int dim, int64_t rows, int64_t rows_per_batch, float eps) {
    __shared__ float red[R][kWaves];
    const int64_t row0 = (int64_t)blockIdx.x * R;
    const int nvec = dim >> 2;
    float4 v[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float4* xr = reinterpret_cast<const float4*>(x + (row0 + r) * (int64_t)dim);
        const bool live = row0 + r < rows;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * kThreads;
            v[r][i] = (live && idx < nvec) ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (threadIdx.x + i * kThreads < nvec) s[r] += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
    }
    block_sum_n<kWaves, R>(s, red);
    float mean[R], q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mean[r] = s[r] / (float)dim;
        q[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (threadIdx.x + i * kThreads < nvec) {
                const float a = v[r][i].x - mean[r], bb = v[r][i].y - mean[r], c = v[r][i].z - mean[r], d = v[r][i].w - mean[r];
                q[r] += (a * a + bb * bb) + (c * c + d * d);
            }
        }
    }
    block_sum_n<kWaves, R>(q, red);
    const int64_t last = (row0 + R < rows ? row0 + R : rows) - 1;
    const int64_t b0 = row0 / rows_per_batch;
    const bool one_sample = last / rows_per_batch == b0;          // (a group that straddles two samples fetches per row)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * kThreads;
        if (idx < nvec) {
            float4 a, c;
            auto fetch = [&](int64_t b) {
                a = make_float4(add_one, add_one, add_one, add_one);
                c = make_float4(0.f, 0.f, 0.f, 0.f);
                if (scale) { const float4 t = reinterpret_cast<const float4*>(scale + b * dim)[idx]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
                else if (add_one == 0.f) { a = make_float4(1.f, 1.f, 1.f, 1.f); }
                if (shift) c = reinterpret_cast<const float4*>(shift + b * dim)[idx];
            };
            fetch(b0);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (row0 + r < rows) {
                    if (!one_sample) fetch((row0 + r) / rows_per_batch);
                    const float rstd = rsqrtf(q[r] / (float)dim + eps);
                    const float y0 = (v[r][i].x - mean[r]) * rstd * a.x + c.x;
                    const float y1 = (v[r][i].y - mean[r]) * rstd * a.y + c.y;
                    const float y2 = (v[r][i].z - mean[r]) * rstd * a.z + c.z;
                    const float y3 = (v[r][i].w - mean[r]) * rstd * a.w + c.w;
                    u32x2 o = {pack_bf16x2(y0, y1), pack_bf16x2(y2, y3)};
                    reinterpret_cast<u32x2*>(out + (row0 + r) * (int64_t)dim)[idx] = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ RMSNorm + RoPE
struct RopeDev {
    int F, Hp, Wp, mode, f_src, ground_end, max_pos;
    int ct, ch;                 // complex pairs on the t and h axes (rest = w)
    int64_t token_offset, rows_per_batch;
};

template <int NV>   // 16-byte chunks (8 bf16) per thread
__global__ __launch_bounds__(kThreads) void rmsnorm_rope_kernel(
    bf16_t* __restrict__ x0, const float* __restrict__ w0, bf16_t* __restrict__ x1,
    const float* __restrict__ w1, int64_t ld, int dim, int head_dim, float eps,
    const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, RopeDev rp, float x0_scale,
    bf16_t* __restrict__ out0, bf16_t* __restrict__ out1, int out_slabs, int out_batch, float x1_scale, int out_fp8, int out_split) {
    __shared__ float red[kWaves];
    __shared__ __attribute__((aligned(16))) float2 cs[128];   // (cos, sin) of this token's head_dim/2 pairs
    const int64_t row = blockIdx.x;
    bf16_t* xb = blockIdx.y == 0 ? x0 : x1;
    bf16_t* ob = blockIdx.y == 0 ? out0 : out1;               // nullptr: in place
    const float* w = blockIdx.y == 0 ? w0 : w1;
    u32x4* xr = reinterpret_cast<u32x4*>(xb + row * ld);
    const int nchunk = dim >> 3;
    const int half = head_dim >> 1;

    bool rotate = rope_cos != nullptr;
    if (rotate) {
        const int64_t tok = rp.token_offset + (row % rp.rows_per_batch);
        const int64_t hw = (int64_t)rp.Hp * rp.Wp;
        rotate = tok < (int64_t)rp.F * hw;                 // rows past the grid pass through (:202)
        if (rotate && threadIdx.x < half) {
            const int f = (int)(tok / hw);
            const int rem = (int)(tok - (int64_t)f * hw);
            const int hh = rem / rp.Wp, ww = rem - hh * rp.Wp;
            int pt = f;                                     // default 0..F-1 (:191)
            if (rp.mode == 1) pt = f < rp.f_src ? f : f - rp.f_src;                    // paired (:183-188)
            else if (rp.mode == 2)                                                     // CoF (:160-179)
                pt = f < rp.f_src ? f + 1 : (f < rp.ground_end ? 0 : f - rp.ground_end + 1);
            const int p = threadIdx.x;
            int pos = p < rp.ct ? pt : (p < rp.ct + rp.ch ? hh : ww);
            pos = pos < rp.max_pos ? pos : rp.max_pos - 1;
            cs[p] = make_float2(rope_cos[(int64_t)pos * half + p], rope_sin[(int64_t)pos * half + p]);
        }
    }

    u32x4 v[NV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * kThreads;
        if (idx < nchunk) {
            v[i] = xr[idx];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = bf16lo_to_f32(v[i][j]), b = bf16hi_to_f32(v[i][j]);
                ss += a * a + b * b;
            }
        }
    }
    const float rstd = rsqrtf(block_sum<kWaves>(ss, red) / (float)dim + eps)    // also orders cs[] writes
                       * (blockIdx.y == 0 ? x0_scale : x1_scale);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * kThreads;
        if (idx < nchunk) {
            const float4 wa = reinterpret_cast<const float4*>(w)[idx * 2];
            const float4 wb = reinterpret_cast<const float4*>(w)[idx * 2 + 1];
            const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
            const int p0 = ((idx << 3) % head_dim) >> 1;    // first complex pair of this chunk within its head
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = bf16lo_to_f32(v[i][j]) * rstd * wv[2 * j];
                float b = bf16hi_to_f32(v[i][j]) * rstd * wv[2 * j + 1];
                if (rotate) {
                    const float2 c = cs[p0 + j];
                    const float ra = a * c.x - b * c.y;
                    const float rb = a * c.y + b * c.x;
                    a = ra; b = rb;
                }
                o[j] = pack_bf16x2(a, b);
            }
            if (ob == nullptr) {
                xr[idx] = o;
            } else if (out_fp8) {
                // OCP e4m3 copy of the (scaled) row, dense [rows][dim] bytes: the QK^T operands of the fp8 attention variant.
                // Re-derived from the bf16-rounded values so that it is a quantisation of exactly what the bf16 path would see.
                u32x2 q8 = {pack_fp8x4(bf16lo_to_f32(o[0]), bf16hi_to_f32(o[0]), bf16lo_to_f32(o[1]), bf16hi_to_f32(o[1])),
                            pack_fp8x4(bf16lo_to_f32(o[2]), bf16hi_to_f32(o[2]), bf16lo_to_f32(o[3]), bf16hi_to_f32(o[3]))};
                *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(ob) + row * (int64_t)dim + (idx << 3)) = q8;
            } else {
                // Ulysses send layout [slab][t][b][Cl] (layout_kernels.hip): the row's channels are split into `out_slabs` head
                // groups, one per destination rank, so the all-to-all sends this buffer as it stands
                const int Cl = dim / out_slabs, c = idx << 3;
                const int slab = c / Cl, cl = c - slab * Cl;
                const int64_t bi = row / rp.rows_per_batch, t = row - bi * rp.rows_per_batch;
                const int64_t cell = ((int64_t)slab * rp.rows_per_batch + t) * out_batch + bi;       // (slab, token, sample)
                // out_split > 0: two head groups, each a complete wire buffer of its own, one behind the other (layout_kernels.hip)
                const int64_t at = out_split <= 0 ? cell * Cl + cl
                                 : cl < out_split ? cell * out_split + cl
                                                  : (int64_t)out_slabs * rp.rows_per_batch * out_batch * out_split + cell * (Cl - out_split) + (cl - out_split);
                *reinterpret_cast<u32x4*>(ob + at) = o;
            }
        }
    }
}


template <int NV, int R>   // 16-byte chunks (8 bf16) per thread and row; see ln_modulate_rows_kernel
__global__ __launch_bounds__(kThreads) void rmsnorm_rope_rows_kernel(
    bf16_t* __restrict__ x0, const float* __restrict__ w0, bf16_t* __restrict__ x1,
    const float* __restrict__ w1, int64_t ld, int64_t rows, int dim, int head_dim, float eps,
    const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, RopeDev rp, float x0_scale,
    bf16_t* __restrict__ out0, bf16_t* __restrict__ out1, int out_slabs, int out_batch, float x1_scale, int out_fp8, int out_split) {
    __shared__ float red[R][kWaves];
    __shared__ __attribute__((aligned(16))) float2 cs[R][128];   // (cos, sin) of each row's head_dim/2 pairs
    const int64_t row0 = (int64_t)blockIdx.x * R;
    bf16_t* xb = blockIdx.y == 0 ? x0 : x1;
    bf16_t* ob = blockIdx.y == 0 ? out0 : out1;               // nullptr: in place
    const float* w = blockIdx.y == 0 ? w0 : w1;
    const int nchunk = dim >> 3;
    const int half = head_dim >> 1;

    u32x4 v[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const u32x4* xr = reinterpret_cast<const u32x4*>(xb + (row0 + r) * ld);
        const bool live = row0 + r < rows;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * kThreads;
            if (live && idx < nchunk) v[r][i] = xr[idx]; else v[r][i] = u32x4{0u, 0u, 0u, 0u};
        }
    }
    bool rotate[R];
    const int64_t hw = (int64_t)rp.Hp * rp.Wp;
#pragma unroll
    for (int r = 0; r < R; ++r)
        rotate[r] = rope_cos != nullptr && rp.token_offset + ((row0 + r) % rp.rows_per_batch) < (int64_t)rp.F * hw;   // rows past the grid pass through (:202)
    if (rope_cos != nullptr) {
        for (int e = threadIdx.x; e < R * half; e += kThreads) {
            const int r = e / half, p = e - r * half;
            const int64_t tok = rp.token_offset + ((row0 + r) % rp.rows_per_batch);
            if (row0 + r < rows && tok < (int64_t)rp.F * hw) {
                const int f = (int)(tok / hw);
                const int rem = (int)(tok - (int64_t)f * hw);
                const int hh = rem / rp.Wp, ww = rem - hh * rp.Wp;
                int pt = f;                                     // default 0..F-1 (:191)
                if (rp.mode == 1) pt = f < rp.f_src ? f : f - rp.f_src;                    // paired (:183-188)
                else if (rp.mode == 2)                                                     // CoF (:160-179)
                    pt = f < rp.f_src ? f + 1 : (f < rp.ground_end ? 0 : f - rp.ground_end + 1);
                int pos = p < rp.ct ? pt : (p < rp.ct + rp.ch ? hh : ww);
                pos = pos < rp.max_pos ? pos : rp.max_pos - 1;
                cs[r][p] = make_float2(rope_cos[(int64_t)pos * half + p], rope_sin[(int64_t)pos * half + p]);
            }
        }
    }
    float ss[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        ss[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (threadIdx.x + i * kThreads < nchunk) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = bf16lo_to_f32(v[r][i][j]), b = bf16hi_to_f32(v[r][i][j]);
                    ss[r] += a * a + b * b;
                }
            }
        }
    }
    block_sum_n<kWaves, R>(ss, red);                       // also orders the cs[][] writes
    const float post = blockIdx.y == 0 ? x0_scale : x1_scale;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * kThreads;
        if (idx < nchunk) {
            const float4 wa = reinterpret_cast<const float4*>(w)[idx * 2];        // once for the R rows
            const float4 wb = reinterpret_cast<const float4*>(w)[idx * 2 + 1];
            const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
            const int p0 = ((idx << 3) % head_dim) >> 1;    // first complex pair of this chunk within its head
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t row = row0 + r;
                if (row >= rows) continue;
                const float rstd = rsqrtf(ss[r] / (float)dim + eps) * post;
                u32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = bf16lo_to_f32(v[r][i][j]) * rstd * wv[2 * j];
                    float b = bf16hi_to_f32(v[r][i][j]) * rstd * wv[2 * j + 1];
                    if (rotate[r]) {
                        const float2 c = cs[r][p0 + j];
                        const float ra = a * c.x - b * c.y;
                        const float rb = a * c.y + b * c.x;
                        a = ra; b = rb;
                    }
                    o[j] = pack_bf16x2(a, b);
                }
                if (ob == nullptr) {
                    reinterpret_cast<u32x4*>(xb + row * ld)[idx] = o;
                } else if (out_fp8) {
                    u32x2 q8 = {pack_fp8x4(bf16lo_to_f32(o[0]), bf16hi_to_f32(o[0]), bf16lo_to_f32(o[1]), bf16hi_to_f32(o[1])),
                                pack_fp8x4(bf16lo_to_f32(o[2]), bf16hi_to_f32(o[2]), bf16lo_to_f32(o[3]), bf16hi_to_f32(o[3]))};
                    *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(ob) + row * (int64_t)dim + (idx << 3)) = q8;
                } else {
                    const int Cl = dim / out_slabs, c = idx << 3;
                    const int slab = c / Cl, cl = c - slab * Cl;
                    const int64_t bi = row / rp.rows_per_batch, t = row - bi * rp.rows_per_batch;
                    const int64_t cell = ((int64_t)slab * rp.rows_per_batch + t) * out_batch + bi;       // (slab, token, sample)
                    const int64_t at = out_split <= 0 ? cell * Cl + cl
                                     : cl < out_split ? cell * out_split + cl
                                                      : (int64_t)out_slabs * rp.rows_per_batch * out_batch * out_split + cell * (Cl - out_split) + (cl - out_split);
                    *reinterpret_cast<u32x4*>(ob + at) = o;
                }
            }
        }
    }
}

}  // namespace

extern "C" wan_status_t wan_ln_modulate(const float* x, const float* scale, const float* shift, int add_one,
                                        void* out_bf16, int64_t rows, int dim, int64_t rows_per_batch,
                                        float eps, void* stream) {
    WAN_REQUIRE(x && out_bf16, WAN_ERR_INVALID, "wan_ln_modulate: null tensor");
    WAN_REQUIRE(dim > 0 && dim % 4 == 0, WAN_ERR_INVALID, "wan_ln_modulate: dim=%d must be a multiple of 4", dim);
    WAN_REQUIRE(dim <= 8192, WAN_ERR_UNSUPPORTED, "wan_ln_modulate: dim=%d > 8192", dim);
    WAN_REQUIRE(rows >= 0 && rows_per_batch > 0, WAN_ERR_INVALID, "wan_ln_modulate: rows=%lld rows_per_batch=%lld",
                (long long)rows, (long long)rows_per_batch);
    if (rows == 0) return WAN_OK;
    hipStream_t s = (hipStream_t)stream;
    const int nv = (dim / 4 + kThreads - 1) / kThreads;
    dim3 grid((unsigned)rows), block(kThreads);
    bf16_t* out = (bf16_t*)out_bf16;
    const float ao = add_one ? 1.f : 0.f;
    const int R = wan_tune(WAN_TUNE_ROW_GROUP);                   // rows per workgroup: 2 (default), 4, or 1 = the one-row kernel
    if ((R == 2 || R == 4) && nv <= 6) {
        dim3 ggrid((unsigned)((rows + R - 1) / R));
#define LG_CASE(N) case N: if (R == 4) hipLaunchKernelGGL((ln_modulate_rows_kernel<N, 4>), ggrid, block, 0, s, x, scale, shift, ao, out, dim, rows, rows_per_batch, eps); \
                           else hipLaunchKernelGGL((ln_modulate_rows_kernel<N, 2>), ggrid, block, 0, s, x, scale, shift, ao, out, dim, rows, rows_per_batch, eps); break;
        switch (nv) { LG_CASE(1) LG_CASE(2) LG_CASE(3) LG_CASE(4) LG_CASE(5) LG_CASE(6) }
#undef LG_CASE
        WAN_CHECK_LAUNCH("wan_ln_modulate");
        return WAN_OK;
    }
#define LN_CASE(N) case N: hipLaunchKernelGGL(ln_modulate_kernel<N>, grid, block, 0, s, x, scale, shift, ao, out, dim, rows_per_batch, eps); break;
    switch (nv) { LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8) }
#undef LN_CASE
    WAN_CHECK_LAUNCH("wan_ln_modulate");
    return WAN_OK;
}

extern "C" wan_status_t wan_ln_modulate_fp8(const float* x, const float* scale, const float* shift, int add_one,
                                            void* out_fp8, float* out_row_scale, int64_t rows, int dim,
                                            int64_t rows_per_batch, float eps, void* stream) {
    WAN_REQUIRE(x && out_fp8 && out_row_scale, WAN_ERR_INVALID, "wan_ln_modulate_fp8: null tensor");
    WAN_REQUIRE(dim > 0 && dim % 16 == 0, WAN_ERR_INVALID, "wan_ln_modulate_fp8: dim=%d must be a multiple of 16", dim);
    WAN_REQUIRE(dim <= 8192, WAN_ERR_UNSUPPORTED, "wan_ln_modulate_fp8: dim=%d > 8192", dim);
    WAN_REQUIRE(rows >= 0 && rows_per_batch > 0, WAN_ERR_INVALID, "wan_ln_modulate_fp8: rows=%lld rows_per_batch=%lld",
                (long long)rows, (long long)rows_per_batch);
    if (rows == 0) return WAN_OK;
    hipStream_t s = (hipStream_t)stream;
    const int nv = (dim / 4 + kThreads - 1) / kThreads;
    dim3 grid((unsigned)rows), block(kThreads);
    const float ao = add_one ? 1.f : 0.f;
#define LN_CASE(N) case N: hipLaunchKernelGGL((ln_modulate_kernel<N, true>), grid, block, 0, s, x, scale, shift, ao, (bf16_t*)out_fp8, dim, rows_per_batch, eps, out_row_scale); break;
    switch (nv) { LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8) }
#undef LN_CASE
    WAN_CHECK_LAUNCH("wan_ln_modulate_fp8");
    return WAN_OK;
}

namespace {
// bf16 [rows, cols] (row stride ldx) -> e4m3 [rows, cols] (row stride ldo bytes) + one scale per row: s = max|x| / 448.
// One workgroup per row; the row is read twice (max, then convert) -- the second read hits L2.
__global__ __launch_bounds__(kThreads) void quantize_rows_fp8_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                                     unsigned char* __restrict__ out, int64_t ldo,
                                                                     float* __restrict__ row_scale, int cols) {
    __shared__ float red[kWaves];
    const int64_t row = blockIdx.x;
    const u32x2* xr = reinterpret_cast<const u32x2*>(x + row * ldx);      // 4 bf16 per item
    const int n4 = cols >> 2;
    float amax = 0.f;
    for (int i = threadIdx.x; i < n4; i += kThreads) {
        const u32x2 w = xr[i];
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(bf16lo_to_f32(w[0])), fabsf(bf16hi_to_f32(w[0])))),
                     fmaxf(fabsf(bf16lo_to_f32(w[1])), fabsf(bf16hi_to_f32(w[1]))));
    }
    amax = fmaxf(block_max<kWaves>(amax, red), 1e-12f);
    const float inv = 448.0f / amax;
    if (threadIdx.x == 0) row_scale[row] = amax * (1.0f / 448.0f);
    unsigned int* qr = reinterpret_cast<unsigned int*>(out + row * ldo);
    for (int i = threadIdx.x; i < n4; i += kThreads) {
        const u32x2 w = xr[i];
        qr[i] = pack_fp8x4(bf16lo_to_f32(w[0]) * inv, bf16hi_to_f32(w[0]) * inv, bf16lo_to_f32(w[1]) * inv, bf16hi_to_f32(w[1]) * inv);
    }
}
}  // namespace

extern "C" wan_status_t wan_quantize_rows_fp8(const void* x_bf16, int64_t ldx, void* out_fp8, int64_t ldo,
                                              float* out_row_scale, int64_t rows, int cols, void* stream) {
    WAN_REQUIRE(x_bf16 && out_fp8 && out_row_scale, WAN_ERR_INVALID, "wan_quantize_rows_fp8: null tensor");
    WAN_REQUIRE(cols > 0 && cols % 4 == 0 && ldx >= cols && ldx % 4 == 0 && ldo >= cols && ldo % 4 == 0, WAN_ERR_INVALID,
                "wan_quantize_rows_fp8: cols=%d ldx=%lld ldo=%lld must be multiples of 4 with ld >= cols", cols, (long long)ldx, (long long)ldo);
    WAN_REQUIRE(rows >= 0, WAN_ERR_INVALID, "wan_quantize_rows_fp8: rows=%lld", (long long)rows);
    if (rows == 0) return WAN_OK;
    hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3((unsigned)rows), dim3(kThreads), 0, (hipStream_t)stream, (const bf16_t*)x_bf16,
                       ldx, (unsigned char*)out_fp8, ldo, out_row_scale, cols);
    WAN_CHECK_LAUNCH("wan_quantize_rows_fp8");
    return WAN_OK;
}

static wan_status_t rmsnorm_rope_impl_ex(void* x0, const float* w0, void* x1, const float* w1,
                                         int64_t ld, int64_t rows, int dim, int head_dim, float eps,
                                         const float* rope_cos, const float* rope_sin, const wan_rope_params* rp,
                                         float x0_scale, void* out0, void* out1, int out_slabs, int out_batch, void* stream,
                                         float x1_scale, int out_fp8, int out_split = 0);

static wan_status_t rmsnorm_rope_impl(void* x0, const float* w0, void* x1, const float* w1,
                                      int64_t ld, int64_t rows, int dim, int head_dim, float eps,
                                      const float* rope_cos, const float* rope_sin, const wan_rope_params* rp,
                                      float x0_scale, void* out0, void* out1, int out_slabs, int out_batch, void* stream) {
    return rmsnorm_rope_impl_ex(x0, w0, x1, w1, ld, rows, dim, head_dim, eps, rope_cos, rope_sin, rp, x0_scale, out0, out1, out_slabs,
                                out_batch, stream, 1.0f, 0);
}

static wan_status_t rmsnorm_rope_impl_ex(void* x0, const float* w0, void* x1, const float* w1,
                                         int64_t ld, int64_t rows, int dim, int head_dim, float eps,
                                         const float* rope_cos, const float* rope_sin, const wan_rope_params* rp,
                                         float x0_scale, void* out0, void* out1, int out_slabs, int out_batch, void* stream,
                                         float x1_scale, int out_fp8, int out_split) {
    WAN_REQUIRE(x0 && w0, WAN_ERR_INVALID, "wan_rmsnorm_rope: null tensor");
    if (out_fp8) {
        WAN_REQUIRE(out0 != nullptr && (out1 != nullptr) == (x1 != nullptr), WAN_ERR_INVALID, "wan_rmsnorm_rope_fp8: one output per input tensor");
        WAN_REQUIRE(x1_scale == x1_scale && x1_scale != 0.f, WAN_ERR_INVALID, "wan_rmsnorm_rope_fp8: x1_scale must be a non-zero number");
    } else if (out0 || out1) {
        WAN_REQUIRE(rp != nullptr && rope_cos != nullptr, WAN_ERR_INVALID, "wan_rmsnorm_rope_sp: needs rope tables and parameters");
        WAN_REQUIRE(out0 != nullptr && (out1 != nullptr) == (x1 != nullptr), WAN_ERR_INVALID, "wan_rmsnorm_rope_sp: one output per input tensor");
        WAN_REQUIRE(out_slabs > 0 && out_batch > 0 && dim % out_slabs == 0 && (dim / out_slabs) % 8 == 0 &&
                        rows == (int64_t)out_batch * rp->rows_per_batch, WAN_ERR_INVALID,
                    "wan_rmsnorm_rope_sp: slabs=%d batch=%d rows=%lld rows_per_batch=%lld dim=%d", out_slabs, out_batch,
                    (long long)rows, (long long)rp->rows_per_batch, dim);
        WAN_REQUIRE(out_split >= 0 && out_split % 8 == 0 && out_split < dim / out_slabs, WAN_ERR_INVALID,
                    "wan_rmsnorm_rope_sp_split: split=%d must be a multiple of 8 inside a slab of %d channels (0 = one group)", out_split,
                    dim / out_slabs);
    }
    WAN_REQUIRE(x0_scale == x0_scale && x0_scale != 0.f, WAN_ERR_INVALID, "wan_rmsnorm_rope: x0_scale must be a non-zero number (1 = none)");
    WAN_REQUIRE((x1 == nullptr) == (w1 == nullptr), WAN_ERR_INVALID, "wan_rmsnorm_rope: x1/w1 must both be set or both NULL");
    WAN_REQUIRE(dim > 0 && dim % 8 == 0 && ld % 8 == 0 && ld >= dim, WAN_ERR_INVALID,
                "wan_rmsnorm_rope: dim=%d ld=%lld must be multiples of 8, ld >= dim", dim, (long long)ld);
    WAN_REQUIRE(dim <= 8192, WAN_ERR_UNSUPPORTED, "wan_rmsnorm_rope: dim=%d > 8192", dim);
    WAN_REQUIRE(head_dim > 0 && head_dim % 8 == 0 && head_dim <= 256 && dim % head_dim == 0, WAN_ERR_INVALID,
                "wan_rmsnorm_rope: head_dim=%d invalid for dim=%d", head_dim, dim);
    WAN_REQUIRE((rope_cos == nullptr) == (rope_sin == nullptr), WAN_ERR_INVALID, "wan_rmsnorm_rope: cos/sin mismatch");
    RopeDev d = {};
    d.rows_per_batch = rows > 0 ? rows : 1;
    if (rope_cos) {
        WAN_REQUIRE(rp != nullptr, WAN_ERR_INVALID, "wan_rmsnorm_rope: rope tables given without wan_rope_params");
        WAN_REQUIRE(rp->F > 0 && rp->Hp > 0 && rp->Wp > 0 && rp->max_pos > 0 && rp->rows_per_batch > 0,
                    WAN_ERR_INVALID, "wan_rmsnorm_rope: bad grid (%d,%d,%d)", rp->F, rp->Hp, rp->Wp);
        WAN_REQUIRE(rp->mode >= 0 && rp->mode <= 2, WAN_ERR_INVALID, "wan_rmsnorm_rope: mode=%d", rp->mode);
        d.F = rp->F; d.Hp = rp->Hp; d.Wp = rp->Wp; d.mode = rp->mode; d.f_src = rp->f_src;
        d.ground_end = rp->ground_end; d.max_pos = rp->max_pos;
        d.token_offset = rp->token_offset; d.rows_per_batch = rp->rows_per_batch;
        const int c = head_dim / 2;                 // split (c - 2*(c/3), c/3, c/3)  wan_transformer3d.py:141
        d.ct = c - 2 * (c / 3); d.ch = c / 3;
    }
    if (rows == 0) return WAN_OK;
    hipStream_t s = (hipStream_t)stream;
    const int nv = (dim / 8 + kThreads - 1) / kThreads;
    dim3 grid((unsigned)rows, x1 ? 2 : 1), block(kThreads);
    const int R = wan_tune(WAN_TUNE_ROW_GROUP);                   // rows per workgroup: 2 (default), 4, or 1 = the one-row kernel
    if ((R == 2 || R == 4) && head_dim <= 256) {
        dim3 ggrid((unsigned)((rows + R - 1) / R), x1 ? 2 : 1);
#define RG_CASE(N) case N: if (R == 4) hipLaunchKernelGGL((rmsnorm_rope_rows_kernel<N, 4>), ggrid, block, 0, s, (bf16_t*)x0, w0, (bf16_t*)x1, w1, ld, rows, dim, head_dim, eps, rope_cos, rope_sin, d, x0_scale, (bf16_t*)out0, (bf16_t*)out1, out_slabs, out_batch, x1_scale, out_fp8, out_split); \
                           else hipLaunchKernelGGL((rmsnorm_rope_rows_kernel<N, 2>), ggrid, block, 0, s, (bf16_t*)x0, w0, (bf16_t*)x1, w1, ld, rows, dim, head_dim, eps, rope_cos, rope_sin, d, x0_scale, (bf16_t*)out0, (bf16_t*)out1, out_slabs, out_batch, x1_scale, out_fp8, out_split); break;
        switch (nv) { RG_CASE(1) RG_CASE(2) RG_CASE(3) RG_CASE(4) }
#undef RG_CASE
        WAN_CHECK_LAUNCH("wan_rmsnorm_rope");
        return WAN_OK;
    }
#define RR_CASE(N) case N: hipLaunchKernelGGL(rmsnorm_rope_kernel<N>, grid, block, 0, s, (bf16_t*)x0, w0, (bf16_t*)x1, w1, ld, dim, head_dim, eps, rope_cos, rope_sin, d, x0_scale, (bf16_t*)out0, (bf16_t*)out1, out_slabs, out_batch, x1_scale, out_fp8, out_split); break;
    switch (nv) { RR_CASE(1) RR_CASE(2) RR_CASE(3) RR_CASE(4) }
#undef RR_CASE
    WAN_CHECK_LAUNCH("wan_rmsnorm_rope");
    return WAN_OK;
}

extern "C" wan_status_t wan_rmsnorm_rope(void* x0, const float* w0, void* x1, const float* w1,
                                         int64_t ld, int64_t rows, int dim, int head_dim, float eps,
                                         const float* rope_cos, const float* rope_sin,
                                         const wan_rope_params* rp, float x0_scale, void* stream) {
    return rmsnorm_rope_impl(x0, w0, x1, w1, ld, rows, dim, head_dim, eps, rope_cos, rope_sin, rp, x0_scale, nullptr, nullptr, 1, 1, stream);
}

extern "C" wan_status_t wan_rmsnorm_rope_sp(const void* x0, const float* w0, const void* x1, const float* w1,
                                            int64_t ld, int64_t rows, int dim, int head_dim, float eps,
                                            const float* rope_cos, const float* rope_sin, const wan_rope_params* rp,
                                            float x0_scale, void* wire0, void* wire1, int slabs, int batch, void* stream) {
    WAN_REQUIRE(wire0 != nullptr, WAN_ERR_INVALID, "wan_rmsnorm_rope_sp: null wire buffer");
    return rmsnorm_rope_impl(const_cast<void*>(x0), w0, const_cast<void*>(x1), w1, ld, rows, dim, head_dim, eps, rope_cos, rope_sin, rp,
                             x0_scale, wire0, wire1, slabs, batch, stream);
}

extern "C" wan_status_t wan_rmsnorm_rope_sp_split(const void* x0, const float* w0, const void* x1, const float* w1,
                                                  int64_t ld, int64_t rows, int dim, int head_dim, float eps,
                                                  const float* rope_cos, const float* rope_sin, const wan_rope_params* rp,
                                                  float x0_scale, void* wire0, void* wire1, int slabs, int batch, int split, void* stream) {
    WAN_REQUIRE(wire0 != nullptr, WAN_ERR_INVALID, "wan_rmsnorm_rope_sp_split: null wire buffer");
    return rmsnorm_rope_impl_ex(const_cast<void*>(x0), w0, const_cast<void*>(x1), w1, ld, rows, dim, head_dim, eps, rope_cos, rope_sin, rp,
                                x0_scale, wire0, wire1, slabs, batch, stream, 1.0f, 0, split);
}

extern "C" wan_status_t wan_rmsnorm_rope_fp8(const void* x0, const float* w0, const void* x1, const float* w1,
                                             int64_t ld, int64_t rows, int dim, int head_dim, float eps,
                                             const float* rope_cos, const float* rope_sin, const wan_rope_params* rp,
                                             float x0_scale, float x1_scale, void* out0_fp8, void* out1_fp8, void* stream) {
    WAN_REQUIRE(out0_fp8 != nullptr, WAN_ERR_INVALID, "wan_rmsnorm_rope_fp8: null output");
    return rmsnorm_rope_impl_ex(const_cast<void*>(x0), w0, const_cast<void*>(x1), w1, ld, rows, dim, head_dim, eps, rope_cos, rope_sin, rp,
                                x0_scale, out0_fp8, out1_fp8, 1, 1, stream, x1_scale, 1);
}

// ------------------------------------------------------------------ K smoothing for the fp8 QK^T attention variant
// softmax_j(q_i . k_j) does not change when one vector mu is subtracted from every k_j (it adds q_i . mu to a whole row), so the
// e4m3 copy of k is taken of k - mean_tokens(k): a channel that carries a large common offset (the "massive activation" pattern of
// trained checkpoints; SageAttention's smooth_k) would otherwise spend its 3 mantissa bits on the offset.  Deterministic two-stage
// column mean (fixed summation order: no atomics), then one elementwise pass that writes both e4m3 operands.
namespace {
constexpr int kMeanChunks = 128;

// grid (ceil(dim / 1024), kMeanChunks, batch), 128 threads x 8 columns; rows [0, valid_rows) of each sample
__global__ __launch_bounds__(128) void col_partial_sums_kernel(const bf16_t* __restrict__ x, int64_t ld, int64_t rows_per_batch,
                                                               int valid_rows, int dim, float* __restrict__ partial) {
    const int c0 = (blockIdx.x * 128 + threadIdx.x) * 8;
    if (c0 >= dim) return;
    const int chunk = blockIdx.y, b = blockIdx.z;
    const int per = (valid_rows + kMeanChunks - 1) / kMeanChunks;
    const int r0 = chunk * per, r1 = min(valid_rows, r0 + per);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16_t* p = x + ((int64_t)b * rows_per_batch + r0) * ld + c0;
    for (int r = r0; r < r1; ++r, p += ld) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[2 * j] += bf16lo_to_f32(v[j]); s[2 * j + 1] += bf16hi_to_f32(v[j]); }
    }
    float* o = partial + ((int64_t)b * kMeanChunks + chunk) * dim + c0;
    *reinterpret_cast<float4*>(o) = make_float4(s[0], s[1], s[2], s[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(s[4], s[5], s[6], s[7]);
}

__global__ __launch_bounds__(256) void col_mean_finish_kernel(const float* __restrict__ partial, int dim, float inv_rows, float* __restrict__ mean) {
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (c >= dim) return;
    float s = 0.f;
    for (int k = 0; k < kMeanChunks; ++k) s += partial[((int64_t)b * kMeanChunks + k) * dim + c];
    mean[(int64_t)b * dim + c] = s * inv_rows;
}

// one workgroup per row; q8 = e4m3(q * q_scale), k8 = e4m3((k - mean[b]) * k_scale); dense [rows][dim] byte outputs.
// Either operand may be absent (q == nullptr or k == nullptr, workgroup-uniform): the Ulysses head-group pipeline quantises k once
// when it has arrived and every q group when ITS exchange completes.
__global__ __launch_bounds__(kThreads) void qk_quantize_fp8_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, int64_t ld,
                                                                   int dim, int64_t rows_per_batch, const float* __restrict__ mean,
                                                                   float q_scale, float k_scale, unsigned char* __restrict__ q8,
                                                                   unsigned char* __restrict__ k8) {
    const int64_t row = blockIdx.x;
    const float* mu = mean ? mean + (row / rows_per_batch) * dim : nullptr;
    for (int c = threadIdx.x * 8; c < dim; c += kThreads * 8) {
        if (q != nullptr) {
            const u32x4 qv = *reinterpret_cast<const u32x4*>(q + row * ld + c);
            float qf[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { qf[2 * j] = bf16lo_to_f32(qv[j]) * q_scale; qf[2 * j + 1] = bf16hi_to_f32(qv[j]) * q_scale; }
            const u32x2 qo = {pack_fp8x4(qf[0], qf[1], qf[2], qf[3]), pack_fp8x4(qf[4], qf[5], qf[6], qf[7])};
            *reinterpret_cast<u32x2*>(q8 + row * (int64_t)dim + c) = qo;
        }
        if (k != nullptr) {
            const u32x4 kv = *reinterpret_cast<const u32x4*>(k + row * ld + c);
            float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (mu) {
                const float4 m0 = *reinterpret_cast<const float4*>(mu + c), m1 = *reinterpret_cast<const float4*>(mu + c + 4);
                m[0] = m0.x; m[1] = m0.y; m[2] = m0.z; m[3] = m0.w; m[4] = m1.x; m[5] = m1.y; m[6] = m1.z; m[7] = m1.w;
            }
            float kf[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                kf[2 * j] = (bf16lo_to_f32(kv[j]) - m[2 * j]) * k_scale; kf[2 * j + 1] = (bf16hi_to_f32(kv[j]) - m[2 * j + 1]) * k_scale;
            }
            const u32x2 ko = {pack_fp8x4(kf[0], kf[1], kf[2], kf[3]), pack_fp8x4(kf[4], kf[5], kf[6], kf[7])};
            *reinterpret_cast<u32x2*>(k8 + row * (int64_t)dim + c) = ko;
        }
    }
}
}  // namespace

extern "C" int64_t wan_col_mean_workspace_bytes(int batch, int dim) {
    return batch > 0 && dim > 0 ? (int64_t)batch * kMeanChunks * dim * (int64_t)sizeof(float) : 0;
}

extern "C" wan_status_t wan_col_mean_bf16(const void* x_bf16, int64_t ld, int64_t rows_per_batch, int valid_rows, int batch, int dim,
                                          void* workspace, float* mean, void* stream) {
    WAN_REQUIRE(x_bf16 && workspace && mean, WAN_ERR_INVALID, "wan_col_mean_bf16: null tensor");
    WAN_REQUIRE(dim > 0 && dim % 8 == 0 && ld % 8 == 0 && ld >= dim, WAN_ERR_INVALID, "wan_col_mean_bf16: dim=%d ld=%lld", dim, (long long)ld);
    WAN_REQUIRE(batch > 0 && valid_rows > 0 && rows_per_batch >= valid_rows, WAN_ERR_INVALID,
                "wan_col_mean_bf16: batch=%d valid_rows=%d rows_per_batch=%lld", batch, valid_rows, (long long)rows_per_batch);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(col_partial_sums_kernel, dim3((unsigned)((dim + 1023) / 1024), kMeanChunks, (unsigned)batch), dim3(128), 0, s,
                       (const bf16_t*)x_bf16, ld, rows_per_batch, valid_rows, dim, (float*)workspace);
    hipLaunchKernelGGL(col_mean_finish_kernel, dim3((unsigned)((dim + 255) / 256), (unsigned)batch), dim3(256), 0, s,
                       (const float*)workspace, dim, 1.0f / (float)valid_rows, mean);
    WAN_CHECK_LAUNCH("wan_col_mean_bf16");
    return WAN_OK;
}

extern "C" wan_status_t wan_qk_quantize_fp8(const void* q_bf16, const void* k_bf16, int64_t ld, int64_t rows, int dim,
                                            int64_t rows_per_batch, const float* k_mean, float q_scale, float k_scale,
                                            void* q8, void* k8, void* stream) {
    WAN_REQUIRE((q_bf16 != nullptr) == (q8 != nullptr) && (k_bf16 != nullptr) == (k8 != nullptr) && (q_bf16 || k_bf16), WAN_ERR_INVALID,
                "wan_qk_quantize_fp8: each operand comes with its output (q and q8, k and k8), and at least one of the two is given");
    WAN_REQUIRE(k_bf16 != nullptr || k_mean == nullptr, WAN_ERR_INVALID, "wan_qk_quantize_fp8: k_mean without k");
    WAN_REQUIRE(dim > 0 && dim % 8 == 0 && ld % 8 == 0 && ld >= dim && rows >= 0 && rows_per_batch > 0, WAN_ERR_INVALID,
                "wan_qk_quantize_fp8: dim=%d ld=%lld rows=%lld rows_per_batch=%lld", dim, (long long)ld, (long long)rows, (long long)rows_per_batch);
    WAN_REQUIRE(q_scale == q_scale && k_scale == k_scale && q_scale != 0.f && k_scale != 0.f, WAN_ERR_INVALID, "wan_qk_quantize_fp8: scales must be non-zero numbers");
    if (rows == 0) return WAN_OK;
    hipLaunchKernelGGL(qk_quantize_fp8_kernel, dim3((unsigned)rows), dim3(kThreads), 0, (hipStream_t)stream, (const bf16_t*)q_bf16,
                       (const bf16_t*)k_bf16, ld, dim, rows_per_batch, k_mean, q_scale, k_scale, (unsigned char*)q8, (unsigned char*)k8);
    WAN_CHECK_LAUNCH("wan_qk_quantize_fp8");
    return WAN_OK;
}
