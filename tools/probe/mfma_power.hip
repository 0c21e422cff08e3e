// Probe: what the matrix pipes sustain CHIP-WIDE (256 CUs x 4 waves, ~ms-long launches, i.e. at the power limit) for the two bf16
// MFMA shapes a 128 x 128 wave tile can be built from, on random and on zero operands, alone and beside the LDS fragment reads a
// GEMM main loop issues (one ds_read_b128 per 2 MFMAs of 32x32x16 / per 4 of 16x16x32).  HIP events; TFLOP/s = issued MFMA FLOPs / time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int LDS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(const u32x4* src, float* out, int iters) {
    __shared__ u32x4 sm[4096];                               // 64 KiB of operand data
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = src[i];
    __syncthreads();
    u32x4 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = sm[(threadIdx.x * 8 + i) & 4095]; b[i] = sm[(threadIdx.x * 8 + i + 1024) & 4095]; }
    const u32x4* lp = sm + (threadIdx.x & 63);
    float xs[8], ex[8], sum = 0.f; unsigned pk[4] = {0, 0, 0, 0};
    unsigned ones; asm volatile("v_mov_b32 %0, 0x3f803f80" : "=v"(ones));
    for (int i = 0; i < 8; ++i) { xs[i] = -0.37f * (float)((threadIdx.x * 7 + i * 13) % 29); ex[i] = 0.f; }
    if constexpr (SHAPE == 32) {
        f32x16 acc[16];
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i * 4 + j]) : "v"(a[i]), "v"(b[j]));
                    if (LDS && ((i * 4 + j) & 1) == 0) { const int f = (i * 4 + j) / 2; if (f < 4) a[4 + f] = lp[(f * 64 + it * 7) & 4032]; else b[f] = lp[(f * 64 + it * 5) & 4032]; }
                    if (LDS >= 2) {      // the softmax stream of an attention tile: per 32x32x16 MFMA one v_exp_f32, one v_add_f32, half a v_cvt_pk_bf16_f32
                        // LDS == 3: HALF the exponentials (the optimistic bound of an exponential that produces two values per instruction,
                        // e.g. a packed-f16 form, with nothing else added); LDS == 4: none at all (adds and converts only)
                        if (LDS == 2 || LDS >= 5 || (LDS == 3 && ((i * 4 + j) & 1) == 0))
                        asm volatile("v_exp_f32 %0, %1" : "=v"(ex[(i * 4 + j) & 7]) : "v"(xs[(i * 4 + j) & 7]));
                        // LDS == 5 (round 6): the row sums taken from the PACKED bf16 pair with one v_dot2c_f32_bf16 (pair . (1, 1) + sum) per
                        // conversion instead of one v_add_f32 per exponential; LDS == 6: no row-sum instruction at all (the bound of that idea)
                        if (LDS < 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(ex[(i * 4 + j + 4) & 7]));
                        if (((i * 4 + j) & 1) == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(i * 4 + j) >> 1 & 3]) : "v"(ex[(i * 4 + j) & 7]), "v"(ex[(i * 4 + j + 1) & 7]));
                        if (LDS == 5 && ((i * 4 + j) & 1) == 0) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(sum) : "v"(pk[((i * 4 + j) >> 1) + 2 & 3]), "v"(ones));
                    }
                }
            if (LDS) for (int f = 0; f < 4; ++f) { a[f] = a[4 + f]; }
        }
        float s = sum + (float)pk[0] + (float)pk[1] + (float)pk[2] + (float)pk[3]; for (int i = 0; i < 16; ++i) s += acc[i][0];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x4 acc[64];
        for (int i = 0; i < 64; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i * 8 + j]) : "v"(a[i]), "v"(b[j]));
                    if (LDS && ((i * 8 + j) & 3) == 0) { const int f = (i * 8 + j) / 4; if (f < 8) a[f] = lp[(f * 64 + it * 7) & 4032]; else b[f - 8] = lp[(f * 64 + it * 5) & 4032]; }
                    if (LDS == 2 && ((i * 8 + j) & 1) == 0) {      // the same stream per FLOP: one exp + one add per two 16x16x32 MFMAs, a convert per four
                        const int q = (i * 8 + j) >> 1;
                        asm volatile("v_exp_f32 %0, %1" : "=v"(ex[q & 7]) : "v"(xs[q & 7]));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(ex[(q + 4) & 7]));
                        if ((q & 1) == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(q >> 1) & 3]) : "v"(ex[q & 7]), "v"(ex[(q + 1) & 7]));
                    }
                }
        }
        float s = sum + (float)pk[0] + (float)pk[1] + (float)pk[2] + (float)pk[3]; for (int i = 0; i < 64; ++i) s += acc[i][0];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

// Round 6: the MIXED body the round-5 review asked about -- QK^T on 32x32x16 (its S^T layout feeds the softmax lane-locally), P.V on
// 16x16x32 (the shape that sustains more FLOP/s per watt on its own) -- with the same LDS fragment reads and softmax stream PER FLOP as the
// rows above: per iteration 8 MFMAs of 32x32x16 + 16 of 16x16x32 (equal FLOPs), 4 + 4 ds_read_b128, 16 v_exp + 16 v_add + 8 v_cvt_pk.
// (An upper bound for that design: the P fragments would also have to change lanes between the two shapes, which is not modelled.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void kmix(const u32x4* src, float* out, int iters) {
    __shared__ u32x4 sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = src[i];
    __syncthreads();
    u32x4 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = sm[(threadIdx.x * 8 + i) & 4095]; b[i] = sm[(threadIdx.x * 8 + i + 1024) & 4095]; }
    const u32x4* lp = sm + (threadIdx.x & 63);
    float xs[8], ex[8], sum = 0.f; unsigned pk[4] = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) { xs[i] = -0.37f * (float)((threadIdx.x * 7 + i * 13) % 29); ex[i] = 0.f; }
    f32x16 accS[8];
    f32x4 accO[16];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) accS[i][r] = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) accO[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {                      // the QK^T half
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(accS[s]) : "v"(a[s & 3]), "v"(b[s >> 1]));
            if ((s & 1) == 0) a[4 + (s >> 1)] = lp[((s >> 1) * 64 + it * 7) & 4032];
            asm volatile("v_exp_f32 %0, %1" : "=v"(ex[s & 7]) : "v"(xs[s & 7]));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(ex[(s + 4) & 7]));
            if ((s & 1) == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(s >> 1) & 3]) : "v"(ex[s & 7]), "v"(ex[(s + 1) & 7]));
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {                     // the P.V half: twice the instructions for the same FLOPs
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(accO[s]) : "v"(a[s & 7]), "v"(b[(s >> 1) & 7]));
            if ((s & 3) == 0) b[4 + (s >> 2)] = lp[((s >> 2) * 64 + it * 5) & 4032];
            if ((s & 1) == 0) {
                const int q = s >> 1;
                asm volatile("v_exp_f32 %0, %1" : "=v"(ex[q & 7]) : "v"(xs[q & 7]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(ex[(q + 4) & 7]));
                if ((q & 1) == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(q >> 1) & 3]) : "v"(ex[q & 7]), "v"(ex[(q + 1) & 7]));
            }
        }
        for (int f = 0; f < 4; ++f) { a[f] = a[4 + f]; }
    }
    float s = sum + (float)pk[0] + (float)pk[1] + (float)pk[2] + (float)pk[3];
    for (int i = 0; i < 8; ++i) s += accS[i][0];
    for (int i = 0; i < 16; ++i) s += accO[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Round 6: would TWO waves per SIMD on 16x16x32 (an 8-wave workgroup, 32 query rows per wave: half the accumulators, every K / V^T fragment
// read feeding 2 MFMAs instead of 4) lift the issue bound that holds the one-wave 16x16x32 form at 1.35-1.43?  Same softmax stream and
// FLOPs per iteration pair as k<16, 2>: per wave and iteration 32 MFMAs, 16 ds_read_b128, 16 v_exp + 16 v_add + 8 v_cvt_pk.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k8w(const u32x4* src, float* out, int iters) {
    __shared__ u32x4 sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) sm[i] = src[i];
    __syncthreads();
    u32x4 a[4], b[8];
    for (int i = 0; i < 4; ++i) a[i] = sm[(threadIdx.x * 8 + i) & 4095];
    for (int i = 0; i < 8; ++i) b[i] = sm[(threadIdx.x * 8 + i + 1024) & 4095];
    const u32x4* lp = sm + (threadIdx.x & 63);
    float xs[8], ex[8], sum = 0.f; unsigned pk[4] = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) { xs[i] = -0.37f * (float)((threadIdx.x * 7 + i * 13) % 29); ex[i] = 0.f; }
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i * 8 + j]) : "v"(a[i]), "v"(b[j]));
                if (((i * 8 + j) & 1) == 0) { const int f = (i * 8 + j) / 2; if (f < 4) a[f] = lp[(f * 64 + it * 7) & 4032]; else if (f < 12) b[f - 4] = lp[(f * 64 + it * 5) & 4032]; else a[f - 12] = lp[(f * 64 + it * 3) & 4032]; }
                if (((i * 8 + j) & 1) == 0) {
                    const int q = (i * 8 + j) >> 1;
                    asm volatile("v_exp_f32 %0, %1" : "=v"(ex[q & 7]) : "v"(xs[q & 7]));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(ex[(q + 4) & 7]));
                    if ((q & 1) == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(q >> 1) & 3]) : "v"(ex[q & 7]), "v"(ex[(q + 1) & 7]));
                }
            }
    }
    float s = sum + (float)pk[0] + (float)pk[1] + (float)pk[2] + (float)pk[3]; for (int i = 0; i < 32; ++i) s += acc[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
void run_8w(const char* name, const u32x4* src, float* out) {
    const int iters = 10000;                                  // 8 waves x 32 MFMAs = the 4 waves x 64 of k<16, .> per iteration
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k8w, dim3(256), dim3(512), 0, 0, src, out, iters / 10);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k8w, dim3(256), dim3(512), 0, 0, src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 256.0 * 8 * iters * (32 * 16.0 * 16 * 32 * 2);
        printf("%-58s %8.3f ms  %7.0f TFLOP/s\n", name, ms, fl / ms / 1e9);
    }
}

void run_mix(const char* name, const u32x4* src, float* out) {
    const int iters = 40000;                                  // 16 x 32x32x16-equivalents per iteration, as run<32, .>
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kmix, dim3(256), dim3(256), 0, 0, src, out, iters / 10);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kmix, dim3(256), dim3(256), 0, 0, src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 256.0 * 4 * iters * (8 * 32.0 * 32 * 16 * 2 + 16 * 16.0 * 16 * 32 * 2);
        printf("%-58s %8.3f ms  %7.0f TFLOP/s\n", name, ms, fl / ms / 1e9);
    }
}

template <int SHAPE, int LDS> void run(const char* name, const u32x4* src, float* out) {
    const int iters = SHAPE == 32 ? 40000 : 10000;           // 640 000 MFMAs of either shape's flop count ratio 2:1 -> equal FLOPs
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(256), dim3(256), 0, 0, src, out, iters / 10);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(256), dim3(256), 0, 0, src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 256.0 * 4 * iters * (SHAPE == 32 ? 16 * 32.0 * 32 * 16 * 2 : 64 * 16.0 * 16 * 32 * 2);
        printf("%-58s %8.3f ms  %7.0f TFLOP/s\n", name, ms, fl / ms / 1e9);
    }
}
int main() {
    std::vector<unsigned> h(4096 * 4);
    u32x4* src; float* out; hipMalloc(&src, 65536); hipMalloc(&out, 256 * 512 * 4);
    for (int pass = 0; pass < 2; ++pass) {
        srand(1);
        for (auto& x : h) {                                   // two bf16 values ~ U(-1, 1) per word, or zeros
            auto bf = [] { float f = (rand() / (float)RAND_MAX) * 2 - 1; unsigned u; __builtin_memcpy(&u, &f, 4); return u >> 16; };
            x = pass == 0 ? (bf() | (bf() << 16)) : 0u;
        }
        hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
        const char* d = pass == 0 ? "random" : "zeros ";
        char nm[128];
        snprintf(nm, sizeof nm, "32x32x16, %s operands, MFMA only", d); run<32, 0>(nm, src, out);
        snprintf(nm, sizeof nm, "16x16x32, %s operands, MFMA only", d); run<16, 0>(nm, src, out);
        snprintf(nm, sizeof nm, "32x32x16, %s operands, + ds_read_b128 per 2 MFMA", d); run<32, 1>(nm, src, out);
        snprintf(nm, sizeof nm, "16x16x32, %s operands, + ds_read_b128 per 4 MFMA", d); run<16, 1>(nm, src, out);
        snprintf(nm, sizeof nm, "32x32x16, %s, + LDS reads + softmax VALU stream", d); run<32, 2>(nm, src, out);
        snprintf(nm, sizeof nm, "16x16x32, %s, + LDS reads + softmax VALU stream", d); run<16, 2>(nm, src, out);
        snprintf(nm, sizeof nm, "MIXED 32x32x16 (QK^T) + 16x16x32 (P.V), %s, + LDS + softmax", d); run_mix(nm, src, out);
        snprintf(nm, sizeof nm, "32x32x16, %s, + LDS reads + stream with HALF the v_exp", d); run<32, 3>(nm, src, out);
        snprintf(nm, sizeof nm, "32x32x16, %s, + LDS reads + stream with NO v_exp", d); run<32, 4>(nm, src, out);
        snprintf(nm, sizeof nm, "16x16x32, TWO waves / SIMD, %s, + LDS (1 per 2 MFMA) + softmax", d); run_8w(nm, src, out);
        snprintf(nm, sizeof nm, "32x32x16, %s, + LDS + softmax stream (reference again)", d); run<32, 2>(nm, src, out);
        snprintf(nm, sizeof nm, "32x32x16, %s, + LDS + stream, row sums by v_dot2c_f32_bf16", d); run<32, 5>(nm, src, out);
        snprintf(nm, sizeof nm, "32x32x16, %s, + LDS + stream with NO row-sum op", d); run<32, 6>(nm, src, out);
    }
    return 0;
}
