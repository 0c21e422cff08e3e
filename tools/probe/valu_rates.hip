// Probe: issue cost (cycles per instruction, one wave64 alone on its SIMD) of the VALU instructions the attention kernels' softmax
// streams are made of.  16 independent chains per loop body, 4096 iterations, s_memtime around the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define BODY16(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7) STMT(8) STMT(9) STMT(10) STMT(11) STMT(12) STMT(13) STMT(14) STMT(15)

template <int OP>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, float seed) {
    float x[16], y[16];
    for (int i = 0; i < 16; ++i) { x[i] = seed + 0.001f * i + 0.0001f * threadIdx.x; y[i] = 0.5f + 0.01f * i; }
    i16x2 h[16]; f32x2 p[16];
    for (int i = 0; i < 16; ++i) { h[i] = i16x2{0, 0}; p[i] = f32x2{x[i], y[i]}; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 4096; ++it) {
#define S_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
#define S_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
#define S_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(y[i]));
#define S_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
#define S_CVTBF(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(x[i]), "v"(y[i]));
#define S_CVTF8(i) asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(h[i]) : "v"(x[i]), "v"(y[i]), "v"(y[0]));
#define S_CVTF8P(i) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(h[i]) : "v"(x[i]), "v"(y[i]));
#define S_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(y[i]));
#define S_SWAP(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[i]), "+v"(y[i]));
#define S_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
#define S_LDEXP(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[i]) : "v"(1));
        if constexpr (OP == 0) { BODY16(S_EXP) }
        if constexpr (OP == 1) { BODY16(S_ADD) }
        if constexpr (OP == 2) { BODY16(S_FMA) }
        if constexpr (OP == 3) { BODY16(S_PKADD) }
        if constexpr (OP == 4) { BODY16(S_CVTBF) }
        if constexpr (OP == 5) { BODY16(S_CVTF8) }
        if constexpr (OP == 6) { BODY16(S_CVTF8P) }
        if constexpr (OP == 7) { BODY16(S_MAX3) }
        if constexpr (OP == 8) { BODY16(S_SWAP) }
        if constexpr (OP == 9) { BODY16(S_RCP) }
        if constexpr (OP == 10) { BODY16(S_LDEXP) }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i] + y[i] + (float)h[i][0] + p[i][0];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP> void run(const char* name, float seed) {
    float* out; long long* cyc; long long h;
    hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, out, cyc, seed);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, out, cyc, seed);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s %6.2f cycles / instruction (shader clock; %lld cycles for %d instructions)\n", name, (double)h / (4096.0 * 16), h, 4096 * 16);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1>("v_add_f32", 1.0f); run<2>("v_fma_f32", 0.5f); run<7>("v_max3_f32", 1.0f); run<10>("v_ldexp_f32", 1.0f);
    run<0>("v_exp_f32", -0.5f); run<9>("v_rcp_f32", 1.5f);
    run<3>("v_pk_add_f32", 1.0f); run<4>("v_cvt_pk_bf16_f32", 1.0f); run<6>("v_cvt_pk_fp8_f32", 1.0f); run<5>("v_cvt_scalef32_pk_fp8_f32", 1.0f);
    run<8>("v_permlane32_swap_b32", 1.0f);
    return 0;
}
