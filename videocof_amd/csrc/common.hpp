// Shared device helpers for the gfx950 kernels (wave64, MFMA, LDS-DMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/wan_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define WAN_WAVE 64

// host-side error plumbing -----------------------------------------------------
void wan_set_error(const char* fmt, ...);
int wan_cu_count();      // compute units of the current device (256 on MI355X), cached; 256 if it cannot be queried
// developer switches (api.cpp): environment read once at first use, overridable with wan_set_tuning()
enum { WAN_TUNE_ATTN_TAIL = 0, WAN_TUNE_ATTN_FAST, WAN_TUNE_ATTN_XCD_MAP, WAN_TUNE_GEMM_GM, WAN_TUNE_GEMM_PHASES,
       WAN_TUNE_DEBUG_CHECKS, WAN_TUNE_GEMM_VARIANT, WAN_TUNE_CONV_XCD, WAN_TUNE_GEMM_W4, WAN_TUNE_CONV_FAST, WAN_TUNE_CONV_PATCH, WAN_TUNE_ATTN_REF, WAN_TUNE_CONV_HEAD, WAN_TUNE_GEMM_EXP, WAN_TUNE_GEMM_RING, WAN_TUNE_GEMM_PK, WAN_TUNE_GEMM_PK_WORKERS, WAN_TUNE_GEMM_PK_MIN_UNITS, WAN_TUNE_GEMM_PK_ORDER, WAN_TUNE_GEMM_PK_FORM, WAN_TUNE_ROW_GROUP, WAN_TUNE_SP_INLINE, WAN_TUNE_GEMM_SPLITK, WAN_TUNE_CONV_MFMA, WAN_TUNE_ATTN_PERSIST, WAN_TUNE_COUNT };
int wan_tune(int which);
void wan_note_attn_variant(int variant);     // read back through wan_get_tuning("last_attn_variant")
#ifdef __cplusplus
#include <atomic>
wan_status_t wan_once_per_device(std::atomic<uint64_t>& done, wan_status_t (*init)());
#endif
#define WAN_REQUIRE(cond, code, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            wan_set_error(__VA_ARGS__);         \
            return (code);                      \
        }                                       \
    } while (0)
#define WAN_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            wan_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return WAN_ERR_LAUNCH;                                                    \
        }                                                                             \
    } while (0)

// device helpers ---------------------------------------------------------------
__device__ __forceinline__ float bf16lo_to_f32(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi_to_f32(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    f32x2 v = {lo, hi};
    bf16x2 b = __builtin_convertvector(v, bf16x2);   // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(unsigned int, b);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blockDim.x = NW*64; `red` is NW floats of LDS. All threads get the result.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6;
    __syncthreads();   // protect `red` from the previous use
    if ((threadIdx.x & 63) == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide max for blockDim.x = NW*64 (same protocol as block_sum).
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int wid = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wid] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
    return t;
}

// four fp32 -> four OCP e4m3 bytes (v_cvt_pk_fp8_f32 x 2, round-to-nearest-even); inputs are clamped to the finite range
// +-448 first (e4m3fn has no infinity: an overflowing conversion would give NaN)
__device__ __forceinline__ unsigned int pack_fp8x4(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned int)w;
}

__device__ __forceinline__ float gelu_tanh_f32(float x) {
    // 0.5 x (1 + tanh(u)),  u = sqrt(2/pi) (x + 0.044715 x^3)   ==   x * sigmoid(2u) = x / (1 + 2^(-2 u log2 e))
    // 7 VALU (v_exp_f32 + v_rcp_f32, 1 ulp each) instead of the ~18 of expf() and an IEEE division; the result is
    // rounded to bf16 by every caller.
    constexpr float k1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    constexpr float k3 = k1 * 0.044715f;
    const float a = __builtin_fmaf(x * x, k3, k1) * x;
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a));
}

// async global -> LDS copy of 16 B per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gsrc,
        (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
