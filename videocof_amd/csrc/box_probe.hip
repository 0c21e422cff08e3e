// wan_box_probe: a fixed ~0.4 s calibration of THE BOX, not of the product kernels -- what this chip, in this chassis, at this
// moment sustains on (a) the matrix pipes under the instruction mix of a flash-attention tile (32x32x16 bf16 MFMAs on random operands
// beside LDS fragment reads and a softmax VALU stream: one v_exp_f32 + one v_add_f32 per MFMA, a packed bf16 convert per two), chip-wide
// and long enough to sit at the power limit, and (b) a plain 16-byte-per-lane HBM copy.  bench.py runs it before and after the timed
// region and prints `box` + `value_normalised`, so that two rounds measured on two boxes can be compared (MI355X boxes of this pool
// differ by ~3.5 % in the clock they hold under these kernels -- more than a round of kernel work moves the headline).
// The kernels here never change with the product kernels: a round that speeds attention up moves `value`, not the probe.
// No reference counterpart (the reference has no native code and no benchmark harness).
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kProbeOperandBytes = 65536;                 // 64 KiB of bf16 operand words, staged into LDS by every workgroup
constexpr int kProbeWGs = 256, kProbeThreads = 256;       // one 4-wave workgroup per CU, one wave per SIMD

__global__ __launch_bounds__(256) void probe_fill_kernel(unsigned* w, int n) {
    // two bf16 values ~ U(-1, 1) per word from a counter hash: the toggle rate of random data (zeros would flatter the clock)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        auto bf = [](unsigned r) { float f = (float)(r & 0xffffu) * (2.0f / 65535.0f) - 1.0f; unsigned u; __builtin_memcpy(&u, &f, 4); return u >> 16; };
        w[i] = bf(h) | (bf(h >> 16) << 16);
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe_mfma_mix_kernel(const u32x4* src, float* out, int iters) {
    __shared__ u32x4 sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = src[i];
    __syncthreads();
    u32x4 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = sm[(threadIdx.x * 8 + i) & 4095]; b[i] = sm[(threadIdx.x * 8 + i + 1024) & 4095]; }
    const u32x4* lp = sm + (threadIdx.x & 63);
    float xs[8], ex[8], sum = 0.f;
    unsigned pk[4] = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) { xs[i] = -0.37f * (float)((threadIdx.x * 7 + i * 13) % 29); ex[i] = 0.f; }
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = i * 4 + j;
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[s]) : "v"(a[i]), "v"(b[j]));
                if ((s & 1) == 0) {                          // one ds_read_b128 per two MFMAs (a GEMM / attention main loop's fragment reads)
                    const int f = s / 2;
                    if (f < 4) a[4 + f] = lp[(f * 64 + it * 7) & 4032]; else b[f] = lp[(f * 64 + it * 5) & 4032];
                }
                asm volatile("v_exp_f32 %0, %1" : "=v"(ex[s & 7]) : "v"(xs[s & 7]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(ex[(s + 4) & 7]));
                if ((s & 1) == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(s >> 1) & 3]) : "v"(ex[s & 7]), "v"(ex[(s + 1) & 7]));
            }
        for (int f = 0; f < 4; ++f) a[f] = a[4 + f];
    }
    float s = sum + (float)pk[0] + (float)pk[1] + (float)pk[2] + (float)pk[3];
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void probe_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

constexpr double kFlopPerIter = (double)kProbeWGs * 4 * 16 * (2.0 * 32 * 32 * 16);     // chip-wide MFMA FLOP of one loop iteration
}  // namespace

extern "C" int64_t wan_box_probe_scratch_bytes(void) {
    // operands + per-thread results + two 256 MiB copy buffers (larger than the 256 MB Infinity Cache together with the L2s)
    return (int64_t)kProbeOperandBytes + (int64_t)kProbeWGs * kProbeThreads * 4 + 2 * (int64_t)(256u << 20);
}

extern "C" wan_status_t wan_box_probe(wan_box_probe_result* res, void* scratch, int64_t scratch_bytes, int target_ms, void* stream_) {
    if (!res || !scratch) { wan_set_error("wan_box_probe: null result or scratch"); return WAN_ERR_INVALID; }
    if (scratch_bytes < wan_box_probe_scratch_bytes()) {
        wan_set_error("wan_box_probe: scratch of %lld bytes, need %lld", (long long)scratch_bytes, (long long)wan_box_probe_scratch_bytes());
        return WAN_ERR_INVALID;
    }
    if (target_ms <= 0) target_ms = 300;
    hipStream_t stream = (hipStream_t)stream_;
    char* base = (char*)scratch;
    u32x4* ops = (u32x4*)base;
    float* outv = (float*)(base + kProbeOperandBytes);
    const size_t copy_bytes = (size_t)256u << 20;
    u32x4* c0 = (u32x4*)(base + kProbeOperandBytes + kProbeWGs * kProbeThreads * 4);
    u32x4* c1 = (u32x4*)((char*)c0 + copy_bytes);
    hipEvent_t e[4];
    for (auto& ev : e) if (hipEventCreate(&ev) != hipSuccess) { wan_set_error("wan_box_probe: hipEventCreate failed"); return WAN_ERR_LAUNCH; }
    auto done = [&](wan_status_t st) { for (auto& ev : e) (void)hipEventDestroy(ev); return st; };

    probe_fill_kernel<<<64, 256, 0, stream>>>((unsigned*)ops, kProbeOperandBytes / 4);
    probe_fill_kernel<<<1024, 256, 0, stream>>>((unsigned*)c0, (int)(copy_bytes / 4));
    // (a) matrix pipes: a short launch sizes the long ones (1 ms-scale -> target_ms in all), then three equal launches; the LAST
    // TWO are the measurement (the first brings clocks / power to steady state)
    const int cal_iters = 2000;
    probe_mfma_mix_kernel<<<kProbeWGs, kProbeThreads, 0, stream>>>(ops, outv, cal_iters);      // warm: code object, LDS
    (void)hipEventRecord(e[0], stream);
    probe_mfma_mix_kernel<<<kProbeWGs, kProbeThreads, 0, stream>>>(ops, outv, cal_iters);
    (void)hipEventRecord(e[1], stream);
    if (hipEventSynchronize(e[1]) != hipSuccess) { wan_set_error("wan_box_probe: calibration launch failed: %s", hipGetErrorString(hipGetLastError())); return done(WAN_ERR_LAUNCH); }
    float cal_ms = 0.f;
    (void)hipEventElapsedTime(&cal_ms, e[0], e[1]);
    if (!(cal_ms > 0.f)) cal_ms = 1.f;
    long iters = (long)((double)cal_iters * (target_ms / 3.0) / cal_ms);
    if (iters < cal_iters) iters = cal_iters;
    if (iters > 4000000) iters = 4000000;
    probe_mfma_mix_kernel<<<kProbeWGs, kProbeThreads, 0, stream>>>(ops, outv, (int)iters);
    (void)hipEventRecord(e[0], stream);
    probe_mfma_mix_kernel<<<kProbeWGs, kProbeThreads, 0, stream>>>(ops, outv, (int)iters);
    probe_mfma_mix_kernel<<<kProbeWGs, kProbeThreads, 0, stream>>>(ops, outv, (int)iters);
    (void)hipEventRecord(e[1], stream);
    // (b) copy: 256 MiB -> 256 MiB, 4 warm + 40 timed passes back and forth (read + write = 2 x 256 MiB of HBM traffic per pass)
    const size_t n16 = copy_bytes / 16;
    const int copy_reps = 40;
    for (int r = 0; r < 4; ++r) probe_copy_kernel<<<256 * 16, 256, 0, stream>>>((r & 1) ? c1 : c0, (r & 1) ? c0 : c1, n16);
    (void)hipEventRecord(e[2], stream);
    for (int r = 0; r < copy_reps; ++r) probe_copy_kernel<<<256 * 16, 256, 0, stream>>>((r & 1) ? c1 : c0, (r & 1) ? c0 : c1, n16);
    (void)hipEventRecord(e[3], stream);
    if (hipEventSynchronize(e[3]) != hipSuccess) { wan_set_error("wan_box_probe: launch failed: %s", hipGetErrorString(hipGetLastError())); return done(WAN_ERR_LAUNCH); }
    float mfma_ms = 0.f, copy_ms = 0.f;
    (void)hipEventElapsedTime(&mfma_ms, e[0], e[1]);
    (void)hipEventElapsedTime(&copy_ms, e[2], e[3]);
    res->mfma_ms = mfma_ms;
    res->copy_ms = copy_ms;
    res->mfma_mix_tflops = (float)(2.0 * iters * kFlopPerIter / (mfma_ms * 1e-3) / 1e12);
    res->copy_tbps = (float)(2.0 * copy_bytes * copy_reps / (copy_ms * 1e-3) / 1e12);
    return done(WAN_OK);
}
