// What ONE compute unit can store (and load) per clock: G workgroups of 256 threads (one per CU while G <= 256), each streaming over
// its own slice of a 2 GiB buffer with dwordx4 accesses, fully coalesced.  Bandwidth against G tells a per-CU limit from a chip limit:
// the persistent GEMM's epilogue (a lone workgroup per CU) moved its bytes at ~13-16 B per clock whatever the access shape.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/cu_store_rate.hip -o tools/probe/cu_store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>   // 0 store, 1 load, 2 copy (load + store)
__global__ __launch_bounds__(256) void stream_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t per_wg, uint4* sink) {
    const size_t base = (size_t)blockIdx.x * per_wg;
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < per_wg; i += 256 * 4) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t k = i + (size_t)u * 256;
            if (MODE >= 1) v[u] = k < per_wg ? src[base + k] : uint4{0, 0, 0, 0}; else v[u] = uint4{(unsigned)k, 1, 2, 3};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t k = i + (size_t)u * 256;
            if (MODE != 1) { if (k < per_wg) dst[base + k] = v[u]; }
            else { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
    }
    if (MODE == 1 && acc.x == 0x12345678u) *sink = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30, n16 = bytes / 16;
    uint4 *a, *b, *sink;
    HIP(hipMalloc(&a, bytes)); HIP(hipMalloc(&b, bytes)); HIP(hipMalloc(&sink, 64));
    HIP(hipMemset(a, 1, bytes)); HIP(hipMemset(b, 2, bytes));
    hipDeviceProp_t p; HIP(hipGetDeviceProperties(&p, 0));
    const double ghz = 2.1;
    printf("device %s, %d CUs (B/clk figures assume %.1f GHz)\n", p.name, p.multiProcessorCount, ghz);
    hipEvent_t e0, e1; HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode)
        for (int G : {8, 32, 64, 128, 256, 512, 2048}) {
            // every workgroup moves 8 MiB per direction: the time is that of one workgroup's stream, G of them side by side
            const size_t per_wg = ((size_t)8 << 20) / 16;
            if ((size_t)G * per_wg > n16) continue;
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                HIP(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(G), dim3(256), 0, 0, a, b, per_wg, sink);
                if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(G), dim3(256), 0, 0, a, b, per_wg, sink);
                if (mode == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(G), dim3(256), 0, 0, a, b, per_wg, sink);
                HIP(hipEventRecord(e1)); HIP(hipEventSynchronize(e1));
                float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
                if (rep) best = ms < best ? ms : best;
            }
            const double gb = (double)G * per_wg * 16 * (mode == 2 ? 2 : 1) / 1e9;
            const int cus = G < p.multiProcessorCount ? G : p.multiProcessorCount;
            printf("%-5s G=%4d: %7.3f ms  %7.1f GB/s  = %6.1f GB/s per busy CU = %5.1f B/clk/CU\n", mode == 0 ? "store" : mode == 1 ? "load" : "copy", G, best,
                   gb / best * 1e3, gb / best * 1e3 / cus, gb / best * 1e3 / cus / ghz);
        }
    return 0;
}
