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

// The persistent GEMM's fp32 epilogue as a store pattern: a workgroup writes 256 x 256 fp32 tiles of a [rows][5120] matrix (row stride
// 20 KB), wave (wr, wc) its 128 x 128 quarter.  FORM 0: dword stores, a register = 4 rows x 16 consecutive columns (64 B per row group);
// FORM 1: dwordx4 stores, 4 rows x 64 consecutive columns (256 B per row group); FORM 2: the swapped-product form, dwordx4 per lane
// with consecutive lanes on consecutive ROWS (16 rows x 64 B per instruction, every lane a request of its own).
template <int FORM>
__global__ __launch_bounds__(256) void tile_store_kernel(float* __restrict__ out, int ldo, int tiles_per_wg) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1, l15 = lane & 15, kg = lane >> 4;
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int tile = blockIdx.x * tiles_per_wg + t;
        const int tm = tile / 20, tn = tile % 20;                       // 20 tiles across N = 5120
        float* base = out + ((size_t)tm * 256 + wr * 128) * ldo + tn * 256 + wc * 128;
        const float v = (float)t;
        if (FORM == 0) {
#pragma unroll 4
            for (int i = 0; i < 8; ++i)
                for (int r = 0; r < 4; ++r)
                    for (int j = 0; j < 8; ++j) base[(size_t)(i * 16 + 4 * kg + r) * ldo + j * 16 + l15] = v;
        } else if (FORM == 1) {
#pragma unroll 4
            for (int i = 0; i < 8; ++i)
                for (int r = 0; r < 4; ++r)
                    for (int h = 0; h < 2; ++h)
                        *reinterpret_cast<float4*>(base + (size_t)(i * 16 + 4 * kg + r) * ldo + (4 * h + (l15 & 3)) * 16 + 4 * (l15 >> 2)) = make_float4(v, v, v, v);
        } else {
#pragma unroll 4
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(base + (size_t)(i * 16 + l15) * ldo + j * 16 + 4 * kg) = make_float4(v, v, v, v);
        }
    }
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
    // ---- the GEMM epilogue's store pattern: 256 KB tiles, 32 tiles (8 MiB) per workgroup
    for (int form = 0; form < 3; ++form)
        for (int G : {8, 32, 128, 256}) {
            const int tiles_per_wg = 16, ldo = 5120;                     // G * 16 tiles <= 262 * 20 tiles of a 67 080-row matrix; 2 GiB holds 104 857 rows
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                HIP(hipEventRecord(e0));
                if (form == 0) hipLaunchKernelGGL(tile_store_kernel<0>, dim3(G), dim3(256), 0, 0, (float*)a, ldo, tiles_per_wg);
                if (form == 1) hipLaunchKernelGGL(tile_store_kernel<1>, dim3(G), dim3(256), 0, 0, (float*)a, ldo, tiles_per_wg);
                if (form == 2) hipLaunchKernelGGL(tile_store_kernel<2>, dim3(G), dim3(256), 0, 0, (float*)a, ldo, tiles_per_wg);
                HIP(hipEventRecord(e1)); HIP(hipEventSynchronize(e1));
                float ms; HIP(hipEventElapsedTime(&ms, e0, e1));
                if (rep) best = ms < best ? ms : best;
            }
            const double gb = (double)G * tiles_per_wg * 256 * 256 * 4 / 1e9;
            printf("tile store, %-44s G=%4d: %7.3f ms  %7.1f GB/s = %5.1f B/clk/CU  (%.2f us per 256 KB tile)\n",
                   form == 0 ? "dword, 4 rows x 64 B per instruction" : form == 1 ? "dwordx4, 4 rows x 256 B per instruction" : "dwordx4 per lane, 16 rows x 64 B (swapped)",
                   G, best, gb / best * 1e3, gb / best * 1e3 / G / ghz, best * 1e3 / tiles_per_wg);
        }
    return 0;
}
