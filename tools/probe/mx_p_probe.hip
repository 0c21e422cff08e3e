// Probe (developer aid, not part of the library): the three hardware behaviours an fp8 P.V product would rest on --
//   1. v_cvt_scalef32_pk_fp8_f32: is the result fp8(src / scale) or fp8(src * scale); which bits of `scale` count; saturation
//   2. the scale-byte selection (opsel 0..3) of v_mfma_scale_f32_32x32x64_f8f6f4
//   3. buffer_load_dword ... lds (4-byte LDS-DMA): lane-linear dwords at the M0 base
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/mx_p_probe.hip -o tools/probe/mx_p_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

static float e4m3_to_f(unsigned char b) {
    const int e = (b >> 3) & 15, m = b & 7; const float v = e == 0 ? m / 512.f : ldexpf(1.f + m / 8.f, e - 7);
    return (b & 0x80) ? -v : v;
}

__global__ void cvt_kernel(const float* in, const float* scale, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    i16x2 v = {0, 0};
    v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(v, in[2 * i], in[2 * i + 1], scale[i], false);     // low half
    v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(v, in[2 * i + 1], in[2 * i], scale[i], true);      // high half (swapped inputs)
    out[i] = __builtin_bit_cast(unsigned, v);
}

template <int OPSEL>
__global__ void mfma_kernel(const unsigned char* A, const unsigned char* B, float* D, unsigned sa, unsigned sb) {
    const int lane = threadIdx.x, row = lane & 31, hi = lane >> 5;
    i32x8 a, b;
    memcpy(&a, A + row * 64 + hi * 32, 32);
    memcpy(&b, B + row * 64 + hi * 32, 32);
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPSEL, (int)sa, 0, (int)sb);
    for (int r = 0; r < 16; ++r) D[lane * 16 + r] = c[r];
}

__global__ void dma4_kernel(const unsigned* src, unsigned* out) {
    __shared__ unsigned lds[64];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
    // lane l fetches dword (63 - l): a lane-linear destination shows up as reversed contents
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 4, (63 - (int)threadIdx.x) * 4, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}

int main() {
    {   // 1
        const float in[] = {3.0f, -20.0f, 1000.f, 0.001f, 100.f, 7.f, 448.f, 500.f};
        const float sc[] = {4.0f, 1.0f, 5.0f, 0.25f};
        float *din, *dsc; unsigned* dout; unsigned out[4];
        hipMalloc(&din, sizeof in); hipMalloc(&dsc, sizeof sc); hipMalloc(&dout, sizeof out);
        hipMemcpy(din, in, sizeof in, hipMemcpyHostToDevice); hipMemcpy(dsc, sc, sizeof sc, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, din, dsc, dout, 4);
        hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
        for (int i = 0; i < 4; ++i)
            printf("cvt_scalef32_pk_fp8: (%g, %g) scale %g -> low half (%g, %g), high half [inputs swapped] (%g, %g)   raw 0x%08x\n", in[2 * i], in[2 * i + 1], sc[i],
                   e4m3_to_f(out[i] & 0xff), e4m3_to_f((out[i] >> 8) & 0xff), e4m3_to_f((out[i] >> 16) & 0xff), e4m3_to_f(out[i] >> 24), out[i]);
    }
    {   // 2
        std::vector<unsigned char> A(32 * 64, 0x38), B(32 * 64, 0x38);      // all ones: every D element = 64 * scale_a * scale_b
        unsigned char *dA, *dB; float* dD; std::vector<float> D(64 * 16);
        hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, D.size() * 4);
        hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
        const unsigned sa = 0x7f7e7d7cu, sb = 0x7f7f7f7fu;                 // bytes 0..3 = 2^-3, 2^-2, 2^-1, 2^0
        auto run = [&](auto k, int sel) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, sa, sb);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            printf("mfma scale_a = 0x%08x opsel_a %d: D[0] / 64 = %g (byte %d would give %g)\n", sa, sel, D[0] / 64.f, sel, ldexpf(1.f, sel - 3));
        };
        run(mfma_kernel<0>, 0); run(mfma_kernel<1>, 1); run(mfma_kernel<2>, 2); run(mfma_kernel<3>, 3);
    }
    {   // 3
        unsigned src[64], out[64]; for (int i = 0; i < 64; ++i) src[i] = 1000 + i;
        unsigned *ds, *dout; hipMalloc(&ds, sizeof src); hipMalloc(&dout, sizeof out);
        hipMemcpy(ds, src, sizeof src, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(dma4_kernel, dim3(1), dim3(64), 0, 0, ds, dout);
        hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
        printf("4-byte LDS-DMA: lds[0] = %u (lane 0 fetched 1063), lds[1] = %u, lds[63] = %u (lane 63 fetched 1000)\n", out[0], out[1], out[63]);
    }
    return 0;
}
