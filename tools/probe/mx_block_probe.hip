// Probe: does a lane's E8M0 scale cover exactly that lane's 32 bytes?  A is zero except bytes [lo, lo + 16) of the lanes with hi = H
// (value 1.0), B all ones with unit scale; scale_a = 2^-3 on hi = 0 lanes and 2^0 on hi = 1 lanes.  D = 16 * (the scale that was applied).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* D, int H, int lo, unsigned s_lo, unsigned s_hi, unsigned s_b) {
    const int lane = threadIdx.x, hi = lane >> 5;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (hi == H && 4 * i >= lo && 4 * i < lo + 16) ? 0x38383838 : 0; b[i] = 0x38383838; }
    const unsigned sa = hi ? s_hi : s_lo;
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, (int)sa, 0, (int)s_b);
    D[lane] = c[0];          // (every lane stores: a store guarded by lane == 0 lets the compiler sink the MFMA into the branch)
}
int main() {
    float* dD; float h;
    hipMalloc(&dD, 256);
    for (int H = 0; H < 2; ++H) for (int lo = 0; lo < 32; lo += 16) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dD, H, lo, 0x7c7c7c7cu, 0x7f7f7f7fu, 0x7f7f7f7fu);
        hipMemcpy(&h, dD, 4, hipMemcpyDeviceToHost);
        printf("ones in bytes [%2d, %2d) of the hi = %d lanes: D = %g  -> scale applied = 2^%d (hi = 0 lanes carry 2^-3, hi = 1 lanes 2^0)\n", lo, lo + 16, H, h, h == 2.f ? -3 : (h == 16.f ? 0 : 99));
    }
    return 0;
}
