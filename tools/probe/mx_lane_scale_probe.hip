// Probe: which lane supplies the E8M0 scale of which (row, 32-element k block) of v_mfma_scale_f32_32x32x64_f8f6f4.
// A = B = all ones; scale_a (or scale_b) differs per lane.  Own-lane hypothesis: D[i][j] = 32 * (2^sa(lane i) + 2^sa(lane 32 + i)) * ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* D, int which) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
    const unsigned var = 120u + (unsigned)(lane % 7) + ((lane >> 5) ? 8u : 0u), one = 127u;
    f32x16 c = {};
    if (which == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, (int)var, 0, (int)one);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, (int)one, 0, (int)var);
    for (int r = 0; r < 16; ++r) D[lane * 16 + r] = c[r];
}
int main() {
    float* dD; std::vector<float> D(1024);
    hipMalloc(&dD, 4096);
    for (int which = 0; which < 2; ++which) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dD, which);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        auto sc = [](int lane) { return ldexp(1.0, (120 + lane % 7 + ((lane >> 5) ? 8 : 0)) - 127); };
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r) {
            const int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);       // D[row][col]
            const int idx = which == 0 ? row : col;                                         // the operand whose scales vary: A rows / B cols
            const double want = 32.0 * (sc(idx) + sc(32 + idx));
            if (fabs(D[lane * 16 + r] - want) > 1e-6 * want) { if (bad < 4) printf("  which=%d D[%d][%d] = %g, own-lane hypothesis %g\n", which, row, col, D[lane * 16 + r], want); ++bad; }
        }
        printf("scale_%c per lane: own-lane hypothesis (lane = row-or-col + 32 * kblock): %d mismatches of 1024\n", which ? 'b' : 'a', bad);
    }
    return 0;
}
