// Developer probe (not product): the operand / result layout and the E8M0 scale semantics of
// v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands, determined empirically on the device.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_f8_layout.hip -o tools/probe/mfma_f8_layout && tools/probe/mfma_f8_layout
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A: [32 rows][64 k] bytes, B: [32 cols][64 k] bytes (row-major over k), hypothesis: lane l holds row/col (l & 31), k = 32 (l >> 5) .. +31
__global__ void probe(const uint8_t* A, const uint8_t* B, float* D, int scale_a, int scale_b, int opsel_a, int hyp) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    const uint8_t* ap = A + (lane & 31) * 64 + 32 * (lane >> 5);
    const uint8_t* bp = B + (lane & 31) * 64 + 32 * (lane >> 5);
    for (int i = 0; i < 8; ++i) { int va, vb; memcpy(&va, ap + 4 * i, 4); memcpy(&vb, bp + 4 * i, 4); a[i] = va; b[i] = vb; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    if (opsel_a == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, scale_a, 0, scale_b);
    for (int r = 0; r < 16; ++r) D[lane * 16 + r] = c[r];
}

static uint8_t f2e4m3(float f) {   // small exact values only
    if (f == 0) return 0;
    uint8_t s = f < 0 ? 0x80 : 0; f = f < 0 ? -f : f;
    int e = 0; while (f >= 2) { f /= 2; ++e; } while (f < 1) { f *= 2; --e; }
    int m = (int)((f - 1) * 8 + 0.5f);
    return s | (uint8_t)((e + 7) << 3) | (uint8_t)m;
}

int main() {
    std::vector<uint8_t> A(32 * 64), B(32 * 64);
    std::vector<float> Af(32 * 64), Bf(32 * 64);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) {
        Af[i * 64 + k] = (float)(((i * 7 + k * 3) % 9) - 4) * 0.5f;      // small multiples of 0.5: exact in e4m3
        Bf[i * 64 + k] = (float)(((i * 5 + k * 11) % 7) - 3);
        A[i * 64 + k] = f2e4m3(Af[i * 64 + k]); B[i * 64 + k] = f2e4m3(Bf[i * 64 + k]);
    }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 64 * 16 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    std::vector<float> D(64 * 16);
    auto run = [&](int sa, int sb, int opsel) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, sa, sb, opsel, 0);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    };
    run(0x7f7f7f7f, 0x7f7f7f7f, 0);
    // reference C[i][j] = sum_k A[i][k] B[j][k]; hypothesis for D: lane l, reg r -> row i = (r & 3) + 8 (r >> 2) + 4 (l >> 5), col j = l & 31
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
        float ref = 0; for (int k = 0; k < 64; ++k) ref += Af[i * 64 + k] * Bf[j * 64 + k];
        if (ref != D[l * 16 + r]) ++bad;
    }
    printf("layout hypothesis (A row = lane&31, k = 32*(lane>>5)+byte; D as 32x32x16 bf16): %d mismatches of 1024\n", bad);
    if (bad) { printf("  sample: D[lane0][0..3] = %g %g %g %g\n", D[0], D[1], D[2], D[3]); }
    const float base = D[5 * 16 + 3];
    run(0x7f7f7f7a, 0x7f7f7f7f, 0);     // byte 0 of scale_a = 0x7a = 2^-5
    printf("scale_a byte0 = 0x7a (2^-5), opsel 0: ratio %g (expect 0.03125)\n", D[5 * 16 + 3] / base);
    run(0x7f7f7a7f, 0x7f7f7f7f, 1);     // byte 1, opsel 1
    printf("scale_a byte1 = 0x7a, opsel_a 1: ratio %g (expect 0.03125 if opsel selects the byte)\n", D[5 * 16 + 3] / base);
    run(0x7f7f7f7f, 0x7f7f7f7d, 0);
    printf("scale_b byte0 = 0x7d (2^-2): ratio %g (expect 0.25)\n", D[5 * 16 + 3] / base);
    return 0;
}
