// Developer harness (NOT product, NOT the parity oracle): links libwan_hip.so through its C ABI,
// checks every kernel against naive double-precision host loops at small shapes and times the
// hot kernels at Wan2.1 shapes with HIP events.  Avoids the 1-2 min `import torch` on a fresh box.
//
//   hipcc -O2 -std=c++17 tools/kernel_check.cpp -Iinclude -Lvideocof_amd -lwan_hip -Wl,-rpath,'$ORIGIN/../videocof_amd' -o tools/kernel_check
//   tools/kernel_check [check|perf|all] [--big]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <string>
#include <vector>

#include "wan_hip.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define WAN(x) do { wan_status_t s_ = (x); if (s_ != WAN_OK) { printf("WAN error %d (%s) at %s:%d\n", s_, wan_last_error(), __FILE__, __LINE__); exit(3); } } while (0)

typedef uint16_t bf16;
static inline bf16 f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffff) > 0x7f800000) return 0x7fc0;
    u += 0x7fff + ((u >> 16) & 1);
    return (bf16)(u >> 16);
}
static inline float bf2f(bf16 b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

static std::mt19937 rng(1234);
static std::vector<float> randn(size_t n, float scale = 1.f) {
    std::normal_distribution<float> d(0.f, scale);
    std::vector<float> v(n);
    for (auto& x : v) x = d(rng);
    return v;
}
static std::vector<bf16> to_bf(const std::vector<float>& v) { std::vector<bf16> o(v.size()); for (size_t i = 0; i < v.size(); ++i) o[i] = f2bf(v[i]); return o; }
static std::vector<float> bf_round(const std::vector<float>& v) { std::vector<float> o(v.size()); for (size_t i = 0; i < v.size(); ++i) o[i] = bf2f(f2bf(v[i])); return o; }

template <typename T> struct Dev {
    T* p = nullptr; size_t n = 0;
    explicit Dev(size_t n_) : n(n_) { HIP(hipMalloc(&p, std::max<size_t>(n * sizeof(T), 16))); }
    explicit Dev(const std::vector<T>& h) : Dev(h.size()) { HIP(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice)); }
    ~Dev() { (void)hipFree(p); }
    std::vector<T> host() const { std::vector<T> h(n); HIP(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost)); return h; }
    void zero() { HIP(hipMemset(p, 0, n * sizeof(T))); }
};

static int g_fail = 0;
static void report(const char* name, double err, double tol, const char* metric = "rel_l2") {
    const bool ok = err <= tol && err == err;
    printf("  [%s] %-44s %s=%.3e (tol %.1e)\n", ok ? "PASS" : "FAIL", name, metric, err, tol);
    if (!ok) ++g_fail;
}
static double rel_l2(const std::vector<double>& ref, const std::vector<float>& got) {
    double num = 0, den = 0;
    for (size_t i = 0; i < ref.size(); ++i) { double d = got[i] - ref[i]; num += d * d; den += ref[i] * ref[i]; }
    return sqrt(num / std::max(den, 1e-300));
}
static std::vector<float> bf_to_f(const std::vector<bf16>& v) { std::vector<float> o(v.size()); for (size_t i = 0; i < v.size(); ++i) o[i] = bf2f(v[i]); return o; }

template <typename F> static double time_ms(F&& f, int iters = 5, int warm = 2) {
    hipEvent_t a, b; HIP(hipEventCreate(&a)); HIP(hipEventCreate(&b));
    for (int i = 0; i < warm; ++i) f();
    HIP(hipDeviceSynchronize());
    HIP(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) f();
    HIP(hipEventRecord(b, 0));
    HIP(hipEventSynchronize(b));
    float ms; HIP(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

// ------------------------------------------------------------------ checks
static void check_ln() {
    printf("wan_ln_modulate\n");
    for (int dim : {256, 1536, 5120}) {
        const int rows = 37, rpb = 19;   // 2 "samples" (19 + 18 rows)
        auto x = randn((size_t)rows * dim, 2.f); for (auto& v : x) v += 0.5f;
        auto sc = randn(2 * dim, 0.5f), sh = randn(2 * dim, 0.5f);
        Dev<float> dx(x), dsc(sc), dsh(sh); Dev<bf16> dout((size_t)rows * dim);
        for (int variant = 0; variant < 3; ++variant) {
            const bool mod = variant == 0, affine = variant == 1;
            WAN(wan_ln_modulate(dx.p, variant == 2 ? nullptr : dsc.p, variant == 2 ? nullptr : dsh.p, mod ? 1 : 0,
                                dout.p, rows, dim, affine ? rows : rpb, 1e-6f, nullptr));
            HIP(hipDeviceSynchronize());
            auto got = bf_to_f(dout.host());
            std::vector<double> ref((size_t)rows * dim);
            for (int r = 0; r < rows; ++r) {
                double mu = 0, var = 0;
                for (int c = 0; c < dim; ++c) mu += x[(size_t)r * dim + c];
                mu /= dim;
                for (int c = 0; c < dim; ++c) { double d = x[(size_t)r * dim + c] - mu; var += d * d; }
                var /= dim;
                const double rstd = 1.0 / sqrt(var + 1e-6);
                const int b = affine ? 0 : r / rpb;
                for (int c = 0; c < dim; ++c) {
                    double y = (x[(size_t)r * dim + c] - mu) * rstd;
                    if (variant != 2) y = y * ((mod ? 1.0 : 0.0) + sc[(size_t)b * dim + c]) + sh[(size_t)b * dim + c];
                    ref[(size_t)r * dim + c] = y;
                }
            }
            char nm[96]; snprintf(nm, sizeof nm, "dim=%d %s", dim, mod ? "modulate" : affine ? "affine" : "plain");
            report(nm, rel_l2(ref, got), 4e-3);
        }
    }
}

static void rope_tables(int head_dim, int max_pos, std::vector<float>& c, std::vector<float>& s) {
    const int half = head_dim / 2, d = head_dim;
    const int dims[3] = {d - 4 * (d / 6), 2 * (d / 6), 2 * (d / 6)};
    c.assign((size_t)max_pos * half, 0); s.assign((size_t)max_pos * half, 0);
    int col = 0;
    for (int ax = 0; ax < 3; ++ax)
        for (int i = 0; i < dims[ax] / 2; ++i, ++col) {
            const double inv = 1.0 / pow(10000.0, (2.0 * i) / dims[ax]);
            for (int p = 0; p < max_pos; ++p) { c[(size_t)p * half + col] = (float)cos(p * inv); s[(size_t)p * half + col] = (float)sin(p * inv); }
        }
}

static void check_rmsnorm_rope() {
    printf("wan_rmsnorm_rope\n");
    const int hd = 128, maxpos = 1024;
    std::vector<float> ct, st; rope_tables(hd, maxpos, ct, st);
    Dev<float> dct(ct), dst(st);
    for (int heads : {2, 12, 40}) {
        const int dim = heads * hd;
        const int F = 7, Hp = 3, Wp = 5, L = F * Hp * Wp;
        const int rows = L + 3;                         // 3 pad rows: normalised, not rotated
        const int64_t ld = 2 * dim;                     // q | k packed in one buffer
        for (int mode = 0; mode < 4; ++mode) {          // 3 = no rope
            auto x = bf_round(randn((size_t)rows * ld, 1.5f));
            auto wq = randn(dim, 0.2f), wk = randn(dim, 0.2f);
            for (auto& v : wq) v += 1.f; for (auto& v : wk) v += 1.f;
            Dev<bf16> dx(to_bf(x)); Dev<float> dwq(wq), dwk(wk);
            wan_rope_params rp = {F, Hp, Wp, mode % 3, 3, 4, 0, rows, maxpos};
            const float qs = (mode & 1) ? WAN_ATTN_QSCALE(0.0883883f) : 1.0f;   // x0 only
            WAN(wan_rmsnorm_rope(dx.p, dwq.p, dx.p + dim, dwk.p, ld, rows, dim, hd, 1e-6f,
                                 mode == 3 ? nullptr : dct.p, mode == 3 ? nullptr : dst.p, &rp, qs, nullptr));
            HIP(hipDeviceSynchronize());
            auto got = bf_to_f(dx.host());
            std::vector<double> ref(x.size());
            for (int r = 0; r < rows; ++r)
                for (int which = 0; which < 2; ++which) {
                    const float* xr = &x[(size_t)r * ld + which * dim];
                    const float* w = which ? wk.data() : wq.data();
                    double ss = 0; for (int c = 0; c < dim; ++c) ss += (double)xr[c] * xr[c];
                    const double rstd = 1.0 / sqrt(ss / dim + 1e-6);
                    const int f = r / (Hp * Wp), hh = (r / Wp) % Hp, ww = r % Wp;
                    int pt = f;
                    if (mode == 1) pt = f < 3 ? f : f - 3;
                    if (mode == 2) pt = f < 3 ? f + 1 : (f < 4 ? 0 : f - 4 + 1);
                    for (int c = 0; c < dim; c += 2) {
                        const double sc = which == 0 ? qs : 1.0;
                        double a = xr[c] * rstd * w[c] * sc, b = xr[c + 1] * rstd * w[c + 1] * sc;
                        if (mode != 3 && r < L) {
                            const int p = (c % hd) / 2;
                            const int pos = p < 22 ? pt : (p < 43 ? hh : ww);
                            const double cs = ct[(size_t)pos * 64 + p], sn = st[(size_t)pos * 64 + p];
                            const double ra = a * cs - b * sn, rb = a * sn + b * cs; a = ra; b = rb;
                        }
                        ref[(size_t)r * ld + which * dim + c] = a; ref[(size_t)r * ld + which * dim + c + 1] = b;
                    }
                }
            char nm[96]; snprintf(nm, sizeof nm, "heads=%d mode=%d", heads, mode);
            report(nm, rel_l2(ref, got), 4e-3);
        }
    }
}

static void check_gemm() {
  for (const char* var : {"1", "2", "2w"}) {       // 1 = 128^2, 2 = 256^2 8-wave phased, 2w = 256^2 4-wave (K % 128 == 0 shapes)
    WAN(wan_set_tuning("gemm_variant", var[0] == '1' ? 1 : 2));
    WAN(wan_set_tuning("gemm_w4", strlen(var) > 1 ? 3 : 0));
    printf("wan_gemm_bf16 variant %s\n", var);
    struct Shape { int M, N, K; };
    for (Shape sh : {Shape{300, 384, 256}, Shape{128, 128, 64}, Shape{515, 64, 1024}, Shape{77, 1536, 192}, Shape{1100, 520, 448}, Shape{1100, 520, 512}}) {
        const int M = sh.M, N = sh.N, K = sh.K;
        auto A = bf_round(randn((size_t)M * K)), W = bf_round(randn((size_t)N * K, 0.1f));
        auto bias = randn(N, 0.5f);
        Dev<bf16> dA(to_bf(A)), dW(to_bf(W)); Dev<float> dB(bias);
        std::vector<double> acc((size_t)M * N);
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)A[(size_t)m * K + k] * W[(size_t)n * K + k];
                acc[(size_t)m * N + n] = s + bias[n];
            }
        char nm[96];
        {   // bf16
            Dev<bf16> out((size_t)M * N);
            WAN(wan_gemm_bf16(dA.p, K, dW.p, K, dB.p, out.p, N, M, N, K, WAN_EPI_BF16, nullptr, 0, nullptr));
            HIP(hipDeviceSynchronize());
            snprintf(nm, sizeof nm, "%dx%dx%d bf16", M, N, K); report(nm, rel_l2(acc, bf_to_f(out.host())), 4e-3);
        }
        {   // gelu
            Dev<bf16> out((size_t)M * N);
            WAN(wan_gemm_bf16(dA.p, K, dW.p, K, dB.p, out.p, N, M, N, K, WAN_EPI_GELU_BF16, nullptr, 0, nullptr));
            HIP(hipDeviceSynchronize());
            std::vector<double> ref(acc.size());
            for (size_t i = 0; i < acc.size(); ++i) { double x = acc[i]; ref[i] = 0.5 * x * (1 + tanh(0.7978845608028654 * (x + 0.044715 * x * x * x))); }
            snprintf(nm, sizeof nm, "%dx%dx%d gelu", M, N, K); report(nm, rel_l2(ref, bf_to_f(out.host())), 4e-3);
        }
        {   // f32
            Dev<float> out((size_t)M * N);
            WAN(wan_gemm_bf16(dA.p, K, dW.p, K, dB.p, out.p, N, M, N, K, WAN_EPI_F32, nullptr, 0, nullptr));
            HIP(hipDeviceSynchronize());
            snprintf(nm, sizeof nm, "%dx%dx%d f32", M, N, K); report(nm, rel_l2(acc, out.host()), 1e-5);
        }
        {   // residual + gate, two samples
            const int rpb = (M + 1) / 2;
            auto x0 = randn((size_t)M * N), gate = randn(2 * N);
            Dev<float> out(x0), dG(gate);
            WAN(wan_gemm_bf16(dA.p, K, dW.p, K, dB.p, out.p, N, M, N, K, WAN_EPI_RESID_F32, dG.p, rpb, nullptr));
            HIP(hipDeviceSynchronize());
            std::vector<double> ref(acc.size());
            for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n)
                ref[(size_t)m * N + n] = x0[(size_t)m * N + n] + acc[(size_t)m * N + n] * gate[(size_t)(m / rpb) * N + n];
            snprintf(nm, sizeof nm, "%dx%dx%d resid*gate", M, N, K); report(nm, rel_l2(ref, out.host()), 1e-5);
            Dev<float> out2(x0);
            WAN(wan_gemm_bf16(dA.p, K, dW.p, K, dB.p, out2.p, N, M, N, K, WAN_EPI_RESID_F32, nullptr, 0, nullptr));
            HIP(hipDeviceSynchronize());
            for (size_t i = 0; i < ref.size(); ++i) ref[i] = x0[i] + acc[i];
            snprintf(nm, sizeof nm, "%dx%dx%d resid", M, N, K); report(nm, rel_l2(ref, out2.host()), 1e-5);
        }
        {   // transposed
            const int ldo = (M + 63) / 64 * 64;
            Dev<bf16> out((size_t)N * ldo); out.zero();
            WAN(wan_gemm_bf16(dA.p, K, dW.p, K, dB.p, out.p, ldo, M, N, K, WAN_EPI_BF16_T, nullptr, 0, nullptr));
            HIP(hipDeviceSynchronize());
            auto h = bf_to_f(out.host());
            std::vector<float> got((size_t)M * N); double padmax = 0;
            for (int n = 0; n < N; ++n) for (int m = 0; m < ldo; ++m) {
                if (m < M) got[(size_t)m * N + n] = h[(size_t)n * ldo + m]; else padmax = std::max(padmax, (double)fabs(h[(size_t)n * ldo + m]));
            }
            snprintf(nm, sizeof nm, "%dx%dx%d transposed", M, N, K); report(nm, rel_l2(acc, got), 4e-3);
            report("   pad columns untouched", padmax, 0.0, "max_abs");
        }
    }
  }
  unsetenv("WAN_GEMM_VARIANT"); unsetenv("WAN_GEMM_PHASES");
}

static void attn_ref(const std::vector<float>& q, const std::vector<float>& k, const std::vector<float>& v,
                     int Lq, int Lk, int H, float scale, std::vector<double>& out, const std::vector<int>& rows) {
    const int D = 128, C = H * D;
    out.assign(rows.size() * C, 0);
    std::vector<double> s(Lk);
    for (int h = 0; h < H; ++h)
        for (size_t ri = 0; ri < rows.size(); ++ri) {
            const int i = rows[ri];
            double mx = -1e300;
            for (int j = 0; j < Lk; ++j) {
                double d = 0;
                for (int c = 0; c < D; ++c) d += (double)q[(size_t)i * C + h * D + c] * k[(size_t)j * C + h * D + c];
                s[j] = d * scale; mx = std::max(mx, s[j]);
            }
            double den = 0;
            for (int j = 0; j < Lk; ++j) { s[j] = exp(s[j] - mx); den += s[j]; }
            for (int j = 0; j < Lk; ++j) {
                const double p = s[j] / den;
                for (int c = 0; c < D; ++c) out[ri * C + h * D + c] += p * v[(size_t)j * C + h * D + c];
            }
        }
}

static void check_attn() {
  for (const char* var : {"2", "3"}) {      // 2 = plain q, 3 = WAN_ATTN_Q_PRESCALED
    const bool pre = var[0] == '3';
    printf("wan_attention_fwd variant %s (+ wan_transpose_bf16)\n", var);
    struct Shape { int Lq, Lk, H; float qs; };
    // qscale 40 / 100: log2-domain scores of +-300 .. +-1000 -- far outside any fixed window, tile maxima that differ by > 100:
    // the lazy softmax reference of the 4-wave kernel has to repair (and exp2 overflows to Inf before it does)
    for (Shape sh : {Shape{300, 420, 2, 1.f}, Shape{64, 64, 1, 1.f}, Shape{257, 8, 3, 1.f}, Shape{520, 512, 2, 3.f}, Shape{33, 1000, 1, 6.f},
                     Shape{520, 1500, 2, 40.f}, Shape{256, 4096, 1, 100.f}, Shape{86 * 256 + 10, 1100, 3, 1.f}}) {
        const int Lq = sh.Lq, Lk = sh.Lk, H = sh.H, C = H * 128;
        auto q = bf_round(randn((size_t)Lq * C, sh.qs)), k = bf_round(randn((size_t)Lk * C)), v = bf_round(randn((size_t)Lk * C));
        // make V asymmetric across d and key so a transposed/permuted read cannot pass
        for (int j = 0; j < Lk; ++j) for (int c = 0; c < C; ++c) v[(size_t)j * C + c] = bf2f(f2bf(v[(size_t)j * C + c] + 0.01f * (c % 128) - 0.003f * (j % 97)));
        const int64_t ldvt = (Lk + 63) / 64 * 64;
        Dev<bf16> dq(to_bf(q)), dk(to_bf(k)), dv(to_bf(v)), dvt((size_t)C * ldvt), dout((size_t)Lq * C);
        HIP(hipMemset(dvt.p, 0xff, dvt.n * 2));   // poison: transpose must zero the pad
        WAN(wan_transpose_bf16(dv.p, C, dvt.p, ldvt, Lk, C, nullptr));
        const float scale = 1.f / sqrtf(128.f);
        if (pre) {   // q handed over pre-multiplied by scale*log2(e); the reference sees the same rounded values
            const float cc = WAN_ATTN_QSCALE(scale);
            for (auto& x : q) x = bf2f(f2bf(x * cc));
            HIP(hipMemcpy(dq.p, to_bf(q).data(), q.size() * 2, hipMemcpyHostToDevice));
            for (auto& x : q) x /= cc;
        }
        const int64_t wsb = wan_attention_workspace_bytes(1, Lq, Lk, H, 128);
        Dev<char> ws((size_t)std::max<int64_t>(wsb, 16)); ws.zero();      // the header of the scratch must start as zero
        WAN(wan_attention_fwd(dq.p, C, 0, dk.p, C, 0, dvt.p, ldvt, 0, dout.p, C, 0, 1, Lq, Lk, H, 128, scale, pre ? WAN_ATTN_Q_PRESCALED : 0,
                              wsb ? ws.p : nullptr, wsb, nullptr));
        if (wsb) printf("  (tail split active: workspace %lld B)\n", (long long)wsb);
        HIP(hipDeviceSynchronize());
        // reference on every row for small shapes; on the head, the 4-wave tail region and a stride for large ones
        std::vector<int> rows;
        for (int i = 0; i < Lq; ++i) if (Lq <= 1024 || i < 64 || i >= Lq - 600 || i % 211 == 0) rows.push_back(i);
        std::vector<double> ref; attn_ref(q, k, v, Lq, Lk, H, scale, ref, rows);
        auto all = bf_to_f(dout.host());
        std::vector<float> got(rows.size() * C);
        for (size_t ri = 0; ri < rows.size(); ++ri) memcpy(&got[ri * C], &all[(size_t)rows[ri] * C], C * sizeof(float));
        double maxabs = 0; for (size_t i = 0; i < ref.size(); ++i) maxabs = std::max(maxabs, fabs(got[i] - ref[i]));
        char nm[96]; snprintf(nm, sizeof nm, "Lq=%d Lk=%d H=%d qscale=%.0f", Lq, Lk, H, sh.qs);
        report(nm, rel_l2(ref, got), 6e-3);
        report("   max abs err", maxabs, 3.2e-2, "max_abs");      // outputs reach |x| in [4, 8): one bf16 ulp there is 2^-5 = 3.125e-2
    }
  }
}

static void check_layout() {
    printf("wan_patchify / wan_unpatchify\n");
    const int Cin = 16, F = 3, H = 8, W = 12, Fp = 3, Hp = 4, Wp = 6, L = Fp * Hp * Wp;
    auto x = randn((size_t)Cin * F * H * W);
    Dev<float> dx(x); Dev<bf16> tok((size_t)L * 64);
    WAN(wan_patchify(dx.p, 0, tok.p, 64, Cin, F, H, W, 1, 2, 2, nullptr));
    HIP(hipDeviceSynchronize());
    auto got = bf_to_f(tok.host());
    std::vector<double> ref((size_t)L * 64);
    for (int f = 0; f < Fp; ++f) for (int h = 0; h < Hp; ++h) for (int w = 0; w < Wp; ++w)
        for (int c = 0; c < Cin; ++c) for (int b = 0; b < 2; ++b) for (int d = 0; d < 2; ++d)
            ref[(size_t)((f * Hp + h) * Wp + w) * 64 + c * 4 + b * 2 + d] = bf2f(f2bf(x[(((size_t)c * F + f) * H + 2 * h + b) * W + 2 * w + d]));
    report("patchify fp32 in", rel_l2(ref, got), 0.0);
    auto t = randn((size_t)L * 64);
    Dev<float> dt(t), dout((size_t)16 * F * H * W);
    WAN(wan_unpatchify(dt.p, 64, dout.p, 0, 16, Fp, Hp, Wp, 1, 2, 2, 0, nullptr));
    HIP(hipDeviceSynchronize());
    auto g2 = dout.host();
    std::vector<double> r2(g2.size());
    for (int c = 0; c < 16; ++c) for (int f = 0; f < F; ++f) for (int h = 0; h < H; ++h) for (int w = 0; w < W; ++w)
        r2[(((size_t)c * F + f) * H + h) * W + w] = t[(size_t)((f * Hp + h / 2) * Wp + w / 2) * 64 + ((h % 2) * 2 + (w % 2)) * 16 + c];
    report("unpatchify fp32 out", rel_l2(r2, g2), 0.0);
}

// ------------------------------------------------------------------ perf
static void perf(bool big, bool attn_only = false, bool gemm_only = false, bool rows_only = false) {
    hipDeviceProp_t prop; HIP(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs, clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    const int L = big ? 67080 : 32760;
    if (!attn_only && !gemm_only) {   // LN-modulate + rmsnorm/rope, 14B width
        const int C = 5120;
        Dev<float> x((size_t)L * C), sc(C), sh(C); Dev<bf16> o((size_t)L * C);
        HIP(hipMemset(x.p, 0, x.n * 4)); sc.zero(); sh.zero();
        double ms = time_ms([&] { WAN(wan_ln_modulate(x.p, sc.p, sh.p, 1, o.p, L, C, L, 1e-6f, nullptr)); });
        printf("  ln_modulate      L=%d C=%d: %.3f ms  %.0f GB/s (6C B/row)\n", L, C, ms, 6.0 * C * L / ms / 1e6);
        std::vector<float> ct, st; rope_tables(128, 1024, ct, st);
        Dev<float> dct(ct), dst(st), w(C);
        Dev<bf16> qk((size_t)L * 2 * C); qk.zero(); w.zero();
        wan_rope_params rp = {43, 30, 52, 2, 21, 22, 0, L, 1024};
        if (!big) { rp.F = 21; rp.mode = 0; }
        ms = time_ms([&] { WAN(wan_rmsnorm_rope(qk.p, w.p, qk.p + C, w.p, 2 * C, L, C, 128, 1e-6f, dct.p, dst.p, &rp, 1.0f, nullptr)); });
        printf("  rmsnorm_rope q+k L=%d C=%d: %.3f ms  %.0f GB/s (8C B/row)\n", L, C, ms, 8.0 * C * L / ms / 1e6);
    }
    if (rows_only) return;
    struct G { int M, N, K; int epi; const char* what; };
    std::vector<G> gs = {{8392, 5120, 5120, WAN_EPI_BF16, "14B o/q proj, SP8 shard"}, {8392, 13824, 5120, WAN_EPI_GELU_BF16, "14B ffn.0, SP8 shard"},
                         {8392, 5120, 13824, WAN_EPI_RESID_F32, "14B ffn.2, SP8 shard"}, {16776, 5120, 5120, WAN_EPI_BF16, "14B o/q proj, SP4 shard"},
                         {L, 5120, 5120, WAN_EPI_BF16, "14B o/q proj"}, {L, 10240, 5120, WAN_EPI_BF16, "14B qk proj"},
                         {L, 13824, 5120, WAN_EPI_GELU_BF16, "14B ffn.0+gelu"}, {L, 5120, 13824, WAN_EPI_RESID_F32, "14B ffn.2+resid"},
                         {L, 5120, 5120, WAN_EPI_BF16_T, "14B v proj (T)"}, {L, 1536, 1536, WAN_EPI_BF16, "1.3B proj"},
                         {L, 8960, 1536, WAN_EPI_GELU_BF16, "1.3B ffn.0"}};
    if (attn_only) gs.clear();
    for (auto g : gs) {
        auto hA = to_bf(randn((size_t)4096 * 64));   // fill with random data (tile the pattern)
        Dev<bf16> A((size_t)g.M * g.K), W((size_t)g.N * g.K);
        for (size_t off = 0; off < A.n; off += hA.size()) HIP(hipMemcpy(A.p + off, hA.data(), std::min(hA.size(), A.n - off) * 2, hipMemcpyHostToDevice));
        for (size_t off = 0; off < W.n; off += hA.size()) HIP(hipMemcpy(W.p + off, hA.data(), std::min(hA.size(), W.n - off) * 2, hipMemcpyHostToDevice));
        Dev<float> bias(g.N), gate(g.N); bias.zero(); gate.zero();
        const int64_t ldo = g.epi == WAN_EPI_BF16_T ? (g.M + 63) / 64 * 64 : g.N;
        const size_t osz = (size_t)(g.epi == WAN_EPI_BF16_T ? g.N : g.M) * ldo * ((g.epi == WAN_EPI_F32 || g.epi == WAN_EPI_RESID_F32) ? 4 : 2);
        Dev<char> out(osz); out.zero();
        for (int round = 0; round < 2; ++round)
        for (const char* var : {"1", "2", "2w"}) {           // 1 = 128^2 kernel, 2 = 256^2 8-wave phased, 2w = 256^2 4-wave
            WAN(wan_set_tuning("gemm_variant", var[0] == '1' ? 1 : 2));
            WAN(wan_set_tuning("gemm_w4", strlen(var) > 1 ? 3 : 0));
            double ms = time_ms([&] { WAN(wan_gemm_bf16(A.p, g.K, W.p, g.K, bias.p, out.p, ldo, g.M, g.N, g.K, g.epi,
                                                        g.epi == WAN_EPI_RESID_F32 ? gate.p : nullptr, g.M, nullptr)); }, 3, 1);
            printf("  gemm[v%s] %-18s M=%d N=%d K=%d: %.3f ms  %.0f TFLOP/s\n", var, g.what, g.M, g.N, g.K, ms, 2.0 * g.M * g.N * g.K / ms / 1e9);
        }
        WAN(wan_set_tuning("gemm_variant", 0)); WAN(wan_set_tuning("gemm_w4", 1));
    }
    struct A_ { int Lq, Lk, H; const char* what; };
    if (gemm_only) return;
    std::vector<A_> as = {{8192, 8192, 40, "self L=8k H=40"}, {L, L, big ? 40 : 12, "self full"}, {L, 512, 40, "cross Lk=512"},
                          {L, L, 5, "self, SP8 shard (5 heads)"}, {L, L, 10, "self, SP4 shard (10 heads)"}, {L, L, 20, "self, SP2 shard"}};
    for (auto s : as) {
        const int C = s.H * 128; const int64_t ldvt = (s.Lk + 63) / 64 * 64;
        auto hq = to_bf(randn((size_t)4096 * 128));
        Dev<bf16> q((size_t)s.Lq * C), k((size_t)s.Lk * C), vt((size_t)C * ldvt), o((size_t)s.Lq * C);
        auto fill = [&](Dev<bf16>& d) { for (size_t off = 0; off < d.n; off += hq.size()) HIP(hipMemcpy(d.p + off, hq.data(), std::min(hq.size(), d.n - off) * 2, hipMemcpyHostToDevice)); };
        fill(q); fill(k); fill(vt);
        // pre-scaled variants get q * softmax_scale * log2(e), as wan_rmsnorm_rope(x0_scale) hands it over
        Dev<bf16> qs((size_t)s.Lq * C);
        { std::vector<float> hf = bf_to_f(hq); for (auto& x : hf) x *= WAN_ATTN_QSCALE(0.0883883f); auto hs = to_bf(hf);
          for (size_t off = 0; off < qs.n; off += hs.size()) HIP(hipMemcpy(qs.p + off, hs.data(), std::min(hs.size(), qs.n - off) * 2, hipMemcpyHostToDevice)); }
        for (int round = 0; round < (attn_only ? 2 : 1); ++round)
        for (const char* var : {"3s", "3n", "3"}) {       // 3 = pre-scaled q (max-free fast kernel + fix-up); 3s = running-max kernel only; 3n = 3 without the split tail
            if (!attn_only && strcmp(var, "3")) continue;
            WAN(wan_set_tuning("attn_tail", !strcmp(var, "3n") ? 0 : 1));
            WAN(wan_set_tuning("attn_fast", !strcmp(var, "3s") ? 0 : 1));
            const int64_t wsb = wan_attention_workspace_bytes(1, s.Lq, s.Lk, s.H, 128);
            Dev<char> ws((size_t)std::max<int64_t>(wsb, 16)); ws.zero();      // the header of the scratch must start as zero
            double ms = time_ms([&] { WAN(wan_attention_fwd(var[0] == '3' ? qs.p : q.p, C, 0, k.p, C, 0, vt.p, ldvt, 0, o.p, C, 0, 1, s.Lq, s.Lk, s.H, 128, 0.0883883f, var[0] == '3' ? WAN_ATTN_Q_PRESCALED : 0,
                                                            wsb ? ws.p : nullptr, wsb, nullptr)); }, 3, 1);
            printf("  attn[v%s] %-18s Lq=%d Lk=%d H=%d: %.3f ms  %.0f TFLOP/s\n", var, s.what, s.Lq, s.Lk, s.H, ms, 4.0 * s.Lq * s.Lk * C / ms / 1e9);
        }
        WAN(wan_set_tuning("attn_tail", 1)); WAN(wan_set_tuning("attn_fast", 1));
    }
}

int main(int argc, char** argv) {
    std::string mode = argc > 1 ? argv[1] : "all";
    bool big = false;
    for (int i = 1; i < argc; ++i) if (!strcmp(argv[i], "--big")) big = true;
    printf("libwan_hip ABI %d\n", wan_abi_version());
    if (mode == "check" || mode == "all") {
        check_ln(); check_rmsnorm_rope(); check_gemm(); check_attn(); check_layout();
        printf("%s (%d failures)\n", g_fail ? "CHECK FAILED" : "ALL CHECKS PASSED", g_fail);
    }
    if (mode == "rows") {         // the HBM-bound row kernels alone: numerics, then the 14B-width timings at L = 67 080 (three rounds)
        check_ln(); check_rmsnorm_rope();
        printf("%s (%d failures)\n", g_fail ? "CHECK FAILED" : "ALL CHECKS PASSED", g_fail);
        for (int r = 0; r < 3; ++r) perf(true, false, false, true);
    }
    if (mode == "gemm") { check_gemm(); printf("%s (%d failures)\n", g_fail ? "CHECK FAILED" : "ALL CHECKS PASSED", g_fail); perf(big, false, true); }
    if (mode == "attn") { check_attn(); printf("%s (%d failures)\n", g_fail ? "CHECK FAILED" : "ALL CHECKS PASSED", g_fail); perf(big, true); }
    if (mode == "gemmx") {        // timing attribution of the 4-wave GEMM's main loop: with / without its LDS-DMA instructions
        if (wan_get_tuning("dev_experiments") != 1) printf("  NOTE: libwan_hip.so was built without the experiment variants (make clean; make EXPERIMENTS=1): every arm below is the product kernel\n");
        const int M = 67080;
        struct S { int N, K, epi; const char* what; };
        for (S sh : {S{10240, 5120, WAN_EPI_BF16, "qk proj"}, S{5120, 13824, WAN_EPI_RESID_F32, "ffn.2+resid"}}) {
            auto hA = to_bf(randn((size_t)4096 * 64));
            Dev<bf16> A((size_t)M * sh.K), W((size_t)sh.N * sh.K);
            for (size_t off = 0; off < A.n; off += hA.size()) HIP(hipMemcpy(A.p + off, hA.data(), std::min(hA.size(), A.n - off) * 2, hipMemcpyHostToDevice));
            for (size_t off = 0; off < W.n; off += hA.size()) HIP(hipMemcpy(W.p + off, hA.data(), std::min(hA.size(), W.n - off) * 2, hipMemcpyHostToDevice));
            Dev<float> bias(sh.N), gate(sh.N); bias.zero(); gate.zero();
            Dev<char> out((size_t)M * sh.N * 4); out.zero();
            for (int round = 0; round < 2; ++round)
                for (int e : {0, 16, 4, 1, 9, 3}) {
                    WAN(wan_set_tuning("gemm_exp", e));
                    double ms = time_ms([&] { WAN(wan_gemm_bf16(A.p, sh.K, W.p, sh.K, bias.p, out.p, sh.N, M, sh.N, sh.K, sh.epi,
                                                                sh.epi == WAN_EPI_RESID_F32 ? gate.p : nullptr, M, nullptr)); }, 3, 1);
                    printf("  gemmx[%-12s gemm_exp=%d (%s)] %.3f ms  %.0f TFLOP/s\n", sh.what, e, e == 0 ? "product" : e == 16 ? "W tile requested one k-step earlier" : e == 4 ? "all DMA, cache-resident source" : e == 1 ? "no W DMA" : e == 9 ? "W bytes by register loads, discarded" : "no DMA at all",
                           ms, 2.0 * M * sh.N * sh.K / ms / 1e9);
                }
            WAN(wan_set_tuning("gemm_exp", 0));
        }
    }
    if (mode == "gemmpk") {       // the persistent stream-K GEMM (wan_gemm_bf16_ws): numerics vs the one-tile-per-workgroup kernels, repro, in-process A/B
        WAN(wan_set_tuning("gemm_pk", 2));
        for (int i = 2; i < argc; ++i) {                       // extra "key=value" tuning arguments
            const char* eq = strchr(argv[i], '=');
            if (eq && argv[i][0] != '-') { std::string k(argv[i], eq - argv[i]); WAN(wan_set_tuning(k.c_str(), atoi(eq + 1))); printf("tuning %s=%d\n", k.c_str(), atoi(eq + 1)); }
        }
        struct Shape { int M, N, K; };
        const int epis[5] = {WAN_EPI_BF16, WAN_EPI_GELU_BF16, WAN_EPI_F32, WAN_EPI_RESID_F32, WAN_EPI_BF16_T};
        const char* en[5] = {"bf16", "gelu", "f32", "resid*gate", "transposed"};
        for (Shape sh : {Shape{1100, 520, 512}, Shape{2304, 1536, 896}, Shape{700, 1300, 1024}, Shape{4100, 2100, 256}, Shape{9000, 5120, 640}, Shape{16776, 5120, 128}}) {
            const int M = sh.M, N = sh.N, K = sh.K;
            auto hA = to_bf(randn((size_t)M * K)), hW = to_bf(randn((size_t)N * K, 0.1f));
            Dev<bf16> dA(hA), dW(hW); Dev<float> dB(randn(N, 0.5f)), dG(randn(2 * (size_t)N));
            const int64_t wsb = wan_gemm_workspace_bytes(M, N, K);
            Dev<char> ws((size_t)std::max<int64_t>(wsb, 16));
            HIP(hipMemset(ws.p, 0xff, ws.n));                  // garbage in the workspace: the kernel must not depend on its contents
            printf("M=%d N=%d K=%d: plan %d, workspace %lld bytes\n", M, N, K, wan_gemm_ws_plan(M, N, K), (long long)wsb);
            const auto x0 = randn((size_t)M * N);
            for (int e = 0; e < 5; ++e) {
                const bool f32o = epis[e] == WAN_EPI_F32 || epis[e] == WAN_EPI_RESID_F32;
                const int64_t ldo = epis[e] == WAN_EPI_BF16_T ? (M + 63) / 64 * 64 : N;
                const size_t on = (size_t)(epis[e] == WAN_EPI_BF16_T ? N : M) * ldo;
                std::vector<float> res[4];
                for (int arm = 0; arm < 4; ++arm) {            // 0: reference kernels (no workspace), 1 and 2: persistent (product form), twice; 3: the round-4 epilogues
                    WAN(wan_set_tuning("gemm_pk_form", arm == 3 ? 0 : 31));
                    Dev<char> out(on * (f32o ? 4 : 2));
                    if (epis[e] == WAN_EPI_RESID_F32) HIP(hipMemcpy(out.p, x0.data(), on * 4, hipMemcpyHostToDevice)); else out.zero();
                    const float* gate = epis[e] == WAN_EPI_RESID_F32 ? dG.p : nullptr;
                    const int64_t rpb = (M + 1) / 2;
                    if (arm == 0) WAN(wan_gemm_bf16(dA.p, K, dW.p, K, dB.p, out.p, ldo, M, N, K, epis[e], gate, rpb, nullptr));
                    else WAN(wan_gemm_bf16_ws(dA.p, K, dW.p, K, dB.p, out.p, ldo, M, N, K, epis[e], gate, rpb, ws.p, wsb, nullptr));
                    HIP(hipDeviceSynchronize());
                    res[arm].resize(on);
                    if (f32o) HIP(hipMemcpy(res[arm].data(), out.p, on * 4, hipMemcpyDeviceToHost));
                    else { std::vector<bf16> h(on); HIP(hipMemcpy(h.data(), out.p, on * 2, hipMemcpyDeviceToHost)); for (size_t i = 0; i < on; ++i) res[arm][i] = bf2f(h[i]); }
                }
                std::vector<double> ref(res[0].begin(), res[0].end());
                char nm[96];
                snprintf(nm, sizeof nm, "%s: persistent vs per-tile kernel", en[e]); report(nm, rel_l2(ref, res[1]), f32o ? 2e-6 : 2e-3);
                snprintf(nm, sizeof nm, "%s: persistent, run to run", en[e]); report(nm, res[1] == res[2] ? 0.0 : 1.0, 0.0, "mismatch");
                snprintf(nm, sizeof nm, "%s: row-permuted epilogue vs round-4 epilogue (same MFMA chains: bitwise)", en[e]); report(nm, res[1] == res[3] ? 0.0 : 1.0, 0.0, "mismatch");
            }
        }
        printf("%s (%d failures)\n", g_fail ? "CHECK FAILED" : "ALL CHECKS PASSED", g_fail);
        // ---- in-process A/B on the 14B and 1.3B shapes: the persistent kernel with the round-4 epilogues (form 0) vs the row-permuted ones (form 1)
        WAN(wan_set_tuning("gemm_pk", 1));
        WAN(wan_set_tuning("gemm_pk_form", 31));
        const int L = 67080;
        struct G { int M, N, K; int epi; const char* what; };
        std::vector<G> gs = {{L, 5120, 5120, WAN_EPI_BF16, "o/q proj"}, {L, 10240, 5120, WAN_EPI_BF16, "qk proj"},
                             {L, 5120, 5120, WAN_EPI_BF16_T, "v proj (T)"}, {L, 5120, 5120, WAN_EPI_RESID_F32, "o proj+gate+resid"},
                             {L, 13824, 5120, WAN_EPI_GELU_BF16, "ffn.0+gelu"}, {L, 5120, 13824, WAN_EPI_RESID_F32, "ffn.2+resid"},
                             {8392, 5120, 5120, WAN_EPI_BF16, "SP8 o/q proj"}, {8392, 10240, 5120, WAN_EPI_BF16, "SP8 qk proj"},
                             {8392, 13824, 5120, WAN_EPI_GELU_BF16, "SP8 ffn.0"}, {8392, 5120, 13824, WAN_EPI_RESID_F32, "SP8 ffn.2"},
                             {L, 3072, 1536, WAN_EPI_BF16, "1.3B qk proj"}, {L, 1536, 1536, WAN_EPI_BF16_T, "1.3B v proj (T)"},
                             {L, 1536, 1536, WAN_EPI_RESID_F32, "1.3B o+gate+resid"}, {L, 1536, 1536, WAN_EPI_BF16, "1.3B cross q"},
                             {L, 8960, 1536, WAN_EPI_GELU_BF16, "1.3B ffn.0+gelu"}, {L, 1536, 8960, WAN_EPI_RESID_F32, "1.3B ffn.2+resid"}};
        for (auto g : gs) {
            auto hA = to_bf(randn((size_t)4096 * 64));
            Dev<bf16> A((size_t)g.M * g.K), W((size_t)g.N * g.K);
            for (size_t off = 0; off < A.n; off += hA.size()) HIP(hipMemcpy(A.p + off, hA.data(), std::min(hA.size(), A.n - off) * 2, hipMemcpyHostToDevice));
            for (size_t off = 0; off < W.n; off += hA.size()) HIP(hipMemcpy(W.p + off, hA.data(), std::min(hA.size(), W.n - off) * 2, hipMemcpyHostToDevice));
            Dev<float> bias(g.N), gate(g.N); bias.zero(); gate.zero();
            const int64_t ldo = g.epi == WAN_EPI_BF16_T ? (g.M + 63) / 64 * 64 : g.N;
            const size_t osz = (size_t)(g.epi == WAN_EPI_BF16_T ? g.N : g.M) * ldo * ((g.epi == WAN_EPI_F32 || g.epi == WAN_EPI_RESID_F32) ? 4 : 2);
            Dev<char> out(osz); out.zero();
            const int64_t wsb = wan_gemm_workspace_bytes(g.M, g.N, g.K);
            Dev<char> ws((size_t)std::max<int64_t>(wsb, 16)); ws.zero();
            if (wan_get_tuning("gemm_exp") & 64) {          // experiment builds: where a workgroup's cycles go (s_memtime stamps, thread 0 of workers 0..127)
                WAN(wan_gemm_bf16_ws(A.p, g.K, W.p, g.K, bias.p, out.p, ldo, g.M, g.N, g.K, g.epi, g.epi == WAN_EPI_RESID_F32 ? gate.p : nullptr, g.M, ws.p, wsb, nullptr));
                HIP(hipDeviceSynchronize());
                std::vector<int> h(1024); HIP(hipMemcpy(h.data(), ws.p, 4096, hipMemcpyDeviceToHost));
                double a = 0, b = 0, c = 0, d = 0, e = 0; for (int w = 0; w < 64; ++w) { a += h[640 + 5 * w]; b += h[641 + 5 * w]; c += h[642 + 5 * w]; d += h[643 + 5 * w]; e += h[644 + 5 * w]; }
                const double tot = a + b + c + d + e;
                printf("  cycles[%-18s] segment start %.1f %%, K loop %.1f %%, epilogue + switch %.1f %%, forced wait before the epilogue %.1f %%, forced drain after it %.1f %%  (total %.0f x16 ticks per workgroup; form mask %d)\n",
                       g.what, 100 * a / tot, 100 * b / tot, 100 * c / tot, 100 * d / tot, 100 * e / tot, tot / 64, wan_get_tuning("gemm_pk_form"));
            }
            double best[2] = {1e30, 1e30};
            for (int round = 0; round < 5; ++round)
                for (int arm = 0; arm < 2; ++arm) {
                    WAN(wan_set_tuning("gemm_pk_form", arm ? 31 : 0));
                    double ms = time_ms([&] { WAN(wan_gemm_bf16_ws(A.p, g.K, W.p, g.K, bias.p, out.p, ldo, g.M, g.N, g.K, g.epi,
                                                                   g.epi == WAN_EPI_RESID_F32 ? gate.p : nullptr, g.M, ws.p, wsb, nullptr)); }, 8, 2);
                    if (round > 0) best[arm] = std::min(best[arm], ms);         // round 0 warms the chip up
                }
            printf("  gemm %-18s M=%d N=%d K=%d: round-4 epilogue %.3f ms %.0f TF/s | row-permuted %.3f ms %.0f TF/s | ratio %.3f\n", g.what, g.M, g.N, g.K,
                   best[0], 2.0 * g.M * g.N * g.K / best[0] / 1e9, best[1], 2.0 * g.M * g.N * g.K / best[1] / 1e9, best[0] / best[1]);
            fflush(stdout);
            WAN(wan_set_tuning("gemm_pk_form", 31));
        }
    }
    if (mode == "gemmgm") {       // persistent GEMM: the rasterisation group GM (M tiles per group = the shape of the 32 tiles in flight on an XCD: GM x 32/GM) swept in one process
        WAN(wan_set_tuning("gemm_pk", 1));
        const int L = 67080;
        struct G { int M, N, K; int epi; const char* what; };
        std::vector<G> gs = {{L, 13824, 5120, WAN_EPI_GELU_BF16, "ffn.0+gelu"}, {L, 10240, 5120, WAN_EPI_BF16, "qk proj"},
                             {L, 5120, 5120, WAN_EPI_RESID_F32, "o proj+gate+resid"}, {L, 5120, 13824, WAN_EPI_RESID_F32, "ffn.2+resid"},
                             {L, 5120, 5120, WAN_EPI_BF16_T, "v proj (T)"}};
        std::vector<int> gms = {0, 1, 2, 3, 4, 6, 8, 16};
        if (argc > 2 && strchr(argv[2], ',')) { gms.clear(); for (char* t = strtok(argv[2], ","); t; t = strtok(nullptr, ",")) gms.push_back(atoi(t)); }
        const bool once = argc > 3 && !strcmp(argv[3], "once");       // one launch per (shape, gm): the form a rocprofv3 --pmc pass wants
        for (auto g : gs) {
            auto hA = to_bf(randn((size_t)4096 * 64));
            Dev<bf16> A((size_t)g.M * g.K), W((size_t)g.N * g.K);
            for (size_t off = 0; off < A.n; off += hA.size()) HIP(hipMemcpy(A.p + off, hA.data(), std::min(hA.size(), A.n - off) * 2, hipMemcpyHostToDevice));
            for (size_t off = 0; off < W.n; off += hA.size()) HIP(hipMemcpy(W.p + off, hA.data(), std::min(hA.size(), W.n - off) * 2, hipMemcpyHostToDevice));
            Dev<float> bias(g.N), gate(g.N); bias.zero(); gate.zero();
            const int64_t ldo = g.epi == WAN_EPI_BF16_T ? (g.M + 63) / 64 * 64 : g.N;
            const size_t osz = (size_t)(g.epi == WAN_EPI_BF16_T ? g.N : g.M) * ldo * ((g.epi == WAN_EPI_F32 || g.epi == WAN_EPI_RESID_F32) ? 4 : 2);
            Dev<char> out(osz); out.zero();
            const int64_t wsb = wan_gemm_workspace_bytes(g.M, g.N, g.K);
            Dev<char> ws((size_t)std::max<int64_t>(wsb, 16)); ws.zero();
            std::vector<double> best(gms.size(), 1e30);
            for (int round = 0; round < (once ? 1 : 4); ++round)
                for (size_t a = 0; a < gms.size(); ++a) {
                    WAN(wan_set_tuning("gemm_gm", gms[a]));
                    auto launch = [&] { WAN(wan_gemm_bf16_ws(A.p, g.K, W.p, g.K, bias.p, out.p, ldo, g.M, g.N, g.K, g.epi,
                                                             g.epi == WAN_EPI_RESID_F32 ? gate.p : nullptr, g.M, ws.p, wsb, nullptr)); };
                    if (once) { launch(); HIP(hipDeviceSynchronize()); continue; }
                    double ms = time_ms(launch, 6, 2);
                    if (round > 0) best[a] = std::min(best[a], ms);
                }
            if (once) continue;
            printf("  gemm %-18s M=%d N=%d K=%d:", g.what, g.M, g.N, g.K);
            for (size_t a = 0; a < gms.size(); ++a) printf("  gm=%d %.3f ms %.0f TF/s |", gms[a], best[a], 2.0 * g.M * g.N * g.K / best[a] / 1e9);
            printf("\n"); fflush(stdout);
        }
        WAN(wan_set_tuning("gemm_gm", 0));
    }
    if (mode == "gemmcyc") {      // `make EXPERIMENTS=1` builds: where a persistent-GEMM workgroup's cycles go, per epilogue form, with the forced waits
        if (wan_get_tuning("dev_experiments") != 1) { printf("gemmcyc needs a `make EXPERIMENTS=1` build\n"); return 2; }
        const int L = 67080;
        struct G { int M, N, K; int epi; const char* what; };
        std::vector<G> gs = {{L, 10240, 5120, WAN_EPI_BF16, "qk proj"}, {L, 13824, 5120, WAN_EPI_GELU_BF16, "ffn.0+gelu"},
                             {L, 5120, 5120, WAN_EPI_RESID_F32, "o proj+gate+resid"}, {L, 5120, 5120, WAN_EPI_BF16_T, "v proj (T)"},
                             {L, 3072, 1536, WAN_EPI_BF16, "1.3B qk proj"}, {L, 8960, 1536, WAN_EPI_GELU_BF16, "1.3B ffn.0+gelu"}};
        for (auto g : gs) {
            auto hA = to_bf(randn((size_t)4096 * 64));
            Dev<bf16> A((size_t)g.M * g.K), W((size_t)g.N * g.K);
            for (size_t off = 0; off < A.n; off += hA.size()) HIP(hipMemcpy(A.p + off, hA.data(), std::min(hA.size(), A.n - off) * 2, hipMemcpyHostToDevice));
            for (size_t off = 0; off < W.n; off += hA.size()) HIP(hipMemcpy(W.p + off, hA.data(), std::min(hA.size(), W.n - off) * 2, hipMemcpyHostToDevice));
            Dev<float> bias(g.N), gate(g.N); bias.zero(); gate.zero();
            const int64_t ldo = g.epi == WAN_EPI_BF16_T ? (g.M + 63) / 64 * 64 : g.N;
            const size_t osz = (size_t)(g.epi == WAN_EPI_BF16_T ? g.N : g.M) * ldo * ((g.epi == WAN_EPI_F32 || g.epi == WAN_EPI_RESID_F32) ? 4 : 2);
            Dev<char> out(osz); out.zero();
            const int64_t wsb = wan_gemm_workspace_bytes(g.M, g.N, g.K);
            Dev<char> ws((size_t)std::max<int64_t>(wsb, 16)); ws.zero();
            for (int form : {31})
                for (int ex : {0, 4096, 0, 4096, 64 | 4096}) {
                    WAN(wan_set_tuning("gemm_pk_form", form)); WAN(wan_set_tuning("gemm_exp", ex));
                    auto run = [&] { WAN(wan_gemm_bf16_ws(A.p, g.K, W.p, g.K, bias.p, out.p, ldo, g.M, g.N, g.K, g.epi, g.epi == WAN_EPI_RESID_F32 ? gate.p : nullptr, g.M, ws.p, wsb, nullptr)); };
                    const double ms = time_ms(run, 6, 2);
                    printf("  %-18s form %2d exp %4d: %.3f ms %5.0f TF/s", g.what, form, ex, ms, 2.0 * g.M * g.N * g.K / ms / 1e9);
                    if (ex & 64) {
                        HIP(hipDeviceSynchronize());
                        std::vector<int> h(1024); HIP(hipMemcpy(h.data(), ws.p, 4096, hipMemcpyDeviceToHost));
                        double a = 0, b = 0, c = 0, d = 0, e = 0; for (int w = 0; w < 64; ++w) { a += h[640 + 5 * w]; b += h[641 + 5 * w]; c += h[642 + 5 * w]; d += h[643 + 5 * w]; e += h[644 + 5 * w]; }
                        const double tot = a + b + c + d + e;
                        printf("   start %.1f %% | K loop %.1f %% | epilogue+switch %.1f %% | wait before %.1f %% | drain after %.1f %%  (%.0f x16 ticks / workgroup)", 100 * a / tot, 100 * b / tot, 100 * c / tot, 100 * d / tot, 100 * e / tot, tot / 64);
                    }
                    printf("\n"); fflush(stdout);
                }
            WAN(wan_set_tuning("gemm_pk_form", 29)); WAN(wan_set_tuning("gemm_exp", 0));
        }
    }
    if (mode == "sp") {           // the library-owned communicator from a C host: one rank, a pattern through both collectives
        unsigned char uid[WAN_SP_UNIQUE_ID_BYTES];
        WAN(wan_sp_unique_id(uid));
        wan_sp_comm* comm = nullptr;
        WAN(wan_sp_init(&comm, uid, 0, 1));
        const size_t n = 1 << 22;
        std::vector<bf16> h(n); for (size_t i = 0; i < n; ++i) h[i] = (bf16)(i * 2654435761u >> 16);
        Dev<bf16> send(h), recv(n), gath(n);
        recv.zero();
        hipStream_t cs; HIP(hipStreamCreate(&cs));
        WAN(wan_sp_a2a_scatter_heads(comm, send.p, recv.p, (int64_t)n * 2, cs));
        WAN(wan_sp_all_gather(comm, send.p, gath.p, (int64_t)n * 2, cs));
        WAN(wan_sp_wait(comm, cs));
        HIP(hipStreamSynchronize(cs));
        report("wan_sp_a2a_scatter_heads (1 rank: identity)", recv.host() == h ? 0.0 : 1.0, 0.0, "mismatch");
        report("wan_sp_all_gather (1 rank: identity)", gath.host() == h ? 0.0 : 1.0, 0.0, "mismatch");
        report("rank / world", (wan_sp_rank(comm) == 0 && wan_sp_world_size(comm) == 1) ? 0.0 : 1.0, 0.0, "mismatch");
        const bool rejects = wan_sp_a2a_gather_heads(comm, send.p, send.p, 64, cs) == WAN_ERR_INVALID;
        report("in-place exchange rejected", rejects ? 0.0 : 1.0, 0.0, "mismatch");
        WAN(wan_sp_destroy(comm));
        HIP(hipStreamDestroy(cs));
        printf("%s (%d failures)\n", g_fail ? "CHECK FAILED" : "ALL CHECKS PASSED", g_fail);
    }
    if (mode == "attnarms") {     // check_attn under every dispatch arm of wan_attention_fwd
        struct Arm { const char* name; int fast, ref; };
        for (Arm arm : {Arm{"lazy, reference in the accumulator", 0, 1}, Arm{"lazy, packed shift", 0, 2}, Arm{"max-free + fix-up", 1, 1}}) {
            printf("==== arm: %s\n", arm.name);
            WAN(wan_set_tuning("attn_fast", arm.fast)); WAN(wan_set_tuning("attn_ref", arm.ref));
            check_attn();
            printf("  last variant 0x%x\n", wan_get_tuning("last_attn_variant"));
        }
        printf("%s (%d failures)\n", g_fail ? "CHECK FAILED" : "ALL CHECKS PASSED", g_fail);
    }
    if (mode == "attnprof") {     // one shape, final kernel only: the target of the rocprofv3 --pmc passes (tools/profile_attn.sh)
        const int L = 67080, H = 40, C = H * 128; const int64_t ldvt = (L + 63) / 64 * 64;
        auto hq = to_bf(randn((size_t)4096 * 128));
        Dev<bf16> q((size_t)L * C), k((size_t)L * C), vt((size_t)C * ldvt), o((size_t)L * C);
        auto fill = [&](Dev<bf16>& d) { for (size_t off = 0; off < d.n; off += hq.size()) HIP(hipMemcpy(d.p + off, hq.data(), std::min(hq.size(), d.n - off) * 2, hipMemcpyHostToDevice)); };
        fill(k); fill(vt);
        { std::vector<float> hf = bf_to_f(hq); for (auto& x : hf) x *= WAN_ATTN_QSCALE(0.0883883f); hq = to_bf(hf); }
        fill(q);                                // q * softmax_scale * log2(e), as wan_rmsnorm_rope(x0_scale) hands it over
        const int64_t wsb = wan_attention_workspace_bytes(1, L, L, H, 128);
        Dev<char> ws((size_t)std::max<int64_t>(wsb, 16)); ws.zero();      // the header of the scratch must start as zero
        for (int i = 0; i < 3; ++i)
            WAN(wan_attention_fwd(q.p, C, 0, k.p, C, 0, vt.p, ldvt, 0, o.p, C, 0, 1, L, L, H, 128, 0.0883883f, WAN_ATTN_Q_PRESCALED, wsb ? ws.p : nullptr, wsb, nullptr));
        HIP(hipDeviceSynchronize());
        printf("attnprof done\n");
    }
    if (mode == "gemmring") {     // the four-stage-ring form of the 4-wave GEMM: numerics under every epilogue, then in-process A/B on the 14B shapes
        if (wan_get_tuning("dev_experiments") != 1) { printf("gemmring needs a `make EXPERIMENTS=1` build\n"); return 2; }
        WAN(wan_set_tuning("gemm_ring", 1));
        check_gemm();
        printf("%s (%d failures)\n", g_fail ? "CHECK FAILED" : "ALL CHECKS PASSED", g_fail);
        const int L = 67080;
        struct G { int M, N, K; int epi; const char* what; };
        std::vector<G> gs = {{L, 5120, 5120, WAN_EPI_BF16, "14B o/q proj"}, {L, 10240, 5120, WAN_EPI_BF16, "14B qk proj"},
                             {L, 13824, 5120, WAN_EPI_GELU_BF16, "14B ffn.0+gelu"}, {L, 5120, 13824, WAN_EPI_RESID_F32, "14B ffn.2+resid"},
                             {L, 5120, 5120, WAN_EPI_RESID_F32, "14B o+resid"}, {L, 5120, 5120, WAN_EPI_BF16_T, "14B v proj (T)"},
                             {8392, 5120, 13824, WAN_EPI_RESID_F32, "14B ffn.2, SP8 shard"}};
        WAN(wan_set_tuning("gemm_variant", 2)); WAN(wan_set_tuning("gemm_w4", 3));
        for (auto g : gs) {
            auto hA = to_bf(randn((size_t)4096 * 64));
            Dev<bf16> A((size_t)g.M * g.K), W((size_t)g.N * g.K);
            for (size_t off = 0; off < A.n; off += hA.size()) HIP(hipMemcpy(A.p + off, hA.data(), std::min(hA.size(), A.n - off) * 2, hipMemcpyHostToDevice));
            for (size_t off = 0; off < W.n; off += hA.size()) HIP(hipMemcpy(W.p + off, hA.data(), std::min(hA.size(), W.n - off) * 2, hipMemcpyHostToDevice));
            Dev<float> bias(g.N), gate(g.N); bias.zero(); gate.zero();
            const int64_t ldo = g.epi == WAN_EPI_BF16_T ? (g.M + 63) / 64 * 64 : g.N;
            const size_t osz = (size_t)(g.epi == WAN_EPI_BF16_T ? g.N : g.M) * ldo * ((g.epi == WAN_EPI_F32 || g.epi == WAN_EPI_RESID_F32) ? 4 : 2);
            Dev<char> out(osz); out.zero();
            for (int round = 0; round < 2; ++round)
            for (int ring = 0; ring < 2; ++ring) {
                WAN(wan_set_tuning("gemm_ring", ring));
                double ms = time_ms([&] { WAN(wan_gemm_bf16(A.p, g.K, W.p, g.K, bias.p, out.p, ldo, g.M, g.N, g.K, g.epi,
                                                            g.epi == WAN_EPI_RESID_F32 ? gate.p : nullptr, g.M, nullptr)); }, 3, 1);
                printf("  gemm[%s] %-20s M=%d N=%d K=%d: %.3f ms  %.0f TFLOP/s\n", ring ? "ring 4x32" : "2 x 64  ", g.what, g.M, g.N, g.K, ms, 2.0 * g.M * g.N * g.K / ms / 1e9);
                fflush(stdout);
            }
        }
    }
    if (mode == "attnq8") {       // wan_attention_fwd_qk8: exactness against its own (quantised) operands, error against the bf16 operands, speed
        // usage: kernel_check attnq8 [q_exp k_exp]
        const int qe = argc > 3 ? atoi(argv[2]) : 5, ke = argc > 3 ? atoi(argv[3]) : 2;
        auto f2e4m3 = [](float x) -> uint8_t {             // OCP e4m3fn, round to nearest even, saturating at +-448
            const uint8_t sgn = x < 0 ? 0x80 : 0; float ax = fabsf(x);
            if (!(ax == ax)) return 0x7f;
            if (ax >= 448.f) return sgn | 0x7e;
            if (ax < 0.015625f) return sgn | (uint8_t)lrintf(ax * 512.f);              // subnormals: steps of 2^-9 (8 -> the first normal)
            uint32_t u; memcpy(&u, &ax, 4);
            u += 0x7ffffu + ((u >> 20) & 1); u &= ~0xfffffu;                            // keep 3 mantissa bits
            const int e = (int)(u >> 23) - 127 + 7; const uint32_t m = (u >> 20) & 7;
            return sgn | (uint8_t)((e << 3) | m);
        };
        auto e4m32f = [](uint8_t b) -> float {
            const int e = (b >> 3) & 15, m = b & 7; const float v = e == 0 ? m * (1.f / 512.f) : ldexpf(1.f + m / 8.f, e - 7);
            return (b & 0x80) ? -v : v;
        };
        const float scale = 1.f / sqrtf(128.f), cc = WAN_ATTN_QSCALE(scale);
        struct Shape { int Lq, Lk, H; float qs; };
        for (Shape sh : {Shape{300, 420, 2, 1.f}, Shape{64, 64, 1, 1.f}, Shape{257, 8, 3, 1.f}, Shape{33, 1000, 1, 6.f}, Shape{520, 1500, 2, 40.f},
                         Shape{256, 4096, 1, 100.f}, Shape{86 * 256 + 10, 1100, 3, 1.f}}) {
            const int Lq = sh.Lq, Lk = sh.Lk, H = sh.H, C = H * 128;
            auto q = bf_round(randn((size_t)Lq * C, sh.qs)), k = bf_round(randn((size_t)Lk * C)), v = bf_round(randn((size_t)Lk * C));
            for (int j = 0; j < Lk; ++j) for (int c = 0; c < C; ++c) v[(size_t)j * C + c] = bf2f(f2bf(v[(size_t)j * C + c] + 0.01f * (c % 128) - 0.003f * (j % 97)));
            std::vector<uint8_t> q8(q.size()), k8(k.size());
            std::vector<float> qd(q.size()), kd(k.size()), qb(q.size());
            for (size_t i = 0; i < q.size(); ++i) {
                qb[i] = bf2f(f2bf(q[i] * cc));                                           // what the bf16 kernel would be handed
                q8[i] = f2e4m3(ldexpf(qb[i], qe)); qd[i] = ldexpf(e4m32f(q8[i]), -qe) / cc; qb[i] /= cc;
            }
            for (size_t i = 0; i < k.size(); ++i) { k8[i] = f2e4m3(ldexpf(k[i], ke)); kd[i] = ldexpf(e4m32f(k8[i]), -ke); }
            const int64_t ldvt = (Lk + 63) / 64 * 64;
            Dev<uint8_t> dq8(q8), dk8(k8);
            Dev<bf16> dv(to_bf(v)), dvt((size_t)C * ldvt), dout((size_t)Lq * C);
            WAN(wan_transpose_bf16(dv.p, C, dvt.p, ldvt, Lk, C, nullptr));
            const int64_t wsb = wan_attention_workspace_bytes(1, Lq, Lk, H, 128);
            Dev<char> ws((size_t)std::max<int64_t>(wsb, 16)); ws.zero();
            WAN(wan_attention_fwd_qk8(dq8.p, C, 0, qe, dk8.p, C, 0, ke, dvt.p, ldvt, 0, dout.p, C, 0, 1, Lq, Lk, H, 128, wsb ? ws.p : nullptr, wsb, nullptr));
            HIP(hipDeviceSynchronize());
            std::vector<int> rows;
            for (int i = 0; i < Lq; ++i) if (Lq <= 1024 || i < 64 || i >= Lq - 600 || i % 211 == 0) rows.push_back(i);
            std::vector<double> ref, ref16; attn_ref(qd, kd, v, Lq, Lk, H, scale, ref, rows); attn_ref(qb, k, v, Lq, Lk, H, scale, ref16, rows);
            auto all = bf_to_f(dout.host());
            std::vector<float> got(rows.size() * C);
            for (size_t ri = 0; ri < rows.size(); ++ri) memcpy(&got[ri * C], &all[(size_t)rows[ri] * C], C * sizeof(float));
            char nm[128]; snprintf(nm, sizeof nm, "qk8 Lq=%d Lk=%d H=%d qscale=%.0f (variant 0x%x): vs its own e4m3 operands", Lq, Lk, H, sh.qs, wan_get_tuning("last_attn_variant"));
            report(nm, rel_l2(ref, got), 6e-3);
            printf("       error against the bf16 operands (the cost of e4m3 q, k): rel_l2 %.3e\n", rel_l2(ref16, got));
        }
        printf("%s (%d failures)\n", g_fail ? "CHECK FAILED" : "ALL CHECKS PASSED", g_fail);
        {   // speed at the bench shape, alternating with the bf16 dispatch in one process
            const int L = 67080, H = 40, C = H * 128; const int64_t ldvt = (L + 63) / 64 * 64;
            auto hf = randn((size_t)4096 * 128);
            auto hq = to_bf(hf);
            std::vector<uint8_t> h8q(hf.size()), h8k(hf.size());
            for (size_t i = 0; i < hf.size(); ++i) { h8q[i] = f2e4m3(ldexpf(bf2f(f2bf(hf[i] * cc)), qe)); h8k[i] = f2e4m3(ldexpf(bf2f(hq[i]), ke)); }
            Dev<bf16> q((size_t)L * C), k((size_t)L * C), vt((size_t)C * ldvt), o((size_t)L * C);
            Dev<uint8_t> q8((size_t)L * C), k8((size_t)L * C);
            auto fill = [&](auto& d, const auto& src) { for (size_t off = 0; off < d.n; off += src.size()) HIP(hipMemcpy(d.p + off, src.data(), std::min(src.size(), d.n - off) * sizeof(src[0]), hipMemcpyHostToDevice)); };
            fill(k, hq); fill(vt, hq); fill(q8, h8q); fill(k8, h8k);
            { std::vector<float> t = bf_to_f(hq); for (auto& x : t) x *= cc; fill(q, to_bf(t)); }
            const int64_t wsb = wan_attention_workspace_bytes(1, L, L, H, 128);
            Dev<char> ws((size_t)std::max<int64_t>(wsb, 16)); ws.zero();
            for (int round = 0; round < 2; ++round) {
                double ms = time_ms([&] { WAN(wan_attention_fwd(q.p, C, 0, k.p, C, 0, vt.p, ldvt, 0, o.p, C, 0, 1, L, L, H, 128, scale, WAN_ATTN_Q_PRESCALED, wsb ? ws.p : nullptr, wsb, nullptr)); }, 4, 1);
                printf("  bf16 dispatch   L=%d H=%d: %.3f ms  %.0f TFLOP/s  (variant 0x%x)\n", L, H, ms, 4.0 * L * L * C / ms / 1e9, wan_get_tuning("last_attn_variant"));
                ms = time_ms([&] { WAN(wan_attention_fwd_qk8(q8.p, C, 0, qe, k8.p, C, 0, ke, vt.p, ldvt, 0, o.p, C, 0, 1, L, L, H, 128, wsb ? ws.p : nullptr, wsb, nullptr)); }, 4, 1);
                printf("  fp8 QK^T        L=%d H=%d: %.3f ms  %.0f TFLOP/s  (variant 0x%x)\n", L, H, ms, 4.0 * L * L * C / ms / 1e9, wan_get_tuning("last_attn_variant"));
                fflush(stdout);
            }
        }
    }
    if (mode == "attnf8") {       // wan_attention_fwd_f8 (fp8 QK^T and fp8 P.V): the V^T quantiser, the kernel against its own operands, speed
        const int qe = 5, ke = 2;
        auto f2e4m3 = [](float x) -> uint8_t {
            const uint8_t sgn = x < 0 ? 0x80 : 0; float ax = fabsf(x);
            if (!(ax == ax)) return 0x7f;
            if (ax >= 448.f) return sgn | 0x7e;
            if (ax < 0.015625f) return sgn | (uint8_t)lrintf(ax * 512.f);
            uint32_t u; memcpy(&u, &ax, 4);
            u += 0x7ffffu + ((u >> 20) & 1); u &= ~0xfffffu;
            const int e = (int)(u >> 23) - 127 + 7; const uint32_t m = (u >> 20) & 7;
            return sgn | (uint8_t)((e << 3) | m);
        };
        auto e4m32f = [](uint8_t b) -> float {
            const int e = (b >> 3) & 15, m = b & 7; const float v = e == 0 ? m * (1.f / 512.f) : ldexpf(1.f + m / 8.f, e - 7);
            return (b & 0x80) ? -v : v;
        };
        const float scale = 1.f / sqrtf(128.f), cc = WAN_ATTN_QSCALE(scale);
        struct Shape { int Lq, Lk, H; float qs; };
        for (Shape sh : {Shape{300, 420, 2, 1.f}, Shape{64, 64, 1, 1.f}, Shape{257, 8, 3, 1.f}, Shape{33, 1000, 1, 6.f}, Shape{520, 1500, 2, 40.f},
                         Shape{256, 4096, 1, 100.f}, Shape{86 * 256 + 10, 1100, 3, 1.f}}) {
            const int Lq = sh.Lq, Lk = sh.Lk, H = sh.H, C = H * 128;
            auto q = bf_round(randn((size_t)Lq * C, sh.qs)), k = bf_round(randn((size_t)Lk * C)), v = bf_round(randn((size_t)Lk * C));
            for (int j = 0; j < Lk; ++j) for (int c = 0; c < C; ++c) v[(size_t)j * C + c] = bf2f(f2bf(v[(size_t)j * C + c] + 0.01f * (c % 128) - 0.003f * (j % 97)));
            std::vector<uint8_t> q8(q.size()), k8(k.size());
            std::vector<float> qd(q.size()), kd(k.size()), qb(q.size());
            for (size_t i = 0; i < q.size(); ++i) {
                qb[i] = bf2f(f2bf(q[i] * cc));
                q8[i] = f2e4m3(ldexpf(qb[i], qe)); qd[i] = ldexpf(e4m32f(q8[i]), -qe) / cc; qb[i] /= cc;
            }
            for (size_t i = 0; i < k.size(); ++i) { k8[i] = f2e4m3(ldexpf(k[i], ke)); kd[i] = ldexpf(e4m32f(k8[i]), -ke); }
            const int64_t ldvt = (Lk + 63) / 64 * 64; const int nt = (int)(ldvt / 64);
            Dev<uint8_t> dq8(q8), dk8(k8), dv8((size_t)C * ldvt), dvs((size_t)wan_vt_mx_scale_bytes(1, H, Lk));
            Dev<bf16> dv(to_bf(v)), dvt((size_t)C * ldvt), dout((size_t)Lq * C);
            WAN(wan_transpose_bf16(dv.p, C, dvt.p, ldvt, Lk, C, nullptr));
            WAN(wan_vt_quantize_mx(dvt.p, ldvt, 0, 1, H, Lk, dv8.p, ldvt, 0, dvs.p, nullptr));
            HIP(hipDeviceSynchronize());
            // de-quantise V on the host from what the device wrote (layout: position 32 hi + 16 kt + 8 g + j <- key 32 kt + 16 g + 8 hi + j)
            auto hv8 = dv8.host(); auto hvs = dvs.host();
            std::vector<float> vd((size_t)Lk * C);
            double vnum = 0, vden = 0; bool range_ok = true;
            for (int c = 0; c < C; ++c) for (int key = 0; key < Lk; ++key) {
                const int tile = key >> 6, kk = key & 63, kt = kk >> 5, g = (kk >> 4) & 1, hi = (kk >> 3) & 1, j = kk & 7, d = c & 127, head = c >> 7;
                const uint8_t b = hv8[(size_t)c * ldvt + tile * 64 + 32 * hi + 16 * kt + 8 * g + j];
                const uint8_t sb = hvs[((size_t)head * nt + tile) * 256 + (32 * kt + (d & 31)) * 4 + (d >> 5)];      // MX block = (row, 32 consecutive keys)
                if (fabsf(e4m32f(b)) > 256.f) range_ok = false;            // block max / scale is in [128, 256): rounds to <= 256
                const float val = ldexpf(e4m32f(b), (int)sb - 127);
                vd[(size_t)key * C + c] = val;
                const double dd = val - v[(size_t)key * C + c]; vnum += dd * dd; vden += (double)v[(size_t)key * C + c] * v[(size_t)key * C + c];
            }
            if (getenv("F8_DEBUG")) {
                for (int key = 0; key < std::min(Lk, 72); key += 1) {
                    const int c = 5; const int tile = key >> 6, kk = key & 63, kt = kk >> 5, g = (kk >> 4) & 1, hi = (kk >> 3) & 1, j = kk & 7, d = c & 127;
                    printf("    c=5 key=%2d v=% .4f  byte 0x%02x scale 0x%02x -> % .4f | raw row bytes at key pos 0x%02x\n", key, v[(size_t)key * C + c],
                           hv8[(size_t)c * ldvt + tile * 64 + 32 * hi + 16 * kt + 8 * g + j], hvs[((size_t)0 * nt + tile) * 256 + (32 * kt + (d & 31)) * 4 + (d >> 5)],
                           vd[(size_t)key * C + c], hv8[(size_t)c * ldvt + key]);
                }
            }
            char nm[160]; snprintf(nm, sizeof nm, "f8 Lq=%d Lk=%d H=%d qscale=%.0f: wan_vt_quantize_mx round trip (MX e4m3, block values < 256: %s)", Lq, Lk, H, sh.qs, range_ok ? "yes" : "NO");
            report(nm, sqrt(vnum / vden) + (range_ok ? 0 : 1), 4e-2);
            const int64_t wsb = wan_attention_workspace_bytes(1, Lq, Lk, H, 128);
            Dev<char> ws((size_t)std::max<int64_t>(wsb, 16)); ws.zero();
            WAN(wan_attention_fwd_f8(dq8.p, C, 0, qe, dk8.p, C, 0, ke, dv8.p, ldvt, 0, dvs.p, dvt.p, ldvt, 0, dout.p, C, 0, 1, Lq, Lk, H, 128, ws.p, wsb, nullptr));
            HIP(hipDeviceSynchronize());
            std::vector<int> hdr(4); HIP(hipMemcpy(hdr.data(), ws.p, 16, hipMemcpyDeviceToHost));
            std::vector<int> rows;
            for (int i = 0; i < Lq; ++i) if (Lq <= 1024 || i < 64 || i >= Lq - 600 || i % 211 == 0) rows.push_back(i);
            std::vector<double> ref, ref16; attn_ref(qd, kd, vd, Lq, Lk, H, scale, ref, rows); attn_ref(qb, k, v, Lq, Lk, H, scale, ref16, rows);
            auto all = bf_to_f(dout.host());
            std::vector<float> got(rows.size() * C);
            for (size_t ri = 0; ri < rows.size(); ++ri) memcpy(&got[ri * C], &all[(size_t)rows[ri] * C], C * sizeof(float));
            if (getenv("F8_DEBUG")) {
                for (int ri : {0, 1, 40}) if (ri < (int)rows.size())
                    for (int c : {0, 5, 64, 127})
                        printf("    row %d c %d: got % .4f  ref(own operands) % .4f  ref(bf16 operands) % .4f\n", rows[ri], c, got[(size_t)ri * C + c], ref[(size_t)ri * C + c], ref16[(size_t)ri * C + c]);
            }
            snprintf(nm, sizeof nm, "   attention vs its own e4m3 q, k, V (what is left: the MX e4m3 rounding of P; variant 0x%x, redone %d)", wan_get_tuning("last_attn_variant"), hdr[1]);
            report(nm, rel_l2(ref, got), 4e-2);
            printf("       error against the bf16 operands: rel_l2 %.3e\n", rel_l2(ref16, got));
        }
        printf("%s (%d failures)\n", g_fail ? "CHECK FAILED" : "ALL CHECKS PASSED", g_fail);
        {   // speed at the bench shape, alternating in one process
            const int L = 67080, H = 40, C = H * 128; const int64_t ldvt = (L + 63) / 64 * 64;
            auto hf = randn((size_t)4096 * 128);
            auto hq = to_bf(hf);
            std::vector<uint8_t> h8q(hf.size()), h8k(hf.size());
            for (size_t i = 0; i < hf.size(); ++i) { h8q[i] = f2e4m3(ldexpf(bf2f(f2bf(hf[i] * cc)), qe)); h8k[i] = f2e4m3(ldexpf(bf2f(hq[i]), ke)); }
            Dev<bf16> q((size_t)L * C), k((size_t)L * C), vt((size_t)C * ldvt), o((size_t)L * C);
            Dev<uint8_t> q8((size_t)L * C), k8((size_t)L * C), v8((size_t)C * ldvt), vs((size_t)wan_vt_mx_scale_bytes(1, H, L));
            auto fill = [&](auto& d, const auto& src) { for (size_t off = 0; off < d.n; off += src.size()) HIP(hipMemcpy(d.p + off, src.data(), std::min(src.size(), d.n - off) * sizeof(src[0]), hipMemcpyHostToDevice)); };
            fill(k, hq); fill(vt, hq); fill(q8, h8q); fill(k8, h8k);
            { std::vector<float> t = bf_to_f(hq); for (auto& x : t) x *= cc; fill(q, to_bf(t)); }
            const int64_t wsb = wan_attention_workspace_bytes(1, L, L, H, 128);
            Dev<char> ws((size_t)std::max<int64_t>(wsb, 16)); ws.zero();
            double msq = time_ms([&] { WAN(wan_vt_quantize_mx(vt.p, ldvt, 0, 1, H, L, v8.p, ldvt, 0, vs.p, nullptr)); }, 3, 1);
            printf("  wan_vt_quantize_mx L=%d H=%d: %.3f ms (%.0f GB/s)\n", L, H, msq, 3.0 * C * ldvt / msq / 1e6);
            for (int round = 0; round < 2; ++round) {
                double ms = time_ms([&] { WAN(wan_attention_fwd(q.p, C, 0, k.p, C, 0, vt.p, ldvt, 0, o.p, C, 0, 1, L, L, H, 128, scale, WAN_ATTN_Q_PRESCALED, ws.p, wsb, nullptr)); }, 4, 1);
                printf("  bf16 dispatch       L=%d H=%d: %.3f ms  %.0f TFLOP/s  (variant 0x%x)\n", L, H, ms, 4.0 * L * L * C / ms / 1e9, wan_get_tuning("last_attn_variant"));
                ms = time_ms([&] { WAN(wan_attention_fwd_qk8(q8.p, C, 0, qe, k8.p, C, 0, ke, vt.p, ldvt, 0, o.p, C, 0, 1, L, L, H, 128, ws.p, wsb, nullptr)); }, 4, 1);
                printf("  fp8 QK^T            L=%d H=%d: %.3f ms  %.0f TFLOP/s  (variant 0x%x)\n", L, H, ms, 4.0 * L * L * C / ms / 1e9, wan_get_tuning("last_attn_variant"));
                ms = time_ms([&] { WAN(wan_attention_fwd_f8(q8.p, C, 0, qe, k8.p, C, 0, ke, v8.p, ldvt, 0, vs.p, vt.p, ldvt, 0, o.p, C, 0, 1, L, L, H, 128, ws.p, wsb, nullptr)); }, 4, 1);
                std::vector<int> hdr(4); HIP(hipMemcpy(hdr.data(), ws.p, 16, hipMemcpyDeviceToHost));
                printf("  fp8 QK^T + fp8 P.V  L=%d H=%d: %.3f ms  %.0f TFLOP/s  (variant 0x%x, redone workgroups %d)\n", L, H, ms, 4.0 * L * L * C / ms / 1e9, wan_get_tuning("last_attn_variant"), hdr[1]);
                fflush(stdout);
            }
        }
    }
    if (mode == "attnx") {        // in-process A/B of the self-attention launch at the bench shape: tuning key=value sets from argv
        // usage: kernel_check attnx [H] "k1=v1,k2=v2" "k1=v1" ...   (each quoted group is one arm; "" = defaults)
        const int L = 67080; int H = 40; int first = 2; int Lk = L;
        if (argc > 2 && atoi(argv[2]) > 0) { H = atoi(argv[2]); first = 3; }
        if (argc > first && !strncmp(argv[first], "Lk=", 3)) { Lk = atoi(argv[first] + 3); ++first; }     // cross-attention: Lk=512
        const int C = H * 128; const int64_t ldvt = (Lk + 63) / 64 * 64;
        auto hq = to_bf(randn((size_t)4096 * 128));
        Dev<bf16> q((size_t)L * C), k((size_t)Lk * C), vt((size_t)C * ldvt), o((size_t)L * C);
        auto fill = [&](Dev<bf16>& d) { for (size_t off = 0; off < d.n; off += hq.size()) HIP(hipMemcpy(d.p + off, hq.data(), std::min(hq.size(), d.n - off) * 2, hipMemcpyHostToDevice)); };
        fill(k); fill(vt);
        { std::vector<float> hf = bf_to_f(hq); for (auto& x : hf) x *= WAN_ATTN_QSCALE(0.0883883f); hq = to_bf(hf); }
        fill(q);
        const int64_t wsb = wan_attention_workspace_bytes(1, L, Lk, H, 128);
        Dev<char> ws((size_t)std::max<int64_t>(wsb, 16)); ws.zero();
        const char* keys[] = {"attn_tail", "attn_fast", "attn_xcd_map", "attn_persist", "attn_ref"};
        const int nkeys = 5;
        int defaults[5]; for (int i = 0; i < nkeys; ++i) defaults[i] = wan_get_tuning(keys[i]);
        for (int round = 0; round < 2; ++round)
        for (int ai = first; ai < argc; ++ai) {
            for (int i = 0; i < nkeys; ++i) WAN(wan_set_tuning(keys[i], defaults[i]));
            std::string arm = argv[ai], tok;
            for (size_t p0 = 0; p0 < arm.size();) {
                size_t p1 = arm.find(',', p0); if (p1 == std::string::npos) p1 = arm.size();
                tok = arm.substr(p0, p1 - p0); p0 = p1 + 1;
                const size_t eq = tok.find('=');
                if (eq != std::string::npos) WAN(wan_set_tuning(tok.substr(0, eq).c_str(), atoi(tok.c_str() + eq + 1)));
            }
            double ms = time_ms([&] { WAN(wan_attention_fwd(q.p, C, 0, k.p, C, 0, vt.p, ldvt, 0, o.p, C, 0, 1, L, Lk, H, 128, 0.0883883f, WAN_ATTN_Q_PRESCALED, wsb ? ws.p : nullptr, wsb, nullptr)); }, Lk == L ? 4 : 20, 2);
            printf("  attnx[%-28s] L=%d Lk=%d H=%d: %.3f ms  %.0f TFLOP/s  (variant 0x%x)\n", arm.c_str(), L, Lk, H, ms, 4.0 * L * Lk * C / ms / 1e9, wan_get_tuning("last_attn_variant"));
            fflush(stdout);
        }
    }
    if (mode == "perf" || mode == "all") perf(big);
    return g_fail ? 1 : 0;
}
