// experiments/t2_check.hip — standalone check + timing of the LDS-DMA tiled decode GEMM (experiments/decode5_lds_dma_gemm.hip) against the
// validated dec_gemm (controlar_amd/csrc/decode2.hip): all four epilogues, ragged M / K, then isolated times at the XL shapes for chains of
// 384 and 768 rows.  Test infrastructure.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I controlar_amd/csrc experiments/t2_check.hip -o experiments/t2_check && experiments/t2_check
#include "../controlar_amd/csrc/decode2.hip"
#include "decode5_lds_dma_gemm.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
static const int T_CFGS[] = {1, 2, 3, 4, 5};
static int g_rot = 0;

static unsigned long long rng_s = 0x9E3779B97F4A7C15ull;
static inline float frand() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (float)((rng_s >> 11) & 0xFFFFFF) / 8388608.0f - 1.0f; }
static inline float rb(float v) { return bf2f(f2bf(v)); }
static size_t xp_off(int m, int k, int K) { return ((((size_t)(m >> 4) * (K >> 5) + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (m & 15)) << 3) + (k & 7); }
template <typename T> static T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); return p; }
template <typename T> static void h2d(T* d, const std::vector<T>& h) { CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
template <typename T> static std::vector<T> d2h(const T* d, size_t n) { std::vector<T> h(n); CK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost)); return h; }
static std::vector<bf16_t> pack_rows(const std::vector<float>& a, int R, int K) {
    const int Rb = (R + 15) / 16;
    std::vector<bf16_t> o((size_t)Rb * 16 * K, 0);
    for (int r = 0; r < R; ++r) for (int k = 0; k < K; ++k) o[xp_off(r, k, K)] = f2bf(a[(size_t)r * K + k]);
    return o;
}
static int g_fail = 0;
static void report(const char* what, double maxerr, double tol, double frac_diff = -1) {
    const bool ok = maxerr <= tol && maxerr == maxerr;
    if (frac_diff >= 0) printf("%-78s max|d| %.3e  differing %.4f%%  tol %.1e  %s\n", what, maxerr, 100 * frac_diff, tol, ok ? "OK" : "FAIL");
    else printf("%-78s max|d| %.3e  tol %.1e  %s\n", what, maxerr, tol, ok ? "OK" : "FAIL");
    if (!ok) ++g_fail;
}

// the tiled GEMM (every configuration of T_CFGS) against dec_gemm on the same operands, all four epilogues.  The fp32 sums differ in order (one chain vs WAVES partial
// sums), so after the bf16 rounding points a value that sits on a rounding boundary may differ by one bf16 ulp: tolerance = 1 ulp of the
// output scale, and the share of differing elements is printed (expected well below 1 %).
static void check(int M, int N, int K, int H /* heads for the QKV case: N = 3*H*64 */, int pos) {
    std::vector<float> X((size_t)M * K), W((size_t)N * K);
    for (auto& v : X) v = rb(frand()); for (auto& v : W) v = rb(frand() * 0.1f);
    auto xpk = pack_rows(X, M, K), wpk = pack_rows(W, N, K);
    bf16_t* dX = dalloc<bf16_t>(xpk.size()); h2d(dX, xpk);
    bf16_t* dW = dalloc<bf16_t>(wpk.size()); h2d(dW, wpk);
    const int cfg0 = car_pick_gemm_cfg(M, N, K, EPI_LOGITS);
    char nm[160];
    auto cmpf = [&](const std::vector<float>& a, const std::vector<float>& b, double& e, double& fr) {
        e = 0; size_t nd = 0; for (size_t i = 0; i < a.size(); ++i) { const double d = std::fabs((double)a[i] - b[i]); if (d > 0) ++nd; if (!(d <= e)) e = d; } fr = (double)nd / a.size(); };
    auto cmpb = [&](const std::vector<bf16_t>& a, const std::vector<bf16_t>& b, double& e, double& fr) {
        e = 0; size_t nd = 0; for (size_t i = 0; i < a.size(); ++i) { const double d = std::fabs((double)bf2f(a[i]) - bf2f(b[i])); if (d > 0) ++nd; if (!(d <= e)) e = d; } fr = (double)nd / a.size(); };
    // LOGITS
    {
        float* d0 = dalloc<float>((size_t)M * N); float* d1 = dalloc<float>((size_t)M * N);
        GemmDP p; memset(&p, 0, sizeof(p)); p.W = dW; p.X = dX; p.M = M; p.N = N; p.K = K; p.outf = d0;
        car_launch_dec_gemm_cfg(&p, EPI_LOGITS, cfg0, 0); CK(hipDeviceSynchronize());
        auto r0 = d2h(d0, (size_t)M * N);
        for (int I : T_CFGS) {
            CK(hipMemset(d1, 0xff, (size_t)M * N * 4)); p.outf = d1;
            if (car_launch_dec_gemm_lds(&p, EPI_LOGITS, I, 0)) { printf("tiled cfg %d rejected\n", I); ++g_fail; continue; }
            CK(hipDeviceSynchronize()); CK(hipGetLastError());
            auto r1 = d2h(d1, (size_t)M * N); double e, fr; cmpf(r0, r1, e, fr);
            snprintf(nm, sizeof(nm), "tiled cfg %d LOGITS vs dec_gemm cfg %d (M=%d N=%d K=%d)", I, cfg0, M, N, K); report(nm, e, 0.04, fr);
        }
        CK(hipFree(d0)); CK(hipFree(d1));
    }
    // RESID
    {
        std::vector<bf16_t> h0((size_t)M * N); for (auto& v : h0) v = f2bf(frand() * 2.f);
        bf16_t* d0 = dalloc<bf16_t>(h0.size()); bf16_t* d1 = dalloc<bf16_t>(h0.size()); h2d(d0, h0);
        GemmDP p; memset(&p, 0, sizeof(p)); p.W = dW; p.X = dX; p.M = M; p.N = N; p.K = K; p.h = d0;
        car_launch_dec_gemm_cfg(&p, EPI_RESID, car_pick_gemm_cfg(M, N, K, EPI_RESID), 0); CK(hipDeviceSynchronize());
        auto r0 = d2h(d0, h0.size());
        for (int I : T_CFGS) {
            h2d(d1, h0); p.h = d1;
            car_launch_dec_gemm_lds(&p, EPI_RESID, I, 0); CK(hipDeviceSynchronize()); CK(hipGetLastError());
            auto r1 = d2h(d1, h0.size()); double e, fr; cmpb(r0, r1, e, fr);
            snprintf(nm, sizeof(nm), "tiled cfg %d RESID  vs dec_gemm (M=%d N=%d K=%d)", I, M, N, K); report(nm, e, 0.07, fr);
        }
        CK(hipFree(d0)); CK(hipFree(d1));
    }
    // SWIGLU
    {
        const size_t osz = (size_t)((M + 15) / 16) * 16 * (N / 2);
        bf16_t* d0 = dalloc<bf16_t>(osz); bf16_t* d1 = dalloc<bf16_t>(osz); CK(hipMemset(d0, 0, osz * 2));
        GemmDP p; memset(&p, 0, sizeof(p)); p.W = dW; p.X = dX; p.M = M; p.N = N; p.K = K; p.outp = d0;
        car_launch_dec_gemm_cfg(&p, EPI_SWIGLU, car_pick_gemm_cfg(M, N, K, EPI_SWIGLU), 0); CK(hipDeviceSynchronize());
        auto r0 = d2h(d0, osz);
        for (int I : T_CFGS) {
            CK(hipMemset(d1, 0, osz * 2)); p.outp = d1;
            car_launch_dec_gemm_lds(&p, EPI_SWIGLU, I, 0); CK(hipDeviceSynchronize()); CK(hipGetLastError());
            auto r1 = d2h(d1, osz); double e, fr; cmpb(r0, r1, e, fr);
            snprintf(nm, sizeof(nm), "tiled cfg %d SWIGLU vs dec_gemm (M=%d N=%d K=%d)", I, M, N, K); report(nm, e, 0.07, fr);
        }
        CK(hipFree(d0)); CK(hipFree(d1));
    }
    // QKV (N = 3*H*64): q scratch + K / V cache rows at `pos`
    if (H > 0 && N == 3 * H * 64) {
        const int dim = H * 64, SA = ((pos + 1 + 31) / 32) * 32;
        std::vector<float> rope((size_t)(pos + 1) * 64);
        for (int q = 0; q <= pos; ++q) for (int i = 0; i < 32; ++i) { const float a = 0.01f * q * (i + 1); rope[((size_t)q * 32 + i) * 2] = cosf(a); rope[((size_t)q * 32 + i) * 2 + 1] = sinf(a); }
        float* dR = dalloc<float>(rope.size()); h2d(dR, rope);
        int* dPos = dalloc<int>(1); CK(hipMemcpy(dPos, &pos, 4, hipMemcpyHostToDevice));
        const size_t csz = (size_t)M * H * SA * 64;
        bf16_t *k0 = dalloc<bf16_t>(csz), *v0 = dalloc<bf16_t>(csz), *k1 = dalloc<bf16_t>(csz), *v1 = dalloc<bf16_t>(csz), *q0 = dalloc<bf16_t>((size_t)M * dim), *q1 = dalloc<bf16_t>((size_t)M * dim);
        for (bf16_t* b : {k0, v0, k1, v1}) CK(hipMemset(b, 0, csz * 2));
        GemmDP p; memset(&p, 0, sizeof(p)); p.W = dW; p.X = dX; p.M = M; p.N = N; p.K = K; p.qout = q0; p.kc = k0; p.vc = v0; p.rope = dR; p.pos = dPos; p.H = H; p.SA = SA; p.dim = dim;
        car_launch_dec_gemm_cfg(&p, EPI_QKV, car_pick_gemm_cfg(M, N, K, EPI_QKV), 0); CK(hipDeviceSynchronize());
        auto rk = d2h(k0, csz), rv = d2h(v0, csz), rq = d2h(q0, (size_t)M * dim);
        for (int I : T_CFGS) {
            for (bf16_t* b : {k1, v1}) CK(hipMemset(b, 0, csz * 2));
            p.qout = q1; p.kc = k1; p.vc = v1;
            car_launch_dec_gemm_lds(&p, EPI_QKV, I, 0); CK(hipDeviceSynchronize()); CK(hipGetLastError());
            auto sk = d2h(k1, csz), sv = d2h(v1, csz), sq = d2h(q1, (size_t)M * dim);
            double e1, e2, e3, f1, f2, f3; cmpb(rk, sk, e1, f1); cmpb(rv, sv, e2, f2); cmpb(rq, sq, e3, f3);
            snprintf(nm, sizeof(nm), "tiled cfg %d QKV    vs dec_gemm (M=%d H=%d K=%d pos=%d): K cache | V cache | q", I, M, H, K, pos);
            report(nm, std::max(e1, std::max(e2, e3)), 0.07, std::max(f1, std::max(f2, f3 * (double)csz / ((double)M * dim))));
        }
        for (void* b : {(void*)k0, (void*)v0, (void*)k1, (void*)v1, (void*)q0, (void*)q1, (void*)dR, (void*)dPos}) CK(hipFree(b));
    }
    CK(hipFree(dX)); CK(hipFree(dW));
}

__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        const unsigned a = (x & 0x807fu) | (((x >> 7) & 0x3f) + 64) << 7, b2 = ((x >> 16) & 0x807fu) | ((((x >> 23) & 0x3f) + 64) << 7);
        p[i] = a | (b2 << 16); }
}

static void bench(int M) {
    const int D = 1280, Fh = 3584, H = 20, SA = 1152, pos = 631, NL = 8;
    struct Shape { const char* name; int N, K, epi; };
    const Shape shapes[] = {{"wqkv", 3 * D, D, EPI_QKV}, {"wo", D, D, EPI_RESID}, {"w1|w3", 2 * Fh, D, EPI_SWIGLU}, {"w2", D, Fh, EPI_RESID}, {"logits", 16384, D, EPI_LOGITS}};
    const size_t M16 = (size_t)((M + 15) / 16) * 16, kvper = (size_t)M * H * SA * 64;
    bf16_t *xn = dalloc<bf16_t>(M16 * Fh), *hbuf = dalloc<bf16_t>((size_t)M * D), *mid = dalloc<bf16_t>(M16 * Fh), *qb = dalloc<bf16_t>((size_t)M * D), *kv = dalloc<bf16_t>(kvper * 2);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (unsigned*)xn, M16 * Fh / 2, 7u);
    CK(hipMemset(hbuf, 0, (size_t)M * D * 2)); CK(hipMemset(kv, 0, kvper * 4));
    float* lgbuf = dalloc<float>((size_t)M * 16384);
    float* rope = dalloc<float>((size_t)1200 * 64); CK(hipMemset(rope, 0, 1200 * 64 * 4));
    int* dPos = dalloc<int>(1); CK(hipMemcpy(dPos, &pos, 4, hipMemcpyHostToDevice));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (const Shape& s : shapes) {
        const size_t wsz = (size_t)s.N * s.K;
        bf16_t* dW = dalloc<bf16_t>(wsz * NL);
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned*)dW, wsz * NL / 2, 12345u);
        CK(hipDeviceSynchronize());
        printf("M=%-3d %-6s N=%-5d K=%-4d (%5.1f GFLOP):", M, s.name, s.N, s.K, 2.0 * M * s.N * s.K / 1e9);
        for (int c = -1; c < (int)(sizeof(T_CFGS) / sizeof(int)); ++c) {
            auto launch = [&](int it) {
                GemmDP p; memset(&p, 0, sizeof(p)); p.W = dW + wsz * (it % NL); p.X = xn; p.M = M; p.N = s.N; p.K = s.K;
                p.h = hbuf; p.outf = lgbuf; p.outp = mid; p.qout = qb; p.kc = kv; p.vc = kv + kvper; p.rope = rope; p.pos = dPos; p.H = H; p.SA = SA; p.dim = D;
                if (c < 0) { const int cfg = car_pick_gemm_cfg(M, s.N, s.K, s.epi); const int J = (cfg / 10) % 10, Mb = (M + 15) / 16; p.w_nt = (Mb + J - 1) / J == 1; car_launch_dec_gemm_cfg(&p, s.epi, cfg, 0); }
                else if (car_launch_dec_gemm_lds(&p, s.epi, T_CFGS[c], 0)) { printf(" cfg %d rejected", T_CFGS[c]); }
            };
            for (int i = 0; i < 3; ++i) launch(i);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, 0));
            for (int i = 0; i < 48; ++i) launch(i);
            CK(hipEventRecord(t1, 0)); CK(hipEventSynchronize(t1)); CK(hipGetLastError());
            float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
            if (c < 0) printf("  dec_gemm %.1f us |", ms * 1000.f / 48); else printf("  t%d %.1f", T_CFGS[c], ms * 1000.f / 48);
        }
        printf("\n"); fflush(stdout);
        CK(hipFree(dW));
    }
    for (void* b : {(void*)xn, (void*)hbuf, (void*)mid, (void*)qb, (void*)kv, (void*)rope, (void*)dPos}) CK(hipFree(b));
}

int main(int argc, char** argv) {
    for (g_rot = 0; g_rot < 1; ++g_rot) {
    check(50, 256, 384, 0, 0);            // ragged M (Mb = 4), 6 stages
    check(100, 768, 320, 4, 37);          // QKV epilogue, 5 stages
    check(400, 512, 320, 0, 0);           // Mb = 25: ragged last row tile
    check(384, 3840, 1280, 20, 40);       // the XL wqkv at the bench chain size
    }
    printf("== correctness: %d failure(s)\n", g_fail);
    fflush(stdout);
    if (argc > 1 && !strcmp(argv[1], "check")) return g_fail ? 1 : 0;
    bench(384); bench(768);
    return g_fail ? 1 : 0;
}
