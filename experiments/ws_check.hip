// experiments/ws_check.hip — standalone check + timing of the weight-stationary slab GEMM (experiments/decode3_ws_gemm.hip) against
// the validated dec_gemm (decode2.hip), and the measurement the schedule sweeps could not make: ONE half-period of the two-chain decode
// step (the attention of one chain beside the six linears / norms of the other, on two streams) with either GEMM.  Test infrastructure.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I controlar_amd/csrc experiments/ws_check.hip -o experiments/ws_check && experiments/ws_check
#include "../controlar_amd/csrc/decode2.hip"
#include "decode3_ws_gemm.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

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

// ws (I = 2, 4) against dec_gemm on the same operands, all four epilogues.  The fp32 sums differ in order (one chain vs WAVES partial
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
        for (int I : {2, 4}) {
            CK(hipMemset(d1, 0xff, (size_t)M * N * 4)); p.outf = d1;
            if (car_launch_dec_gemm_ws(&p, EPI_LOGITS, I, 0)) { printf("ws I=%d rejected\n", I); ++g_fail; continue; }
            CK(hipDeviceSynchronize()); CK(hipGetLastError());
            auto r1 = d2h(d1, (size_t)M * N); double e, fr; cmpf(r0, r1, e, fr);
            snprintf(nm, sizeof(nm), "ws I=%d LOGITS vs dec_gemm cfg %d (M=%d N=%d K=%d)", I, cfg0, M, N, K); report(nm, e, 0.04, fr);
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
        for (int I : {2, 4}) {
            h2d(d1, h0); p.h = d1;
            car_launch_dec_gemm_ws(&p, EPI_RESID, I, 0); CK(hipDeviceSynchronize()); CK(hipGetLastError());
            auto r1 = d2h(d1, h0.size()); double e, fr; cmpb(r0, r1, e, fr);
            snprintf(nm, sizeof(nm), "ws I=%d RESID  vs dec_gemm (M=%d N=%d K=%d)", I, M, N, K); report(nm, e, 0.07, fr);
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
        for (int I : {2, 4}) {
            CK(hipMemset(d1, 0, osz * 2)); p.outp = d1;
            car_launch_dec_gemm_ws(&p, EPI_SWIGLU, I, 0); CK(hipDeviceSynchronize()); CK(hipGetLastError());
            auto r1 = d2h(d1, osz); double e, fr; cmpb(r0, r1, e, fr);
            snprintf(nm, sizeof(nm), "ws I=%d SWIGLU vs dec_gemm (M=%d N=%d K=%d)", I, M, N, K); report(nm, e, 0.07, fr);
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
        for (int I : {2, 4}) {
            for (bf16_t* b : {k1, v1}) CK(hipMemset(b, 0, csz * 2));
            p.qout = q1; p.kc = k1; p.vc = v1;
            car_launch_dec_gemm_ws(&p, EPI_QKV, I, 0); CK(hipDeviceSynchronize()); CK(hipGetLastError());
            auto sk = d2h(k1, csz), sv = d2h(v1, csz), sq = d2h(q1, (size_t)M * dim);
            double e1, e2, e3, f1, f2, f3; cmpb(rk, sk, e1, f1); cmpb(rv, sv, e2, f2); cmpb(rq, sq, e3, f3);
            snprintf(nm, sizeof(nm), "ws I=%d QKV    vs dec_gemm (M=%d H=%d K=%d pos=%d): K cache | V cache | q", I, M, H, K, pos);
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

__global__ void ts_kernel(long long* out, int slot) { if (threadIdx.x == 0) out[slot] = wall_clock64(); }      // 100 MHz wall clock

// ---- one chain-layer of linears on `st`: wo -> ffn_norm -> w1|w3 -> w2 -> attention_norm -> wqkv (what the OTHER chain runs under an attention)
struct Layer {
    int M, D, Fh, H, SA; bf16_t *wqkv, *wo, *w13, *w2, *xn, *att, *mid, *h, *q, *kc, *vc, *nw; float* rope; int* pos;
};
static void linears(const Layer& L, int mode /* 0 dec_gemm, 2/4 ws I */, hipStream_t st) {
    auto gemm = [&](const bf16_t* W, const bf16_t* X, int N, int K, int epi, GemmDP p) {
        p.W = W; p.X = X; p.M = L.M; p.N = N; p.K = K;
        const int cfg = car_pick_gemm_cfg(L.M, N, K, epi); const int J = (cfg / 10) % 10, Mb = (L.M + 15) / 16;
        p.w_nt = (Mb + J - 1) / J == 1;
        if (mode == 0 || car_launch_dec_gemm_ws(&p, epi, mode, st)) car_launch_dec_gemm_cfg(&p, epi, cfg, st);
    };
    GemmDP z; memset(&z, 0, sizeof(z));
    { GemmDP q = z; q.h = L.h; gemm(L.wo, L.att, L.D, L.D, EPI_RESID, q); }
    { Norm2P n; memset(&n, 0, sizeof(n)); n.h_in = L.h; n.xn = L.xn; n.w = L.nw; n.D = L.D; n.eps = 1e-5f; car_launch_rmsnorm2(&n, L.M, st); }
    { GemmDP q = z; q.outp = L.mid; gemm(L.w13, L.xn, 2 * L.Fh, L.D, EPI_SWIGLU, q); }
    { GemmDP q = z; q.h = L.h; gemm(L.w2, L.mid, L.D, L.Fh, EPI_RESID, q); }
    { Norm2P n; memset(&n, 0, sizeof(n)); n.h_in = L.h; n.xn = L.xn; n.w = L.nw; n.D = L.D; n.eps = 1e-5f; car_launch_rmsnorm2(&n, L.M, st); }
    { GemmDP q = z; q.qout = L.q; q.kc = L.kc; q.vc = L.vc; q.rope = L.rope; q.pos = L.pos; q.H = L.H; q.SA = L.SA; q.dim = L.D; gemm(L.wqkv, L.xn, 3 * L.D, L.D, EPI_QKV, q); }
}

static void bench(int M) {
    const int D = 1280, Fh = 3584, H = 20, T = 120, SA = 1152, pos = 631, NL = 6, NKV = 2, HP = 36;
    const size_t per_layer = (size_t)(3 * D * D + D * D + 2 * Fh * D + D * Fh);
    bf16_t* dW = dalloc<bf16_t>(per_layer * NL);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned*)dW, per_layer * NL / 2, 12345u);
    const size_t kvper = (size_t)M * H * SA * 64;
    bf16_t* dKV = dalloc<bf16_t>(kvper * 2 * NKV);
    hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, (unsigned*)dKV, kvper * 2 * NKV / 2, 999u);
    const size_t M16 = (size_t)((M + 15) / 16) * 16;
    bf16_t *xn = dalloc<bf16_t>(M16 * D), *att = dalloc<bf16_t>(M16 * D), *mid = dalloc<bf16_t>(M16 * Fh), *hbuf = dalloc<bf16_t>((size_t)M * D), *qb = dalloc<bf16_t>((size_t)M * D), *qa = dalloc<bf16_t>((size_t)M * D), *oa = dalloc<bf16_t>(M16 * D), *nw = dalloc<bf16_t>(D);
    for (auto pr : {std::make_pair(xn, M16 * D), std::make_pair(att, M16 * D), std::make_pair(mid, M16 * Fh), std::make_pair(qa, (size_t)M * D), std::make_pair(nw, (size_t)D)})
        hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, (unsigned*)pr.first, pr.second / 2, 7u);
    CK(hipMemset(hbuf, 0, (size_t)M * D * 2));
    float* rope = dalloc<float>((size_t)1200 * 64); CK(hipMemset(rope, 0, 1200 * 64 * 4));
    int* dPos = dalloc<int>(1); CK(hipMemcpy(dPos, &pos, 4, hipMemcpyHostToDevice));
    std::vector<unsigned char> mask((size_t)M * T, 0);
    for (int i = 0; i < M; ++i) { const int Lv = 8 + (i * 13) % 33; for (int t = T - Lv; t < T; ++t) mask[(size_t)i * T + t] = 1; }
    unsigned char* dM = dalloc<unsigned char>(mask.size()); h2d(dM, mask);
    int* dJ = dalloc<int>(M); car_launch_mask_first_valid(dM, dJ, M, T, 0);
    CK(hipDeviceSynchronize());
    hipStream_t sA, sB; CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
    hipEvent_t eA, eB, t0, t1; CK(hipEventCreateWithFlags(&eA, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eB, hipEventDisableTiming)); CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    auto layer = [&](int it) {
        Layer L; L.M = M; L.D = D; L.Fh = Fh; L.H = H; L.SA = SA; bf16_t* w = dW + per_layer * (it % NL);
        L.wqkv = w; L.wo = w + (size_t)3 * D * D; L.w13 = L.wo + (size_t)D * D; L.w2 = L.w13 + (size_t)2 * Fh * D;
        L.xn = xn; L.att = att; L.mid = mid; L.h = hbuf; L.q = qb; L.kc = dKV + kvper * 2 * ((it + 1) % NKV); L.vc = L.kc + kvper; L.nw = nw; L.rope = rope; L.pos = dPos;
        return L;
    };
    auto attention = [&](int it, hipStream_t st) {
        Attn2P a; memset(&a, 0, sizeof(a)); a.q = qa; a.pos = dPos; a.mask = dM; a.jmin = dJ; a.out = oa; a.H = H; a.SA = SA; a.T = T; a.dim = D; a.nsplit = 1; a.out_packed = 1;
        a.kc = dKV + kvper * 2 * (it % NKV); a.vc = a.kc + kvper;
        car_launch_dec_attn2_var(&a, M, 40, 0, st);
    };
    auto timed = [&](const std::function<void(int)>& body) {
        for (int i = 0; i < 3; ++i) body(i);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(t0, sA));
        for (int i = 0; i < HP; ++i) body(i);
        CK(hipEventRecord(eB, sB)); CK(hipStreamWaitEvent(sA, eB, 0));
        CK(hipEventRecord(t1, sA)); CK(hipEventSynchronize(t1));
        float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1)); CK(hipGetLastError());
        return ms * 1000.f / HP;
    };
    double rows = 0; for (int i = 0; i < M; ++i) { const int Lv = 8 + (i * 13) % 33; rows += pos + 1 - (T - Lv); }
    const double abytes = rows * H * 256.0;
    const float ta = timed([&](int it) { attention(it, sA); });
    printf("M=%d  attention alone                    %7.1f us per launch  (%.2f TB/s of valid KV rows, pos %d)\n", M, ta, abytes / 1e6 / ta, pos);
    for (int mode : {0, 4, 2}) {
        const char* nm = mode == 0 ? "dec_gemm (64x64 K-split tiles)" : (mode == 4 ? "ws I=4 (64 rows x all M)      " : "ws I=2 (32 rows x all M)      ");
        const float tl = timed([&](int it) { linears(layer(it), mode, sA); });
        // both streams start each half-period together and join at its end: the steady state of two chains in anti-phase
        const float tc = timed([&](int it) {
            CK(hipEventRecord(eA, sA)); CK(hipStreamWaitEvent(sB, eA, 0));
            attention(it, sA); linears(layer(it), mode, sB);
            CK(hipEventRecord(eB, sB)); CK(hipStreamWaitEvent(sA, eB, 0));
        });
        printf("M=%d  %s  six linears+norms alone %6.1f us | beside the attention %6.1f us per half-period (serial sum %6.1f, attention alone %6.1f)\n", M, nm, tl, tc, ta + tl, ta);
        fflush(stdout);
    }
    {   // where the two-chain graph's hidden 2.5 ms per step comes from: the linears of BOTH chains side by side (lockstep)
        const float t2 = timed([&](int it) {
            CK(hipEventRecord(eA, sA)); CK(hipStreamWaitEvent(sB, eA, 0));
            linears(layer(it), 0, sA); linears(layer(it + 3), 0, sB);
            CK(hipEventRecord(eB, sB)); CK(hipStreamWaitEvent(sA, eB, 0));
        });
        printf("M=%d  dec_gemm: the linears+norms of TWO chains on two streams %6.1f us (one chain alone: see above)\n", M, t2);
        // a resident attention grid (R workgroups per CU walking the items) beside the linears: does a shallower memory queue let them through?
        for (int R : {2, 3, 4}) {
            auto pattn = [&](int it, hipStream_t st) {
                Attn2P a; memset(&a, 0, sizeof(a)); a.q = qa; a.pos = dPos; a.mask = dM; a.jmin = dJ; a.out = oa; a.H = H; a.SA = SA; a.T = T; a.dim = D; a.nsplit = 1; a.out_packed = 1;
                a.kc = dKV + kvper * 2 * (it % NKV); a.vc = a.kc + kvper;
                const long items = (long)M * H, cap = 256L * R, per = (items + cap - 1) / cap; a.n_seq = M; a.pgrid = (int)((items + per - 1) / per);
                car_launch_dec_attn2_var(&a, M, 40, 0, st);
            };
            const float tp = timed([&](int it) { pattn(it, sA); });
            const float tpc = timed([&](int it) {
                CK(hipEventRecord(eA, sA)); CK(hipStreamWaitEvent(sB, eA, 0));
                pattn(it, sA); linears(layer(it), 0, sB);
                CK(hipEventRecord(eB, sB)); CK(hipStreamWaitEvent(sA, eB, 0));
            });
            printf("M=%d  resident attention grid R=%d: alone %6.1f us | beside the dec_gemm linears %6.1f us per half-period\n", M, R, tp, tpc);
        }
        fflush(stdout);
    }
    {   // is it the kernel BOUNDARIES of the linear chain that cannot pass a saturating streamer (end-of-kernel L2 write-back / start-of-kernel
        // invalidate), or any kernel?  ONE long dec_gemm launch (a 6-layer stack of w1|w3 as a single 43008-row weight matrix, LOGITS epilogue)
        // beside the attention: if this overlaps where the 6-kernel chain does not, the chain has to become one launch.
        const int Nbig = 2 * Fh * NL; float* big = dalloc<float>((size_t)M * Nbig);
        auto one = [&](hipStream_t st) {
            GemmDP p; memset(&p, 0, sizeof(p)); p.W = dW; p.X = xn; p.M = M; p.N = Nbig; p.K = D; p.outf = big;
            car_launch_dec_gemm_cfg(&p, EPI_LOGITS, car_pick_gemm_cfg(M, Nbig, D, EPI_LOGITS), st);
        };
        const float t1k = timed([&](int) { one(sA); });
        const float t1c = timed([&](int it) {
            CK(hipEventRecord(eA, sA)); CK(hipStreamWaitEvent(sB, eA, 0));
            attention(it, sA); one(sB);
            CK(hipEventRecord(eB, sB)); CK(hipStreamWaitEvent(sA, eB, 0));
        });
        printf("M=%d  ONE long dec_gemm launch (N=%d): alone %6.1f us | beside the attention %6.1f us (serial sum %6.1f)\n", M, Nbig, t1k, t1c, ta + t1k);
        // the same with the GEMM enqueued FIRST (if a kernel's start-of-kernel cache invalidate has to wait for the streamer's traffic to drain,
        // only a kernel that started before the attention can run beside it)
        const float t1r = timed([&](int it) {
            CK(hipEventRecord(eA, sA)); CK(hipStreamWaitEvent(sB, eA, 0));
            one(sB); attention(it, sA);
            CK(hipEventRecord(eB, sB)); CK(hipStreamWaitEvent(sA, eB, 0));
        });
        printf("M=%d  ONE long dec_gemm launch enqueued before the attention: %6.1f us per half-period\n", M, t1r);
        fflush(stdout);
        CK(hipFree(big));
    }
    {   // WHEN does a kernel launched on a second stream start while the attention runs?  Timestamp kernels (one lane, no memory traffic to
        // speak of): stream A = ts0, attention, ts1; stream B (released by an event behind ts0) = ts2 .. ts7 as a chain of six dependent launches.
        // If ts2 - ts0 is a few us, trivial kernels start freely beside the attention and only memory-heavy ones stall (explanation (b) of
        // DESIGN.md §4); if ts2 lands at ts1, NO kernel starts until the attention's traffic has drained (explanation (a)).
        long long* dts = dalloc<long long>(16); CK(hipMemset(dts, 0, 16 * 8));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(ts_kernel, dim3(1), dim3(64), 0, sA, dts, 0);
            CK(hipEventRecord(eA, sA)); CK(hipStreamWaitEvent(sB, eA, 0));
            attention(rep, sA);
            hipLaunchKernelGGL(ts_kernel, dim3(1), dim3(64), 0, sA, dts, 1);
            for (int k = 2; k < 8; ++k) hipLaunchKernelGGL(ts_kernel, dim3(1), dim3(64), 0, sB, dts, k);
            CK(hipDeviceSynchronize());
            long long h[8]; CK(hipMemcpy(h, dts, sizeof(h), hipMemcpyDeviceToHost));
            printf("M=%d  timestamps (us after ts0): attention done %.1f | first kernel on the other stream ran at %.1f, the sixth of its chain at %.1f\n", M,
                   (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0, (h[7] - h[0]) / 100.0);
        }
        // the same with six rmsnorm2 launches (384 rows x 2.5 KB read + written each: light but real memory traffic) between two timestamps
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(ts_kernel, dim3(1), dim3(64), 0, sA, dts, 0);
            CK(hipEventRecord(eA, sA)); CK(hipStreamWaitEvent(sB, eA, 0));
            attention(rep, sA);
            hipLaunchKernelGGL(ts_kernel, dim3(1), dim3(64), 0, sA, dts, 1);
            hipLaunchKernelGGL(ts_kernel, dim3(1), dim3(64), 0, sB, dts, 2);
            for (int k = 0; k < 6; ++k) { Norm2P n; memset(&n, 0, sizeof(n)); n.h_in = hbuf; n.xn = xn; n.w = nw; n.D = D; n.eps = 1e-5f; car_launch_rmsnorm2(&n, M, sB); }
            hipLaunchKernelGGL(ts_kernel, dim3(1), dim3(64), 0, sB, dts, 3);
            CK(hipDeviceSynchronize());
            long long h[4]; CK(hipMemcpy(h, dts, sizeof(h), hipMemcpyDeviceToHost));
            printf("M=%d  timestamps (us after ts0): attention done %.1f | six rmsnorm2 launches on the other stream ran from %.1f to %.1f\n", M,
                   (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0, (h[3] - h[0]) / 100.0);
        }
        fflush(stdout);
        CK(hipFree(dts));
    }
    // isolated per-shape times
    struct Shape { const char* name; int N, K, epi; size_t off; };
    const Shape shapes[] = {{"wqkv", 3 * D, D, EPI_QKV, 0}, {"wo", D, D, EPI_RESID, (size_t)3 * D * D}, {"w13", 2 * Fh, D, EPI_SWIGLU, (size_t)4 * D * D}, {"w2", D, Fh, EPI_RESID, (size_t)4 * D * D + (size_t)2 * Fh * D}};
    for (const Shape& s : shapes) {
        printf("M=%d  %-5s N=%-5d K=%-4d:", M, s.name, s.N, s.K);
        for (int mode : {0, 4, 2}) {
            const float us = timed([&](int it) {
                GemmDP p; memset(&p, 0, sizeof(p)); p.W = dW + per_layer * (it % NL) + s.off; p.X = s.K == D ? xn : mid; p.M = M; p.N = s.N; p.K = s.K;
                p.h = hbuf; p.outp = mid; p.qout = qb; p.kc = dKV; p.vc = dKV + kvper; p.rope = rope; p.pos = dPos; p.H = H; p.SA = SA; p.dim = D;
                const int cfg = car_pick_gemm_cfg(M, s.N, s.K, s.epi); const int J = (cfg / 10) % 10, Mb = (M + 15) / 16; p.w_nt = (Mb + J - 1) / J == 1;
                if (mode == 0 || car_launch_dec_gemm_ws(&p, s.epi, mode, sA)) car_launch_dec_gemm_cfg(&p, s.epi, cfg, sA);
            });
            printf("  %s %.1f us", mode == 0 ? "dec_gemm" : (mode == 4 ? "ws4" : "ws2"), us);
        }
        printf("\n"); fflush(stdout);
    }
    for (void* b : {(void*)dW, (void*)dKV, (void*)xn, (void*)att, (void*)mid, (void*)hbuf, (void*)qb, (void*)qa, (void*)oa, (void*)nw, (void*)rope, (void*)dPos, (void*)dM, (void*)dJ}) CK(hipFree(b));
    CK(hipStreamDestroy(sA)); CK(hipStreamDestroy(sB));
}

int main(int argc, char** argv) {
    check(50, 256, 352, 0, 0);            // ragged M (Mb = 4: waves 4..7 idle), nkb = 11 (ragged last iteration)
    check(100, 768, 352, 4, 37);          // QKV epilogue, J = 1 with 7 m-blocks
    check(400, 512, 320, 0, 0);           // Mb = 25 > 24: two row groups (grid.y = 2)
    check(384, 3840, 1280, 20, 40);       // the XL wqkv at the bench chain size (J = 3)
    printf("== correctness: %d failure(s)\n", g_fail);
    fflush(stdout);
    if (argc > 1 && !strcmp(argv[1], "check")) return g_fail ? 1 : 0;
    bench(384);
    return g_fail ? 1 : 0;
}
