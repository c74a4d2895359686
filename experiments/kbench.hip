// experiments/kbench.hip — standalone validation + timing of the round-2 decode kernels (controlar_amd/csrc/decode2.hip)
// against host references, before/after they are wired into engine.hip.  Test infrastructure, not product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I controlar_amd/csrc experiments/kbench.hip -o experiments/kbench && experiments/kbench [quick]
#include "../controlar_amd/csrc/decode2.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static unsigned long long rng_s = 0x9E3779B97F4A7C15ull;
static inline float frand() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (float)((rng_s >> 11) & 0xFFFFFF) / 8388608.0f - 1.0f; }   // [-1,1)
static inline float rb(float v) { return bf2f(f2bf(v)); }

static size_t xp_off(int m, int k, int K) { return ((((size_t)(m >> 4) * (K >> 5) + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (m & 15)) << 3) + (k & 7); }
static size_t k_off(int p, int d) { return ((size_t)(p >> 4) * 2 + (d >> 5)) * 512 + ((((d & 31) >> 3) * 16 + (p & 15)) << 3) + (d & 7); }
static size_t v_off(int p, int d) { const int w = p & 31, qv = w < 16 ? (w >> 2) : ((w - 16) >> 2), ev = w < 16 ? (w & 3) : (4 + ((w - 16) & 3));
                                    return ((size_t)(p >> 5) * 4 + (d >> 4)) * 512 + ((qv * 16 + (d & 15)) << 3) + ev; }

template <typename T> static T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); return p; }
template <typename T> static void h2d(T* d, const std::vector<T>& h) { CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
template <typename T> static std::vector<T> d2h(const T* d, size_t n) { std::vector<T> h(n); CK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost)); return h; }

static std::vector<bf16_t> pack_rows(const std::vector<float>& a, int R, int K) {      // rows padded to a multiple of 16
    const int Rb = (R + 15) / 16;
    std::vector<bf16_t> o((size_t)Rb * 16 * K, 0);
    for (int r = 0; r < R; ++r) for (int k = 0; k < K; ++k) o[xp_off(r, k, K)] = f2bf(a[(size_t)r * K + k]);
    return o;
}

static int g_fail = 0;
static void report(const char* what, double maxerr, double tol) {
    const bool ok = maxerr <= tol && maxerr == maxerr;
    printf("%-64s max|d| %.3e  tol %.1e  %s\n", what, maxerr, tol, ok ? "OK" : "FAIL");
    if (!ok) ++g_fail;
}

static const int ALL_CFG[] = {110, 111, 120, 121, 140, 141, 210, 211, 220, 221, 240, 241, 410, 411, 420, 421, 440, 441};

// ------------------------------------------------------------------------------------------------ GEMM correctness
static void test_gemm() {
    const int M = 50, N = 256, K = 352;      // Mb = 4 with a ragged last block; nkb = 11 (uneven K split)
    std::vector<float> X((size_t)M * K), W((size_t)N * K);
    for (auto& v : X) v = rb(frand()); for (auto& v : W) v = rb(frand() * 0.1f);
    std::vector<double> ref((size_t)M * N);
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)X[(size_t)m * K + k] * W[(size_t)n * K + k]; ref[(size_t)m * N + n] = s; }
    auto xp = pack_rows(X, M, K), wp = pack_rows(W, N, K);
    bf16_t* dX = dalloc<bf16_t>(xp.size()); h2d(dX, xp);
    bf16_t* dW = dalloc<bf16_t>(wp.size()); h2d(dW, wp);
    float* dO = dalloc<float>((size_t)M * N);
    for (int cfg : ALL_CFG) {
        CK(hipMemset(dO, 0xff, (size_t)M * N * 4));
        GemmDP p; memset(&p, 0, sizeof(p)); p.W = dW; p.X = dX; p.M = M; p.N = N; p.K = K; p.outf = dO; p.w_nt = cfg % 2;
        if (car_launch_dec_gemm_cfg(&p, EPI_LOGITS, cfg, 0)) { printf("cfg %d rejected\n", cfg); ++g_fail; continue; }
        CK(hipDeviceSynchronize());
        auto o = d2h(dO, (size_t)M * N);
        double e = 0; for (size_t i = 0; i < o.size(); ++i) e = std::max(e, std::fabs((double)o[i] - (double)rb((float)ref[i])));
        char nm[96]; snprintf(nm, sizeof(nm), "dec_gemm LOGITS cfg %d (M=%d N=%d K=%d)", cfg, M, N, K);
        report(nm, e, 0.04);       // one bf16 ulp at |x| <= 8 is 0.03: a value on a rounding boundary may flip
    }
    // RESID
    {
        std::vector<bf16_t> h0((size_t)M * N); for (auto& v : h0) v = f2bf(frand() * 2.f);
        bf16_t* dH = dalloc<bf16_t>(h0.size());
        for (int cfg : {220, 441, 111}) {
            h2d(dH, h0);
            GemmDP p; memset(&p, 0, sizeof(p)); p.W = dW; p.X = dX; p.M = M; p.N = N; p.K = K; p.h = dH;
            car_launch_dec_gemm_cfg(&p, EPI_RESID, cfg, 0); CK(hipDeviceSynchronize());
            auto o = d2h(dH, h0.size());
            double e = 0; for (size_t i = 0; i < o.size(); ++i) e = std::max(e, std::fabs((double)bf2f(o[i]) - (double)rb(bf2f(h0[i]) + rb((float)ref[i]))));
            char nm[96]; snprintf(nm, sizeof(nm), "dec_gemm RESID cfg %d", cfg); report(nm, e, 0.07);
        }
        CK(hipFree(dH));
    }
    // SWIGLU: rows of W are w1|w3 interleaved in blocks of 16 -> hidden = N/2
    {
        const int Hd = N / 2;
        bf16_t* dP = dalloc<bf16_t>((size_t)((M + 15) / 16) * 16 * Hd);
        for (int cfg : {240, 421, 211}) {
            CK(hipMemset(dP, 0, (size_t)((M + 15) / 16) * 16 * Hd * 2));
            GemmDP p; memset(&p, 0, sizeof(p)); p.W = dW; p.X = dX; p.M = M; p.N = N; p.K = K; p.outp = dP;
            car_launch_dec_gemm_cfg(&p, EPI_SWIGLU, cfg, 0); CK(hipDeviceSynchronize());
            auto o = d2h(dP, (size_t)((M + 15) / 16) * 16 * Hd);
            double e = 0;
            for (int m = 0; m < M; ++m) for (int hd = 0; hd < Hd; ++hd) {
                const int na = (hd / 16) * 32 + hd % 16, nc = na + 16;
                const float a = rb((float)ref[(size_t)m * N + na]), c = rb((float)ref[(size_t)m * N + nc]);
                const float want = rb(rb(a / (1.0f + expf(-a))) * c);
                e = std::max(e, std::fabs((double)bf2f(o[xp_off(m, hd, Hd)]) - (double)want));
            }
            char nm[96]; snprintf(nm, sizeof(nm), "dec_gemm SWIGLU cfg %d (packed out)", cfg); report(nm, e, 0.07);
        }
        CK(hipFree(dP));
    }
    CK(hipFree(dX)); CK(hipFree(dW)); CK(hipFree(dO));
}

// ------------------------------------------------------------------------------------------------ QKV epilogue + attention
struct AttnCase { int b, H, T, pos, nsplit, packed_out, maskmode, variant; };

static void test_qkv_attn(const AttnCase& c) {
    const int b = c.b, H = c.H, dim = H * 64, K = 96, T = c.T, pos = c.pos, S_max = ((pos + 1 + 7) / 8) * 8, SA = (S_max + 31) / 32 * 32;
    // history K/V (positions < pos) given directly; the new token's q,k,v come out of the QKV epilogue
    std::vector<float> Kh((size_t)b * H * pos * 64), Vh(Kh.size());
    for (auto& v : Kh) v = rb(frand()); for (auto& v : Vh) v = rb(frand());
    std::vector<bf16_t> kc((size_t)b * H * SA * 64), vc(kc.size());
    for (auto& v : kc) v = 0x7fc0; for (auto& v : vc) v = 0;          // never-written K slots hold NaN: must never leak; V slots hold 0 (the engine zero-fills at allocation)
    for (int i = 0; i < b; ++i) for (int h = 0; h < H; ++h) for (int p = 0; p < pos; ++p) for (int d = 0; d < 64; ++d) {
        const size_t sb = ((size_t)i * H + h) * SA * 64, src = (((size_t)i * H + h) * pos + p) * 64 + d;
        kc[sb + k_off(p, d)] = f2bf(Kh[src]); vc[sb + v_off(p, d)] = f2bf(Vh[src]);
    }
    std::vector<float> X((size_t)b * K), W((size_t)3 * dim * K);
    for (auto& v : X) v = rb(frand()); for (auto& v : W) v = rb(frand() * 0.2f);
    std::vector<float> rope((size_t)(pos + 1) * 64);
    for (int p = 0; p <= pos; ++p) for (int i = 0; i < 32; ++i) { const float ang = 0.01f * p * (i + 1); rope[((size_t)p * 32 + i) * 2] = p < T ? 0.f : cosf(ang); rope[((size_t)p * 32 + i) * 2 + 1] = p < T ? 0.f : sinf(ang); }
    std::vector<unsigned char> mask((size_t)b * T, 1);
    for (int i = 0; i < b; ++i) {
        const int L = 3 + (i * 7) % (T - 3);                            // valid length; left padded
        for (int t = 0; t < T; ++t) mask[(size_t)i * T + t] = t >= T - L;
        if (c.maskmode == 2) for (int t = 0; t < T; ++t) mask[(size_t)i * T + t] = ((t * 7 + i) % 3) != 0;     // arbitrary pattern
    }
    // reference: qkv = rnd(X W^T); rope; cache write; attention in fp64 over valid positions
    std::vector<float> qkv((size_t)b * 3 * dim);
    for (int i = 0; i < b; ++i) for (int n = 0; n < 3 * dim; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)X[(size_t)i * K + k] * W[(size_t)n * K + k]; qkv[(size_t)i * 3 * dim + n] = rb((float)s); }
    std::vector<float> qr((size_t)b * dim), kr((size_t)b * dim), want((size_t)b * dim), want_nm((size_t)b * dim);
    for (int i = 0; i < b; ++i) for (int h = 0; h < H; ++h) {
        for (int pr = 0; pr < 32; ++pr) {
            const float cs = rope[((size_t)pos * 32 + pr) * 2], sn = rope[((size_t)pos * 32 + pr) * 2 + 1];
            const float* q = &qkv[(size_t)i * 3 * dim + h * 64]; const float* k = &qkv[(size_t)i * 3 * dim + dim + h * 64];
            qr[(size_t)i * dim + h * 64 + 2 * pr] = rb(q[2 * pr] * cs - q[2 * pr + 1] * sn); qr[(size_t)i * dim + h * 64 + 2 * pr + 1] = rb(q[2 * pr + 1] * cs + q[2 * pr] * sn);
            kr[(size_t)i * dim + h * 64 + 2 * pr] = rb(k[2 * pr] * cs - k[2 * pr + 1] * sn); kr[(size_t)i * dim + h * 64 + 2 * pr + 1] = rb(k[2 * pr + 1] * cs + k[2 * pr] * sn);
        }
        std::vector<double> s(pos + 1); double mx = -1e300;
        for (int p = 0; p <= pos; ++p) {
            const bool ok = !(c.maskmode && p < T && !mask[(size_t)i * T + p]);
            double a = 0;
            for (int d = 0; d < 64; ++d) { const float kk = p < pos ? Kh[(((size_t)i * H + h) * pos + p) * 64 + d] : kr[(size_t)i * dim + h * 64 + d]; a += (double)qr[(size_t)i * dim + h * 64 + d] * kk; }
            s[p] = ok ? a * 0.125 : -1e300; mx = std::max(mx, s[p]);
        }
        double L = 0; std::vector<double> o(64, 0.0);
        for (int p = 0; p <= pos; ++p) if (s[p] > -1e299) {
            const double w = exp(s[p] - mx); L += w;
            for (int d = 0; d < 64; ++d) { const float vv = p < pos ? Vh[(((size_t)i * H + h) * pos + p) * 64 + d] : qkv[(size_t)i * 3 * dim + 2 * dim + h * 64 + d]; o[d] += w * vv; }
        }
        for (int d = 0; d < 64; ++d) want[(size_t)i * dim + h * 64 + d] = (float)(o[d] / L);
        {   // diagnosis only: the same without the text-pad mask
            double mx2 = -1e300; std::vector<double> s2(pos + 1);
            for (int p = 0; p <= pos; ++p) { double a = 0; for (int d = 0; d < 64; ++d) { const float kk = p < pos ? Kh[(((size_t)i * H + h) * pos + p) * 64 + d] : kr[(size_t)i * dim + h * 64 + d]; a += (double)qr[(size_t)i * dim + h * 64 + d] * kk; } s2[p] = a * 0.125; mx2 = std::max(mx2, s2[p]); }
            double L2 = 0; std::vector<double> o2(64, 0.0);
            for (int p = 0; p <= pos; ++p) { const double w = exp(s2[p] - mx2); L2 += w; for (int d = 0; d < 64; ++d) { const float vv = p < pos ? Vh[(((size_t)i * H + h) * pos + p) * 64 + d] : qkv[(size_t)i * 3 * dim + 2 * dim + h * 64 + d]; o2[d] += w * vv; } }
            for (int d = 0; d < 64; ++d) want_nm[(size_t)i * dim + h * 64 + d] = (float)(o2[d] / L2);
        }
    }
    auto xp = pack_rows(X, b, K), wp = pack_rows(W, 3 * dim, K);
    bf16_t* dX = dalloc<bf16_t>(xp.size()); h2d(dX, xp);
    bf16_t* dW = dalloc<bf16_t>(wp.size()); h2d(dW, wp);
    bf16_t* dK = dalloc<bf16_t>(kc.size()); h2d(dK, kc);
    bf16_t* dV = dalloc<bf16_t>(vc.size()); h2d(dV, vc);
    float* dR = dalloc<float>(rope.size()); h2d(dR, rope);
    unsigned char* dM = dalloc<unsigned char>(mask.size()); h2d(dM, mask);
    int* dPos = dalloc<int>(1); CK(hipMemcpy(dPos, &pos, 4, hipMemcpyHostToDevice));
    bf16_t* dQ = dalloc<bf16_t>((size_t)b * dim);
    const size_t osz = (size_t)((b + 15) / 16) * 16 * dim;
    bf16_t* dO = dalloc<bf16_t>(osz); CK(hipMemset(dO, 0, osz * 2));
    float* dPart = dalloc<float>((size_t)b * H * c.nsplit * 66);
    GemmDP g; memset(&g, 0, sizeof(g)); g.W = dW; g.X = dX; g.M = b; g.N = 3 * dim; g.K = K; g.qout = dQ; g.kc = dK; g.vc = dV; g.rope = dR; g.pos = dPos; g.H = H; g.SA = SA; g.dim = dim;
    const int cfg = b > 16 ? 420 : 410;
    if (car_launch_dec_gemm_cfg(&g, EPI_QKV, cfg, 0)) { printf("qkv cfg rejected\n"); ++g_fail; }
    Attn2P a; memset(&a, 0, sizeof(a)); a.q = dQ; a.kc = dK; a.vc = dV; a.pos = dPos; a.mask = c.maskmode ? dM : nullptr; a.out = dO; a.part = dPart;
    a.H = H; a.SA = SA; a.T = T; a.dim = dim; a.nsplit = c.nsplit; a.out_packed = c.packed_out;
    int* dJ = dalloc<int>(b);
    if (c.maskmode && (c.variant % 2 == 0 || c.variant == 41)) { car_launch_mask_first_valid(dM, dJ, b, T, 0); a.jmin = dJ; }
    if (c.variant == 162 && (b * H) % 8 == 0) { a.pf_wgs = 24; a.pf_p0 = dW; a.pf_b0 = (unsigned)(wp.size() * 2); }      // run-ahead helpers ride along (they only read)     // both jmin sources get exercised
    car_launch_dec_attn2_var(&a, b, c.variant, 0, 0);
    CK(hipDeviceSynchronize());
    // cache rows written by the epilogue
    auto kc2 = d2h(dK, kc.size()), vc2 = d2h(dV, vc.size()); auto q2 = d2h(dQ, (size_t)b * dim);
    double ek = 0, ev = 0, eq = 0;
    for (int i = 0; i < b; ++i) for (int h = 0; h < H; ++h) for (int d = 0; d < 64; ++d) {
        const size_t sb = ((size_t)i * H + h) * SA * 64;
        ek = std::max(ek, std::fabs((double)bf2f(kc2[sb + k_off(pos, d)]) - kr[(size_t)i * dim + h * 64 + d]));
        ev = std::max(ev, std::fabs((double)bf2f(vc2[sb + v_off(pos, d)]) - qkv[(size_t)i * 3 * dim + 2 * dim + h * 64 + d]));
        eq = std::max(eq, std::fabs((double)bf2f(q2[((size_t)i * H + h) * 64 + d]) - 0.125 * qr[(size_t)i * dim + h * 64 + d]));
    }
    char nm[128];
    snprintf(nm, sizeof(nm), "QKV epilogue b=%d H=%d pos=%d: K row", b, H, pos); report(nm, ek, 0.04);
    snprintf(nm, sizeof(nm), "QKV epilogue b=%d H=%d pos=%d: V row", b, H, pos); report(nm, ev, 0.04);
    snprintf(nm, sizeof(nm), "QKV epilogue b=%d H=%d pos=%d: q", b, H, pos); report(nm, eq, 0.005);
    auto o = d2h(dO, osz);
    double e = 0;
    for (int i = 0; i < b; ++i) for (int k = 0; k < dim; ++k) {
        const float got = bf2f(o[c.packed_out ? xp_off(i, k, dim) : (size_t)i * dim + k]);
        e = std::max(e, std::fabs((double)got - want[(size_t)i * dim + k]));
    }
    snprintf(nm, sizeof(nm), "dec_attn2<%d> b=%d H=%d T=%d pos=%d nsplit=%d packed=%d mask=%d", c.variant, b, H, T, pos, c.nsplit, c.packed_out, c.maskmode);
    report(nm, e, 0.02);
    if (e > 0.02) {
        for (int i = 0; i < b && i < 4; ++i) for (int h = 0; h < H; ++h) {
            double ei = 0, en = 0;
            for (int d = 0; d < 64; ++d) { const int k = h * 64 + d; const float got = bf2f(o[c.packed_out ? xp_off(i, k, dim) : (size_t)i * dim + k]);
                ei = std::max(ei, std::fabs((double)got - want[(size_t)i * dim + k])); en = std::max(en, std::fabs((double)got - want_nm[(size_t)i * dim + k])); }
            int first = T; for (int t = 0; t < T; ++t) if (mask[(size_t)i * T + t]) { first = t; break; }
            printf("    seq %d head %d: err vs masked ref %.3e, vs unmasked ref %.3e (first valid text pos %d)\n", i, h, ei, en, first);
        }
    }
    for (void* p : {(void*)dX, (void*)dW, (void*)dK, (void*)dV, (void*)dR, (void*)dM, (void*)dPos, (void*)dQ, (void*)dO, (void*)dPart, (void*)dJ}) CK(hipFree(p));
}

// ------------------------------------------------------------------------------------------------ rmsnorm2
static void test_norm() {
    const int rows = 37, D = 256;
    std::vector<bf16_t> h((size_t)rows * D), w(D); for (auto& v : h) v = f2bf(frand() * 3.f); for (auto& v : w) v = f2bf(1.f + 0.1f * frand());
    bf16_t* dH = dalloc<bf16_t>(h.size()); h2d(dH, h); bf16_t* dWn = dalloc<bf16_t>(D); h2d(dWn, w);
    const size_t osz = (size_t)((rows + 15) / 16) * 16 * D; bf16_t* dX = dalloc<bf16_t>(osz); CK(hipMemset(dX, 0, osz * 2));
    Norm2P p; memset(&p, 0, sizeof(p)); p.h_in = dH; p.xn = dX; p.w = dWn; p.D = D; p.eps = 1e-5f;
    car_launch_rmsnorm2(&p, rows, 0); CK(hipDeviceSynchronize());
    auto o = d2h(dX, osz); double e = 0;
    for (int r = 0; r < rows; ++r) {
        double ss = 0; for (int k = 0; k < D; ++k) ss += (double)bf2f(h[(size_t)r * D + k]) * bf2f(h[(size_t)r * D + k]);
        const float rstd = 1.0f / sqrtf((float)(ss / D) + 1e-5f);
        for (int k = 0; k < D; ++k) e = std::max(e, std::fabs((double)bf2f(o[xp_off(r, k, D)]) - (double)rb(rb(bf2f(h[(size_t)r * D + k]) * rstd) * bf2f(w[k]))));
    }
    report("rmsnorm2 (packed xn)", e, 0.04);
    CK(hipFree(dH)); CK(hipFree(dWn)); CK(hipFree(dX));
}

// ------------------------------------------------------------------------------------------------ fused-norm GEMM (M <= 16)
static void test_gemm_norm() {
    const int M = 5, D = 256, N = 512;
    std::vector<bf16_t> h((size_t)M * D), w(D), ctrl((size_t)M * 3 * D); for (auto& v : h) v = f2bf(frand() * 3.f); for (auto& v : w) v = f2bf(1.f + 0.1f * frand()); for (auto& v : ctrl) v = f2bf(frand());
    std::vector<float> W((size_t)N * D); for (auto& v : W) v = rb(frand() * 0.1f);
    auto wp = pack_rows(W, N, D);
    bf16_t* dH = dalloc<bf16_t>(h.size()); h2d(dH, h); bf16_t* dWn = dalloc<bf16_t>(D); h2d(dWn, w); bf16_t* dC = dalloc<bf16_t>(ctrl.size()); h2d(dC, ctrl);
    bf16_t* dW = dalloc<bf16_t>(wp.size()); h2d(dW, wp);
    bf16_t* dX = dalloc<bf16_t>((size_t)16 * D); bf16_t* dH1 = dalloc<bf16_t>(h.size()); bf16_t* dH2 = dalloc<bf16_t>(h.size());
    float* dO1 = dalloc<float>((size_t)M * N); float* dO2 = dalloc<float>((size_t)M * N);
    int pos = 41; int* dPos = dalloc<int>(1); CK(hipMemcpy(dPos, &pos, 4, hipMemcpyHostToDevice));      // control token index pos - T + 1 = 2 with T = 40
    for (int add = 0; add < 2; ++add) for (int cfg : {110, 111, 210, 411}) {
        // reference path: rmsnorm2 -> packed xn -> dec_gemm
        Norm2P np; memset(&np, 0, sizeof(np)); np.h_in = dH; np.xn = dX; np.w = dWn; np.D = D; np.eps = 1e-5f; np.h_out = dH1;
        if (add) { np.add = 1; np.ctrl = dC; np.pos = dPos; np.T = 40; np.n_tok = 3; np.cs = 0.6f; }
        car_launch_rmsnorm2(&np, M, 0);
        GemmDP p; memset(&p, 0, sizeof(p)); p.W = dW; p.X = dX; p.M = M; p.N = N; p.K = D; p.outf = dO1;
        car_launch_dec_gemm_cfg(&p, EPI_LOGITS, cfg, 0);
        GemmDP q; memset(&q, 0, sizeof(q)); q.W = dW; q.M = M; q.N = N; q.K = D; q.outf = dO2; q.nh_in = dH; q.nw = dWn; q.neps = 1e-5f; q.nh_out = dH2; q.pos = dPos;
        if (add) { q.nadd = 1; q.nctrl = dC; q.nT = 40; q.n_tok = 3; q.ncs = 0.6f; }
        CK(hipMemset(dO2, 0xff, (size_t)M * N * 4));
        if (car_launch_dec_gemm_cfg(&q, EPI_LOGITS, cfg, 0)) { printf("fused-norm cfg %d rejected\n", cfg); ++g_fail; continue; }
        CK(hipDeviceSynchronize());
        auto o1 = d2h(dO1, (size_t)M * N), o2 = d2h(dO2, (size_t)M * N); auto h1 = d2h(dH1, h.size()), h2 = d2h(dH2, h.size());
        double e = 0; for (size_t i = 0; i < o1.size(); ++i) e = std::max(e, std::fabs((double)o1[i] - o2[i]));
        double eh = 0; for (size_t i = 0; i < h1.size(); ++i) eh = std::max(eh, std::fabs((double)bf2f(h1[i]) - bf2f(h2[i])));
        char nm[96]; snprintf(nm, sizeof(nm), "dec_gemm fused-norm cfg %d add=%d vs rmsnorm2 + dec_gemm (bit-exact)", cfg, add); report(nm, e + eh, 0.0);
    }
    for (void* q : {(void*)dH, (void*)dWn, (void*)dC, (void*)dW, (void*)dX, (void*)dH1, (void*)dH2, (void*)dO1, (void*)dO2, (void*)dPos}) CK(hipFree(q));
}


// ------------------------------------------------------------------------------------------------ round 6: normalise-on-the-fly GEMM, 16-wave tiles, small-batch attention
static void test_gemm_normx() {
    for (int M : {2, 5, 16}) {
        const int D = 1280, N = 512, NP = D / 16;
        std::vector<bf16_t> h((size_t)M * D), w(D); for (auto& v : h) v = f2bf(frand() * 3.f); for (auto& v : w) v = f2bf(1.f + 0.1f * frand());
        std::vector<float> W((size_t)N * D); for (auto& v : W) v = rb(frand() * 0.1f);
        std::vector<float> ssq((size_t)M * NP);
        for (int m = 0; m < M; ++m) for (int pr = 0; pr < NP; ++pr) { float a = 0.f; for (int k = 0; k < 16; ++k) { const float x = bf2f(h[(size_t)m * D + pr * 16 + k]); a += x * x; } ssq[(size_t)m * NP + pr] = a; }
        // the kernel's fold: lane q4 adds the four floats of chunks q4, q4 + 4, ... in order, then (s0 + s1) + (s2 + s3)
        std::vector<double> ref((size_t)M * N);
        for (int m = 0; m < M; ++m) {
            float sq[4] = {0, 0, 0, 0};
            for (int q = 0; q < 4; ++q) for (int ci = q; ci < NP / 4; ci += 4) for (int e = 0; e < 4; ++e) sq[q] += ssq[(size_t)m * NP + ci * 4 + e];
            const float tot = (sq[0] + sq[1]) + (sq[2] + sq[3]);
            const float rstd = 1.0f / sqrtf(tot / D + 1e-5f);
            std::vector<float> x(D); for (int k = 0; k < D; ++k) x[k] = rb(rb(bf2f(h[(size_t)m * D + k]) * rstd) * bf2f(w[k]));
            for (int n = 0; n < N; ++n) { double a = 0; for (int k = 0; k < D; ++k) a += (double)x[k] * W[(size_t)n * D + k]; ref[(size_t)m * N + n] = a; }
        }
        auto wp = pack_rows(W, N, D);
        bf16_t* dH = dalloc<bf16_t>(h.size()); h2d(dH, h); bf16_t* dWn = dalloc<bf16_t>(D); h2d(dWn, w); bf16_t* dW = dalloc<bf16_t>(wp.size()); h2d(dW, wp);
        float* dS = dalloc<float>(ssq.size()); h2d(dS, ssq); float* dO = dalloc<float>((size_t)M * N);
        for (int cfg : {111, 110, 211, 411, 120}) {
            CK(hipMemset(dO, 0xff, (size_t)M * N * 4));
            GemmDP q; memset(&q, 0, sizeof(q)); q.W = dW; q.M = M; q.N = N; q.K = D; q.outf = dO; q.nh_in = dH; q.nw = dWn; q.neps = 1e-5f; q.ssq_in = dS; q.ssq_np = NP; q.w_nt = 1;
            if (car_launch_dec_gemm_cfg(&q, EPI_LOGITS, cfg, 0)) { printf("normx cfg %d rejected\n", cfg); ++g_fail; continue; }
            CK(hipDeviceSynchronize());
            auto o = d2h(dO, (size_t)M * N);
            double e = 0; for (size_t i = 0; i < o.size(); ++i) e = std::max(e, std::fabs((double)o[i] - (double)rb((float)ref[i])));
            char nm[96]; snprintf(nm, sizeof(nm), "dec_gemm NORM==2/3 (on-the-fly RMSNorm) cfg %d M=%d K=%d", cfg, M, D); report(nm, e, 0.3);      // |x| up to ~40: one bf16 ulp = 0.25
            if (cfg == 111 || cfg == 211 || cfg == 411) {      // these ran the staged form (NORM == 3): the register form (NORM == 2) of the same tile must give the same bits
                float* dO2 = dalloc<float>((size_t)M * N); CK(hipMemset(dO2, 0xff, (size_t)M * N * 4));
                GemmDP q2 = q; q2.outf = dO2;
                if (cfg == 111) launch_gemm_ij<1, 1, 8, 0, 2>(q2, EPI_LOGITS, 0); else if (cfg == 211) launch_gemm_ij<2, 1, 8, 0, 2>(q2, EPI_LOGITS, 0); else launch_gemm_ij<4, 1, 8, 0, 2>(q2, EPI_LOGITS, 0);
                CK(hipDeviceSynchronize());
                auto o2 = d2h(dO2, (size_t)M * N);
                double eb = 0; for (size_t i = 0; i < o.size(); ++i) eb = std::max(eb, std::fabs((double)o[i] - (double)o2[i]));
                snprintf(nm, sizeof(nm), "   ... staged (NORM==3) vs register (NORM==2) form, cfg %d M=%d: bit-exact", cfg, M); report(nm, eb, 0.0);
                CK(hipFree(dO2));
            }
        }
        for (void* q : {(void*)dH, (void*)dWn, (void*)dW, (void*)dS, (void*)dO}) CK(hipFree(q));
    }
}
static void test_gemm_w16() {      // cfg 112 (16 waves) against cfg 111: RESID with the sum-of-squares partials, with and without run-ahead helper workgroups — bit for bit
    for (int K : {1280, 3584}) {
        const int M = 7, N = 256;
        std::vector<float> X((size_t)M * K), W((size_t)N * K); for (auto& v : X) v = rb(frand()); for (auto& v : W) v = rb(frand() * 0.05f);
        auto xp = pack_rows(X, M, K), wp = pack_rows(W, N, K);
        std::vector<bf16_t> h0((size_t)M * N); for (auto& v : h0) v = f2bf(frand() * 2.f);
        bf16_t* dX = dalloc<bf16_t>(xp.size()); h2d(dX, xp); bf16_t* dW = dalloc<bf16_t>(wp.size()); h2d(dW, wp);
        bf16_t* dH1 = dalloc<bf16_t>(h0.size()); bf16_t* dH2 = dalloc<bf16_t>(h0.size()); float* dS1 = dalloc<float>((size_t)M * N / 16); float* dS2 = dalloc<float>((size_t)M * N / 16);
        unsigned* junk = dalloc<unsigned>((size_t)(8 << 20) / 4); CK(hipMemset(junk, 1, 8 << 20));
        for (int pf : {0, 64}) {
            h2d(dH1, h0); h2d(dH2, h0);
            GemmDP p; memset(&p, 0, sizeof(p)); p.W = dW; p.X = dX; p.M = M; p.N = N; p.K = K; p.w_nt = 1; p.ssq_ld = N / 16;
            p.h = dH1; p.ssq_out = dS1; car_launch_dec_gemm_cfg(&p, EPI_RESID, 111, 0);
            p.h = dH2; p.ssq_out = dS2; p.pf_wgs = pf; p.pf_p0 = junk; p.pf_b0 = 8u << 20; p.pf_p1 = (const char*)junk + 12345 * 128; p.pf_b1 = 3u << 20;
            if (car_launch_dec_gemm_cfg(&p, EPI_RESID, 112, 0)) { printf("cfg 112 rejected\n"); ++g_fail; }
            CK(hipDeviceSynchronize());
            auto a = d2h(dH1, h0.size()), b2 = d2h(dH2, h0.size()); auto s1 = d2h(dS1, (size_t)M * N / 16), s2 = d2h(dS2, (size_t)M * N / 16);
            double e = 0, es = 0; for (size_t i = 0; i < a.size(); ++i) e = std::max(e, std::fabs((double)bf2f(a[i]) - bf2f(b2[i])));
            for (size_t i = 0; i < s1.size(); ++i) es = std::max(es, std::fabs((double)s1[i] - s2[i]) / std::max(1.0, (double)std::fabs(s1[i])));
            char nm[128]; snprintf(nm, sizeof(nm), "dec_gemm RESID cfg 112 (16 waves, %d helper workgroups) vs cfg 111, K=%d: h", pf, K); report(nm, e, 0.07);
            snprintf(nm, sizeof(nm), "   ... sum-of-squares partials (relative)"); report(nm, es, 2e-2);
        }
        for (void* q : {(void*)dX, (void*)dW, (void*)dH1, (void*)dH2, (void*)dS1, (void*)dS2, (void*)junk}) CK(hipFree(q));
    }
}

// ------------------------------------------------------------------------------------------------ timing
__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        // two bf16 in [-1,1): exponent <= 126
        const unsigned a = (x & 0x807fu) | (((x >> 7) & 0x3f) + 64) << 7, b2 = ((x >> 16) & 0x807fu) | ((((x >> 23) & 0x3f) + 64) << 7);
        p[i] = a | (b2 << 16); }
}

static float time_launches(int iters, const std::function<void(int)>& f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) f(i);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1000.f / iters;
}

static void bench_gemm(bool quick) {
    struct Shape { const char* name; int N, K, epi; };
    const Shape shapes[] = {{"wqkv", 3840, 1280, EPI_QKV}, {"wo", 1280, 1280, EPI_RESID}, {"w13", 7168, 1280, EPI_SWIGLU}, {"w2", 1280, 3584, EPI_RESID}, {"logits", 16384, 1280, EPI_LOGITS}};
    const int NL = 24;                                         // distinct weight copies cycled through: 24 x (10..42 MB) > the 256 MiB MALL
    const int H = 20, dim = 1280, SA = 1152;
    std::vector<int> Ms = quick ? std::vector<int>{256} : std::vector<int>{256, 128, 64, 16, 2};
    bf16_t* dKV = dalloc<bf16_t>((size_t)2 * 256 * H * SA * 64);
    float* dRope = dalloc<float>((size_t)1200 * 64); CK(hipMemset(dRope, 0, 1200 * 64 * 4));
    int hp = 700; int* dPos = dalloc<int>(1); CK(hipMemcpy(dPos, &hp, 4, hipMemcpyHostToDevice));
    for (const Shape& s : shapes) {
        const size_t wsz = (size_t)s.N * s.K;
        bf16_t* dW = dalloc<bf16_t>(wsz * NL);
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned*)dW, wsz * NL / 2, 12345u);
        for (int M : Ms) {
            const int Mb = (M + 15) / 16;
            bf16_t* dX = dalloc<bf16_t>((size_t)Mb * 16 * s.K);
            hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (unsigned*)dX, (size_t)Mb * 16 * s.K / 2, 777u);
            bf16_t* dH = dalloc<bf16_t>((size_t)M * s.N); CK(hipMemset(dH, 0, (size_t)M * s.N * 2));
            bf16_t* dP = dalloc<bf16_t>((size_t)Mb * 16 * (s.N / 2));
            float* dF = dalloc<float>((size_t)M * s.N);
            bf16_t* dQ = dalloc<bf16_t>((size_t)M * dim);
            const int pick = car_pick_gemm_cfg(M, s.N, s.K, s.epi);
            float best = 1e30f; int bestc = 0;
            std::string line;
            for (int cfg : ALL_CFG) {
                const int I = cfg / 100, J = (cfg / 10) % 10;
                if (s.epi == EPI_SWIGLU && I < 2) continue;
                if (J > 1 && J / 2 >= Mb) continue;              // tile taller than the batch
                if (s.N % (16 * I)) continue;
                GemmDP p; memset(&p, 0, sizeof(p)); p.X = dX; p.M = M; p.N = s.N; p.K = s.K; p.h = dH; p.outp = dP; p.outf = dF;
                p.qout = dQ; p.kc = dKV; p.vc = dKV + (size_t)256 * H * SA * 64; p.rope = dRope; p.pos = dPos; p.H = H; p.SA = SA; p.dim = dim;
                p.w_nt = (Mb + J - 1) / J == 1;
                const float us = time_launches(quick ? 48 : 96, [&](int it) { GemmDP q = p; q.W = dW + wsz * (it % NL); car_launch_dec_gemm_cfg(&q, s.epi, cfg, 0); });
                CK(hipGetLastError());
                char b[64]; snprintf(b, sizeof(b), " %d:%.1f", cfg, us); line += b;
                if (us < best) { best = us; bestc = cfg; }
            }
            printf("GEMM %-6s M=%-3d N=%-5d K=%-4d W=%.1fMB  best cfg %d %.2f us (%.2f TB/s of weights)  heuristic %d |%s\n", s.name, M, s.N, s.K, wsz * 2 / 1e6, bestc, best,
                   wsz * 2 / 1e6 / best, pick, line.c_str());
            fflush(stdout);
            for (void* q : {(void*)dX, (void*)dH, (void*)dP, (void*)dF, (void*)dQ}) CK(hipFree(q));
        }
        CK(hipFree(dW));
    }
    CK(hipFree(dKV)); CK(hipFree(dRope)); CK(hipFree(dPos));
}

static void bench_attn(bool quick) {
    const int H = 20, dim = 1280, T = 120, S_max = 1144, SA = 1152;
    struct C { int b, nsplit; };
    std::vector<C> cs = quick ? std::vector<C>{{256, 1}} : std::vector<C>{{256, 1}, {128, 1}, {64, 1}, {16, 4}, {2, 16}, {2, 8}};
    const int NLAY = 2;                                       // K+V of one layer at b=256 is 1.5 GB >> MALL; alternate two anyway
    const size_t per = (size_t)256 * H * SA * 64;
    bf16_t* dKV = dalloc<bf16_t>(per * 2 * NLAY);
    hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, (unsigned*)dKV, per * 2 * NLAY / 2, 999u);
    bf16_t* dQ = dalloc<bf16_t>((size_t)256 * dim);
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, (unsigned*)dQ, (size_t)256 * dim / 2, 5u);
    bf16_t* dO = dalloc<bf16_t>((size_t)256 * dim);
    float* dPart = dalloc<float>((size_t)256 * H * 16 * 66);
    std::vector<unsigned char> mask((size_t)256 * T, 0);
    for (int i = 0; i < 256; ++i) { const int L = 8 + (i * 13) % 33; for (int t = T - L; t < T; ++t) mask[(size_t)i * T + t] = 1; }    // U{8..40} valid, left padded (synth.text_embeddings)
    unsigned char* dM = dalloc<unsigned char>(mask.size()); h2d(dM, mask);
    int* dPos = dalloc<int>(1);
    CK(hipDeviceSynchronize());
    int* dJ = dalloc<int>(256); car_launch_mask_first_valid(dM, dJ, 256, T, 0);
    for (const C& c : cs) for (int pos : {127, 631, 1143}) {
        CK(hipMemcpy(dPos, &pos, 4, hipMemcpyHostToDevice));
        double rows = 0; for (int i = 0; i < c.b; ++i) { const int L = 8 + (i * 13) % 33; rows += pos + 1 - (T - L); }
        const double bytes = rows * H * 256.0;
        printf("ATTN b=%-3d nsplit=%-2d pos=%-4d %.1f MB (valid rows) |", c.b, c.nsplit, pos, bytes / 1e6);
        for (int variant : {41, 40, 21, 20, 141}) {
            Attn2P a; memset(&a, 0, sizeof(a)); a.q = dQ; a.pos = dPos; a.mask = dM; a.out = dO; a.part = dPart; a.H = H; a.SA = SA; a.T = T; a.dim = dim; a.nsplit = c.nsplit; a.out_packed = 1;
            a.jmin = variant == 141 ? nullptr : dJ;             // 141: variant 41 with the in-kernel mask scan
            const float us = time_launches(quick ? 30 : 60, [&](int it) { Attn2P q = a; q.kc = dKV + per * 2 * (it % NLAY); q.vc = q.kc + per; car_launch_dec_attn2_var(&q, c.b, variant % 100, 0, 0); });
            printf("  v%d: %.1f us %.2f TB/s", variant, us, bytes / 1e6 / us);
        }
        printf("\n"); fflush(stdout);
    }
    CK(hipFree(dJ));
    (void)S_max;
    for (void* q : {(void*)dKV, (void*)dQ, (void*)dO, (void*)dPart, (void*)dM, (void*)dPos}) CK(hipFree(q));
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    const bool noperf = argc > 1 && !strcmp(argv[1], "check");
    test_gemm();
    test_norm();
    test_gemm_norm();
    test_gemm_normx();
    test_gemm_w16();
    const AttnCase cases[] = {
        {3, 2, 40, 40, 1, 0, 1, 0}, {3, 2, 40, 63, 1, 1, 1, 0}, {3, 2, 40, 64, 1, 1, 1, 0}, {3, 2, 40, 250, 1, 1, 1, 0}, {3, 2, 40, 250, 4, 1, 1, 0}, {20, 2, 40, 131, 2, 0, 1, 0},
        {3, 2, 40, 97, 1, 1, 2, 0}, {3, 2, 40, 97, 4, 1, 2, 0}, {2, 1, 1, 1, 1, 1, 0, 0}, {2, 1, 1, 33, 16, 1, 0, 0}, {17, 3, 120, 600, 1, 1, 1, 0}, {5, 2, 120, 1143, 1, 1, 1, 0},
    };
    for (int variant : {41, 40, 21, 20}) for (auto c : cases) { c.variant = variant; test_qkv_attn(c); }
    // one-launch 16-wave forms (small batch): the round-3 kernel and round 6's two-blocks-in-flight kernel, nsplit 1 only, incl. a 1656-position (MR) cache
    for (int variant : {160, 162}) for (auto c : cases) { if (c.nsplit != 1) continue; c.variant = variant; test_qkv_attn(c); }
    for (int variant : {160, 162}) { AttnCase c = {2, 2, 120, 1650, 1, 1, 1, variant}; test_qkv_attn(c); AttnCase c2 = {8, 4, 120, 1023, 1, 1, 1, variant}; test_qkv_attn(c2); }
    printf("== correctness: %d failure(s)\n", g_fail);
    fflush(stdout);
    const bool only_gemm = argc > 1 && !strcmp(argv[1], "gemm");
    if (!noperf) { if (!only_gemm) bench_attn(quick); bench_gemm(quick); }
    return g_fail ? 1 : 0;
}
