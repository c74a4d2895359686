// experiments/ws3_check.hip (round 5, EXPERIMENT) — does the exact mode's round-5 finding carry over to the bf16 headline step?
// There, a chain's linears overlap another chain's attention once (a) the attention leaves wave slots free (12-wave workgroups) and (b) the linears are 4-wave
// workgroups fed from LDS rings.  Here: ONE half-period of the two-chain bf16 step at the bench shape (384 rows per chain, position 631) on two streams —
// attention as the stock grid or as a RESIDENT grid of R workgroups per CU (dec_attn2's persistent form: R = 6 leaves 8 wave slots per CU), linears as the
// product's dec_gemm or as the LDS-DMA-ring kernel of experiments/decode5_lds_dma_gemm.hip (64 x 64 tiles, 4 waves, 5 stages; 128 x 64, 8 waves).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I controlar_amd/csrc experiments/ws3_check.hip -o experiments/ws3_check && experiments/ws3_check
#include "../controlar_amd/csrc/decode2.hip"
#define CAR_GEMMDP_DEFINED
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
static void linears(const Layer& L, int mode /* 0 dec_gemm, 10 + c: decode5 LDS-ring cfg c */, hipStream_t st) {
    auto gemm = [&](const bf16_t* W, const bf16_t* X, int N, int K, int epi, GemmDP p) {
        p.W = W; p.X = X; p.M = L.M; p.N = N; p.K = K;
        const int cfg = car_pick_gemm_cfg(L.M, N, K, epi); const int J = (cfg / 10) % 10, Mb = (L.M + 15) / 16;
        p.w_nt = (Mb + J - 1) / J == 1;
        if (mode == 0 || car_launch_dec_gemm_lds(&p, epi, mode - 10, st)) car_launch_dec_gemm_cfg(&p, epi, cfg, st);
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
    auto attn_form = [&](int R) {       // R = 0: the stock grid (one 4-wave workgroup per (sequence, head)); R > 0: resident grid of R workgroups per CU
        return [=](int it, hipStream_t st) {
            Attn2P a; memset(&a, 0, sizeof(a)); a.q = qa; a.pos = dPos; a.mask = dM; a.jmin = dJ; a.out = oa; a.H = H; a.SA = SA; a.T = T; a.dim = D; a.nsplit = 1; a.out_packed = 1;
            a.kc = dKV + kvper * 2 * (it % NKV); a.vc = a.kc + kvper;
            if (R > 0) { const long items = (long)M * H, cap = 256L * R, per = (items + cap - 1) / cap; a.n_seq = M; a.pgrid = (int)((items + per - 1) / per); }
            car_launch_dec_attn2_var(&a, M, 40, 0, st);
        };
    };
    for (int mode : {0, 13, 14}) {
        const float tl = timed([&](int it) { linears(layer(it), mode, sA); });
        printf("M=%d  linears %-28s alone %6.1f us per half-period\n", M, mode == 0 ? "dec_gemm (product)" : (mode == 13 ? "LDS ring 64x64, 4 waves" : "LDS ring 128x64, 8 waves"), tl);
        for (int R : {0, 6, 4, 3}) {
            auto at = attn_form(R);
            const float ta = timed([&](int it) { at(it, sA); });
            const float tc = timed([&](int it) {
                CK(hipEventRecord(eA, sA)); CK(hipStreamWaitEvent(sB, eA, 0));
                at(it, sA); linears(layer(it), mode, sB);
                CK(hipEventRecord(eB, sB)); CK(hipStreamWaitEvent(sA, eB, 0));
            });
            printf("M=%d    attention %-22s alone %6.1f us (%.2f TB/s) | beside these linears %6.1f us per half-period (serial sum %6.1f)\n", M,
                   R == 0 ? "stock grid" : (R == 6 ? "resident, 6 WGs per CU" : (R == 4 ? "resident, 4 WGs per CU" : "resident, 3 WGs per CU")), ta, abytes / 1e6 / ta, tc, ta + tl);
            fflush(stdout);
        }
    }
    for (void* b : {(void*)dW, (void*)dKV, (void*)xn, (void*)att, (void*)mid, (void*)hbuf, (void*)qb, (void*)qa, (void*)oa, (void*)nw, (void*)rope, (void*)dPos, (void*)dM, (void*)dJ}) CK(hipFree(b));
    CK(hipStreamDestroy(sA)); CK(hipStreamDestroy(sB));
}

int main() { bench(384); return 0; }
