// experiments/attn_check.hip — standalone validation + timing of controlar_amd/csrc/attn.hip (flash64_kernel) against a host
// reference, for the three modes and the shapes of the path (ViT 1025 / 197 tokens, prefill 120, T5 120).  Test infrastructure.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/attn_check.hip -o experiments/attn_check && experiments/attn_check
#include "../controlar_amd/csrc/attn.hip"

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
static unsigned long long rng_s = 0x9E3779B97F4A7C15ull;
static inline float frand() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (float)((rng_s >> 11) & 0xFFFFFF) / 8388608.0f - 1.0f; }
static inline float rb(float v) { return bf2f(f2bf(v)); }
template <typename T> static T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); return p; }
template <typename T> static void h2d(T* d, const std::vector<T>& h) { CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }

static int g_fail = 0;

// q, k, v interleaved like the prefill's qkv rows ([B][T][3*D]) when packed3, separate planes otherwise
static void run(const char* name, int B, int H, int T, int mode, float scale, bool packed3, bool timeit) {
    const int D = H * 64, Tpad = (T + 31) / 32 * 32;
    const long ld = packed3 ? 3 * D : D;
    std::vector<bf16_t> q((size_t)B * T * ld), k, v;
    std::vector<float> qf((size_t)B * T * D), kf(qf.size()), vf(qf.size());
    for (auto& x : qf) x = rb(frand() * 1.5f);
    for (auto& x : kf) x = rb(frand() * 1.5f);
    for (auto& x : vf) x = rb(frand());
    std::vector<bf16_t> vt((size_t)B * D * Tpad, 0);
    if (!packed3) { k.resize(q.size()); }
    for (int b = 0; b < B; ++b) for (int t = 0; t < T; ++t) for (int d = 0; d < D; ++d) {
        const size_t i = ((size_t)b * T + t) * D + d;
        if (packed3) { q[((size_t)b * T + t) * ld + d] = f2bf(qf[i]); q[((size_t)b * T + t) * ld + D + d] = f2bf(kf[i]); q[((size_t)b * T + t) * ld + 2 * D + d] = f2bf(vf[i]); }
        else { q[i] = f2bf(qf[i]); k[i] = f2bf(kf[i]); }
        vt[((size_t)b * D + d) * Tpad + t] = f2bf(vf[i]);
    }
    std::vector<unsigned char> mask((size_t)B * T, 1);
    if (mode) for (int b = 0; b < B; ++b) {
        const int nv = 1 + (int)((b * 37 + 11) % T);
        for (int t = 0; t < T; ++t) mask[(size_t)b * T + t] = mode == 1 ? (t >= T - nv) : (t < nv);      // prefill: left padded; T5: right padded
    }
    std::vector<float> bias;
    if (mode == 2) { bias.resize((size_t)H * T * T); for (auto& x : bias) x = rb(frand() * 2.f); }
    bf16_t* dq = dalloc<bf16_t>(q.size()); h2d(dq, q);
    bf16_t* dk = dq + D; if (!packed3) { dk = dalloc<bf16_t>(k.size()); h2d(dk, k); }
    bf16_t* dvt = dalloc<bf16_t>(vt.size()); h2d(dvt, vt);
    bf16_t* dout = dalloc<bf16_t>((size_t)B * T * D); CK(hipMemset(dout, 0xff, (size_t)B * T * D * 2));
    unsigned char* dmask = dalloc<unsigned char>(mask.size()); h2d(dmask, mask);
    float* dbias = nullptr; if (mode == 2) { dbias = dalloc<float>(bias.size()); h2d(dbias, bias); }
    FlashP p{}; p.q = dq; p.k = dk; p.vt = dvt; p.o = dout; p.q_sb = (long)T * ld; p.q_st = ld; p.k_sb = (long)T * ld; p.k_st = ld;
    p.vt_sb = (long)D * Tpad; p.vt_ld = Tpad; p.o_sb = (long)T * D; p.o_st = D; p.Tq = T; p.Tk = T; p.H = H; p.scale = scale; p.mode = mode;
    p.mask = dmask; p.bias = dbias;
    if (car_launch_flash64(&p, B, 0) != 0) { printf("%s: launcher refused\n", name); g_fail++; return; }
    CK(hipDeviceSynchronize());
    std::vector<bf16_t> out((size_t)B * T * D);
    CK(hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost));
    // host reference on a subset of (b, h)
    double maxerr = 0; long nbad = 0;
    std::vector<float> s(T), pr(T);
    for (int b = 0; b < B; b += (B > 4 ? B / 3 : 1)) for (int h = 0; h < H; h += (H > 3 ? H / 2 : 1)) for (int i = 0; i < T; ++i) {
        float mx = -INFINITY;
        for (int j = 0; j < T; ++j) {
            double a = 0; for (int d = 0; d < 64; ++d) a += (double)qf[((size_t)b * T + i) * D + h * 64 + d] * kf[((size_t)b * T + j) * D + h * 64 + d];
            float x = (float)a * scale;
            if (mode == 2) x = rb(rb(x) + bias[((size_t)h * T + i) * T + j]);
            bool ok = true;
            if (mode == 1) ok = j <= i && (mask[(size_t)b * T + j] || j == i);
            if (mode == 2) ok = mask[(size_t)b * T + j];
            s[j] = ok ? x : -INFINITY; mx = fmaxf(mx, s[j]);
        }
        double sum = 0; for (int j = 0; j < T; ++j) { pr[j] = s[j] == -INFINITY ? 0.f : expf(s[j] - mx); sum += pr[j]; }
        for (int d = 0; d < 64; ++d) {
            double a = 0; for (int j = 0; j < T; ++j) a += (double)rb((float)(pr[j] / sum)) * vf[((size_t)b * T + j) * D + h * 64 + d];
            const float got = bf2f(out[((size_t)b * T + i) * D + h * 64 + d]);
            const double e = fabs(got - a); if (e > maxerr) maxerr = e; if (!(e <= 0.02)) nbad++;
        }
    }
    const bool ok = nbad == 0 && maxerr == maxerr;
    if (!ok) g_fail++;
    printf("%-44s B %3d H %2d T %4d  max|d| %.3e  %s", name, B, H, T, maxerr, ok ? "OK" : "FAIL");
    if (timeit) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) car_launch_flash64(&p, B, 0);
        CK(hipEventRecord(e0, 0));
        const int n = 20; for (int i = 0; i < n; ++i) car_launch_flash64(&p, B, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= n;
        const double fl = 4.0 * B * H * (double)T * T * 64 * (mode == 1 ? 0.5 : 1.0);
        printf("   %.3f ms  %.1f TFLOP/s", ms, fl / ms / 1e9);
    }
    printf("\n");
    CK(hipFree(dq)); if (!packed3) CK(hipFree(dk)); CK(hipFree(dvt)); CK(hipFree(dout)); CK(hipFree(dmask)); if (dbias) CK(hipFree(dbias));
}

int main() {
    run("vit dinov2-small 1025 (QT=2)", 8, 6, 1025, 0, 0.125f, false, false);
    run("vit-s/16 197 (QT=2)", 4, 6, 197, 0, 0.125f, false, false);
    run("short 33", 3, 2, 33, 0, 0.125f, false, false);
    run("prefill 120 causal+pad, qkv rows", 9, 4, 120, 1, 0.125f, true, false);
    run("prefill 1 row", 2, 2, 1, 1, 0.125f, true, false);
    run("t5 120 bias+mask", 7, 4, 120, 2, 1.0f, false, false);
    run("t5 64", 3, 2, 64, 2, 1.0f, false, false);
    run("time: vit 1025 x 256 img x 6 heads", 256, 6, 1025, 0, 0.125f, false, true);
    run("time: vit-base 1025 x 64 img x 12 heads", 64, 12, 1025, 0, 0.125f, false, true);
    run("time: prefill 120 x 512 rows x 20 heads", 512, 20, 120, 1, 0.125f, true, true);
    run("time: t5 120 x 256 x 32 heads", 256, 32, 120, 2, 1.0f, false, true);
    printf(g_fail ? "FAILED (%d)\n" : "ALL OK\n", g_fail);
    return g_fail ? 1 : 0;
}
