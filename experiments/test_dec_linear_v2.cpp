// Standalone check of experiments/dec_linear_v2.hip against a host reference (run on an MI355X):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/dec_linear_v2.hip experiments/test_dec_linear_v2.cpp -o /tmp/t_v2 && /tmp/t_v2
// Prints max |error| per epilogue and a crude GB/s; exits non-zero on mismatch.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint16_t bf16_t;
struct Lin2P { const bf16_t* W; const bf16_t* X; void* out; int b, N, K, KS; };
extern "C" void exp_launch_dec_linear_v2(const Lin2P* p, int epi, hipStream_t st);

static float bf2f(bf16_t v) { union { uint32_t u; float f; } c; c.u = ((uint32_t)v) << 16; return c.f; }
static bf16_t f2bf(float f) { union { uint32_t u; float f; } c; c.f = f; uint32_t u = c.u; u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static float rnd(float f) { return bf2f(f2bf(f)); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

int main() {
    const int cases[][4] = {{8, 3840, 1280, 1}, {64, 3840, 1280, 1}, {33, 7168, 1280, 1}, {64, 1280, 3584, 7}, {64, 16384, 1280, 1}};
    int bad = 0;
    for (auto& cs : cases) {
        const int b = cs[0], N = cs[1], K = cs[2], KS = cs[3];
        std::vector<bf16_t> W((size_t)N * K), X((size_t)b * K), Wp((size_t)N * K);
        srand(1);
        for (auto& v : W) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.06f);
        for (auto& v : X) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
        const int nkb = K / 32;
        for (int rb = 0; rb < N / 16; ++rb) for (int kb = 0; kb < nkb; ++kb) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e)
            Wp[(((size_t)rb * nkb + kb) * 64 + l) * 8 + e] = W[(size_t)(rb * 16 + (l & 15)) * K + kb * 32 + (l >> 4) * 8 + e];
        std::vector<float> ref((size_t)b * N);
        for (int m = 0; m < b; ++m) for (int n = 0; n < N; ++n) { double a = 0; for (int k = 0; k < K; ++k) a += (double)bf2f(X[(size_t)m * K + k]) * bf2f(W[(size_t)n * K + k]); ref[(size_t)m * N + n] = (float)a; }
        bf16_t *dW, *dX; void* dO;
        CK(hipMalloc(&dW, Wp.size() * 2)); CK(hipMalloc(&dX, X.size() * 2)); CK(hipMalloc(&dO, (size_t)KS * b * N * 4));
        CK(hipMemcpy(dW, Wp.data(), Wp.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X.data(), X.size() * 2, hipMemcpyHostToDevice));
        Lin2P p{dW, dX, dO, b, N, K, KS};
        // --- EPI_F32 (partials summed on the host)
        exp_launch_dec_linear_v2(&p, 0, 0); CK(hipDeviceSynchronize());
        std::vector<float> o32((size_t)KS * b * N); CK(hipMemcpy(o32.data(), dO, o32.size() * 4, hipMemcpyDeviceToHost));
        double e0 = 0; for (int m = 0; m < b; ++m) for (int n = 0; n < N; ++n) { float a = 0; for (int s = 0; s < KS; ++s) a += o32[((size_t)s * b + m) * N + n]; e0 = fmax(e0, fabs(a - ref[(size_t)m * N + n])); }
        double e1 = 0, e2 = 0;
        if (KS == 1) {
            exp_launch_dec_linear_v2(&p, 1, 0); CK(hipDeviceSynchronize());
            std::vector<bf16_t> o16((size_t)b * N); CK(hipMemcpy(o16.data(), dO, o16.size() * 2, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < o16.size(); ++i) e1 = fmax(e1, fabs(bf2f(o16[i]) - rnd(ref[i])) / (fabs(ref[i]) + 1.0));
            if (N % 32 == 0) {
                exp_launch_dec_linear_v2(&p, 2, 0); CK(hipDeviceSynchronize());
                std::vector<bf16_t> og((size_t)b * N / 2); CK(hipMemcpy(og.data(), dO, og.size() * 2, hipMemcpyDeviceToHost));
                for (int m = 0; m < b; ++m) for (int h = 0; h < N / 2; ++h) {
                    const int col = (h >> 4) * 32 + (h & 15);
                    const float a = rnd(ref[(size_t)m * N + col]), c = rnd(ref[(size_t)m * N + col + 16]);
                    const float want = rnd(rnd(a / (1.0f + expf(-a))) * c);
                    e2 = fmax(e2, fabs(bf2f(og[(size_t)m * (N / 2) + h]) - want) / (fabs(want) + 1.0));
                }
            }
        }
        hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
        CK(hipEventRecord(t0, 0)); for (int i = 0; i < 20; ++i) exp_launch_dec_linear_v2(&p, KS == 1 ? 1 : 0, 0); CK(hipEventRecord(t1, 0)); CK(hipEventSynchronize(t1));
        float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
        printf("b=%d N=%d K=%d KS=%d: f32 max|d|=%.3g  bf16 rel=%.3g  swiglu rel=%.3g  %.1f us/launch  %.2f TB/s (weights)\n", b, N, K, KS, e0, e1, e2, ms / 20 * 1e3,
               (double)N * K * 2 / (ms / 20 * 1e-3) / 1e12);
        if (e0 > 2e-3 || e1 > 8e-3 || e2 > 2e-2) ++bad;
        CK(hipFree(dW)); CK(hipFree(dX)); CK(hipFree(dO));
    }
    printf(bad ? "FAILED\n" : "OK\n");
    return bad;
}
