// experiments/conv_check.hip — standalone check + timing of the VQ decoder's 3x3 convolution kernels (controlar_amd/csrc/gemm.hip):
// conv3_halo64_kernel (round 3: 64-channel halo groups, 75 KB of LDS, two workgroups per CU) against conv3_halo_kernel (131 KB, one per CU) on the
// same random NHWC inputs — same taps, same weights, another fp32 summation order (so: equal up to one bf16 ulp on a small share of outputs) —
// then isolated times at the decoder's shapes.  Test infrastructure.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I controlar_amd/csrc experiments/conv_check.hip -o experiments/conv_check && experiments/conv_check
#include "../controlar_amd/csrc/gemm.hip"
#include "../controlar_amd/csrc/ops.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
template <typename T> static T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); return p; }

__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed, unsigned expo) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        // two bf16 values with exponent `expo` .. expo+3: magnitudes 2^(expo-127) .. 2^(expo-124), random sign and mantissa
        const unsigned a = (x & 0x807fu) | ((((x >> 7) & 3) + expo) << 7), b2 = ((x >> 16) & 0x807fu) | (((((x >> 23) & 3) + expo)) << 7);
        p[i] = a | (b2 << 16); }
}

static int g_fail = 0;
struct Shape { const char* name; int B, Ho, Wo, Cin, Cout, ups; };

static void run(const Shape& s, bool timing) {
    const int Hin = s.Ho >> s.ups, Win = s.Wo >> s.ups;
    const size_t nin = (size_t)s.B * Hin * Win * s.Cin, nout = (size_t)s.B * s.Ho * s.Wo * s.Cout, nw = (size_t)s.Cout * 9 * s.Cin;
    bf16_t *x = dalloc<bf16_t>(nin), *w = dalloc<bf16_t>(nw), *bias = dalloc<bf16_t>(s.Cout), *r = dalloc<bf16_t>(nout), *y0 = dalloc<bf16_t>(nout), *y1 = dalloc<bf16_t>(nout);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (unsigned*)x, nin / 2, 1u, 124u);        // |x| in [0.125, 1)
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (unsigned*)w, nw / 2, 2u, 119u);         // |w| in [2^-8, 2^-5)
    hipLaunchKernelGGL(fill_kernel, dim3(8), dim3(64), 0, 0, (unsigned*)bias, (size_t)s.Cout / 2, 3u, 122u);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (unsigned*)r, nout / 2, 4u, 124u);
    CK(hipDeviceSynchronize());
    auto launch = [&](bf16_t* y, bool old) {
        GemmP p; memset(&p, 0, sizeof(p));
        p.A = x; p.W = w; p.C = y; p.lda = 0; p.ldw = 9L * s.Cin; p.ldc = s.Cout; p.M = s.B * s.Ho * s.Wo; p.N = s.Cout; p.K = 9 * s.Cin; p.alpha = 1.f; p.nb0 = p.nb1 = 1;
        p.bias = bias; p.bias_mode = BIAS_N; p.Ho = s.Ho; p.Wo = s.Wo; p.Cin = s.Cin; p.ups = s.ups; p.R = r; p.ldr = s.Cout; p.patch = 1;
        if (old) setenv("CAR_CONV_HALO128", "1", 1); else unsetenv("CAR_CONV_HALO128");
        car_launch_gemm(1, AMODE_CONV3, &p, 0);
    };
    if (!timing) {
        CK(hipMemset(y0, 0xff, nout * 2)); CK(hipMemset(y1, 0xff, nout * 2));
        launch(y0, true); launch(y1, false); CK(hipDeviceSynchronize()); CK(hipGetLastError());
        std::vector<bf16_t> h0(nout), h1(nout);
        CK(hipMemcpy(h0.data(), y0, nout * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), y1, nout * 2, hipMemcpyDeviceToHost));
        double e = 0, scale = 0; size_t nd = 0; bool nan = false;
        for (size_t i = 0; i < nout; ++i) { const double a = bf2f(h0[i]), b2 = bf2f(h1[i]); if (a != a || b2 != b2) nan = true; const double d = std::fabs(a - b2); if (d > 0) ++nd; if (d > e) e = d; if (std::fabs(a) > scale) scale = std::fabs(a); }
        const bool ok = !nan && e <= scale / 64.0 && (double)nd / nout < 0.05;      // <= 2 bf16 ulps of the output scale, on a small share of outputs
        printf("%-34s B=%d %dx%d %d->%d ups=%d: max|d| %.3e (scale %.2f)  differing %.3f%%  %s\n", s.name, s.B, s.Ho, s.Wo, s.Cin, s.Cout, s.ups, e, scale, 100.0 * nd / nout, ok ? "OK" : "FAIL");
        if (!ok) ++g_fail;
    } else {
        hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
        float ms[2];
        for (int v = 0; v < 2; ++v) {
            for (int i = 0; i < 2; ++i) launch(y0, v == 0);
            CK(hipDeviceSynchronize()); CK(hipEventRecord(t0, 0));
            for (int i = 0; i < 6; ++i) launch(y0, v == 0);
            CK(hipEventRecord(t1, 0)); CK(hipEventSynchronize(t1)); CK(hipGetLastError()); CK(hipEventElapsedTime(&ms[v], t0, t1)); ms[v] /= 6;
        }
        const double fl = 2.0 * s.B * s.Ho * s.Wo * (double)s.Cout * 9 * s.Cin;
        printf("%-34s B=%d %dx%d %d->%d ups=%d: halo128 %8.1f us (%4.0f TFLOP/s)   halo64 %8.1f us (%4.0f TFLOP/s)   x%.2f\n", s.name, s.B, s.Ho, s.Wo, s.Cin, s.Cout, s.ups,
               ms[0] * 1e3, fl / ms[0] / 1e9, ms[1] * 1e3, fl / ms[1] / 1e9, ms[0] / ms[1]);
    }
    fflush(stdout);
    for (void* q : {(void*)x, (void*)w, (void*)bias, (void*)r, (void*)y0, (void*)y1}) CK(hipFree(q));
}

// GroupNorm stage-1 partials written by the conv epilogue (GemmP::gn_part) against gn_partial_vec_kernel's pass over the stored output: after the
// fixed-order finalize the (mean, rstd) pairs must agree to fp32 round-off (other summation order inside a tile).  Also the cost of writing them.
static void run_part(const Shape& s, bool timing) {
    const int Hin = s.Ho >> s.ups, Win = s.Wo >> s.ups, HW = s.Ho * s.Wo, nchunk = HW / 256;
    const size_t nin = (size_t)s.B * Hin * Win * s.Cin, nout = (size_t)s.B * HW * s.Cout, nw = (size_t)s.Cout * 9 * s.Cin;
    bf16_t *x = dalloc<bf16_t>(nin), *w = dalloc<bf16_t>(nw), *bias = dalloc<bf16_t>(s.Cout), *r = dalloc<bf16_t>(nout), *y = dalloc<bf16_t>(nout);
    const size_t npart = (size_t)s.B * nchunk * 2 * s.Cout;
    float *p0 = dalloc<float>(npart + 256), *p1 = dalloc<float>(npart + 256), *st0 = dalloc<float>((size_t)s.B * 64 + 64), *st1 = dalloc<float>((size_t)s.B * 64 + 64);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (unsigned*)x, nin / 2, 21u, 124u);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (unsigned*)w, nw / 2, 22u, 119u);
    hipLaunchKernelGGL(fill_kernel, dim3(8), dim3(64), 0, 0, (unsigned*)bias, (size_t)s.Cout / 2, 23u, 122u);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (unsigned*)r, nout / 2, 24u, 124u);
    CK(hipMemset(p1, 0xff, npart * 4)); CK(hipDeviceSynchronize());
    unsetenv("CAR_CONV_HALO128");
    auto conv = [&](float* part) {
        GemmP p; memset(&p, 0, sizeof(p));
        p.A = x; p.W = w; p.C = y; p.ldw = 9L * s.Cin; p.ldc = s.Cout; p.M = s.B * HW; p.N = s.Cout; p.K = 9 * s.Cin; p.alpha = 1.f; p.nb0 = p.nb1 = 1;
        p.bias = bias; p.bias_mode = BIAS_N; p.Ho = s.Ho; p.Wo = s.Wo; p.Cin = s.Cin; p.ups = s.ups; p.R = r; p.ldr = s.Cout; p.patch = 1;
        if (part) { if (!car_conv3_halo64_ok(1, &p)) { printf("not eligible\n"); exit(3); } p.gn_part = part; }
        car_launch_gemm(1, AMODE_CONV3, &p, 0);
    };
    if (!timing) {
        conv(p1);                                                                                                  // y + partials from the epilogue
        car_launch_groupnorm_ex(1, y, bias, bias, nullptr, p1, st1, s.B, HW, s.Cout, 32, 1e-6f, 1, 1, 0);        // finalize only (y == nullptr: no apply)
        car_launch_groupnorm_ex(1, y, bias, bias, nullptr, p0, st0, s.B, HW, s.Cout, 32, 1e-6f, 1, 0, 0);        // stage 1 by the separate pass + finalize
        CK(hipDeviceSynchronize()); CK(hipGetLastError());
        std::vector<float> a((size_t)s.B * 64), b2((size_t)s.B * 64);
        CK(hipMemcpy(a.data(), st0, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b2.data(), st1, b2.size() * 4, hipMemcpyDeviceToHost));
        double e = 0; bool nan = false;
        for (size_t i = 0; i < a.size(); ++i) { if (a[i] != a[i] || b2[i] != b2[i]) nan = true; const double d = std::fabs((double)a[i] - b2[i]) / (std::fabs((double)a[i]) + 1e-3); if (d > e) e = d; }
        const bool ok = !nan && e < 2e-5;
        printf("%-34s B=%d %dx%d %d->%d ups=%d: (mean, rstd) from the epilogue partials vs the separate pass: max rel diff %.2e  %s\n", s.name, s.B, s.Ho, s.Wo, s.Cin, s.Cout, s.ups, e, ok ? "OK" : "FAIL");
        if (!ok) ++g_fail;
    } else {
        hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
        float ms[3];
        for (int v = 0; v < 3; ++v) {
            auto go = [&]() { if (v == 0) conv(nullptr); else if (v == 1) conv(p1); else { conv(nullptr); car_launch_groupnorm_ex(1, y, bias, bias, nullptr, p0, st0, s.B, HW, s.Cout, 32, 1e-6f, 1, 0, 0); } };
            for (int i = 0; i < 2; ++i) go();
            CK(hipDeviceSynchronize()); CK(hipEventRecord(t0, 0));
            for (int i = 0; i < 6; ++i) go();
            CK(hipEventRecord(t1, 0)); CK(hipEventSynchronize(t1)); CK(hipGetLastError()); CK(hipEventElapsedTime(&ms[v], t0, t1)); ms[v] /= 6;
        }
        printf("%-34s B=%d %dx%d %d->%d ups=%d: conv %8.1f us | conv + partials in the epilogue %8.1f us | conv + separate stage-1 pass + finalize %8.1f us\n", s.name, s.B, s.Ho, s.Wo, s.Cin, s.Cout, s.ups, ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3);
    }
    fflush(stdout);
    for (void* q : {(void*)x, (void*)w, (void*)bias, (void*)r, (void*)y, (void*)p0, (void*)p1, (void*)st0, (void*)st1}) CK(hipFree(q));
}

int main() {
    const Shape checks[] = {{"check 128->128", 2, 64, 64, 128, 128, 0}, {"check 256->128", 1, 32, 48, 256, 128, 0}, {"check ups 256->256", 2, 64, 32, 256, 256, 1},
                            {"check 64->128 (one group)", 1, 32, 32, 64, 128, 0}, {"check 512->512", 1, 32, 32, 512, 512, 0}};
    for (const Shape& s : checks) run(s, false);
    const Shape pchecks[] = {{"partials 128->128", 2, 64, 64, 128, 128, 0}, {"partials 256->256 ups", 3, 64, 32, 256, 256, 1}, {"partials 512->512", 1, 32, 32, 512, 512, 0}, {"partials 256->128", 2, 48, 32, 256, 128, 0}};
    for (const Shape& s : pchecks) run_part(s, false);
    printf("== correctness: %d failure(s)\n", g_fail);
    // the VQ-16 decoder's 3x3 convolutions at 512x512 output, 24 images per chunk (engine.hip car_vq_decode): profiles/r02_bench_b768_trace_summary.txt
    const Shape times[] = {{"level 512^2 128->128", 24, 512, 512, 128, 128, 0}, {"upsample -> 512^2 128->128", 24, 512, 512, 128, 128, 1},
                           {"level 256^2 256->128", 24, 256, 256, 256, 128, 0}, {"level 256^2 128->128", 24, 256, 256, 128, 128, 0}, {"upsample -> 256^2 256->256", 24, 256, 256, 256, 256, 1},
                           {"level 128^2 256->256", 24, 128, 128, 256, 256, 0}, {"level 64^2 512->512", 24, 64, 64, 512, 512, 0}, {"level 32^2 512->512", 24, 32, 32, 512, 512, 0}};
    for (const Shape& s : times) run(s, true);
    for (const Shape& s : times) run_part(s, true);
    return g_fail ? 1 : 0;
}
