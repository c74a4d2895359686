// experiments/f32_check.hip — the exact-mode (fp32) decode kernels of controlar_amd/csrc/decode_f32.hip: correctness against host fp64
// references, bit-equality across tile configurations and batch sizes (the batch-invariance contract), and isolated timings at the XL shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I controlar_amd/csrc experiments/f32_check.hip -o experiments/f32_check && experiments/f32_check
#include "../controlar_amd/csrc/decode_f32.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static unsigned rng_state = 12345u;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 65536.0f - 0.5f; }

__global__ void fillf_kernel(float* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = ((x & 0xffffff) / 16777216.0f - 0.5f) * scale; }
}

static int fails = 0;

static void gemm_correctness() {
    // small odd case on the host: M not a multiple of 16, every epilogue
    const int M = 37, N = 128, K = 96, H = 2, dim = 64 /* for QKV: N must be 3*dim */;
    (void)H; (void)dim;
    std::vector<float> X((size_t)M * K), W((size_t)N * K), R((size_t)M * N);
    for (auto& v : X) v = frand(); for (auto& v : W) v = frand(); for (auto& v : R) v = frand();
    float *dX, *dW, *dWp, *dO, *dR;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dWp, W.size() * 4)); CK(hipMalloc(&dO, (size_t)M * N * 4)); CK(hipMalloc(&dR, R.size() * 4));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice));
    car_launch_pack_frag_f32(dW, dWp, N, K, 0);
    std::vector<double> ref((size_t)M * N);
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)X[(size_t)m * K + k] * W[(size_t)n * K + k]; ref[(size_t)m * N + n] = s; }
    std::vector<float> first, first_sw, out((size_t)M * N);
    for (int cfg : {11, 12, 14, 21, 22, 24, 41, 42, 44}) {
        for (int epi : {FEPI_PLAIN, FEPI_RESID, FEPI_SWIGLU}) {
            if (epi == FEPI_SWIGLU && cfg < 20) continue;
            GemmFP p; memset(&p, 0, sizeof(p)); p.W = dWp; p.X = dX; p.ldx = K; p.M = M; p.N = N; p.K = K; p.out = dO; p.ldo = epi == FEPI_SWIGLU ? N / 2 : N; p.R = dR;
            CK(hipMemset(dO, 0, (size_t)M * N * 4));
            if (car_launch_dec_gemm_f32_cfg(&p, epi, cfg, 0)) { printf("cfg %d epi %d rejected\n", cfg, epi); ++fails; continue; }
            CK(hipDeviceSynchronize()); CK(hipGetLastError());
            CK(hipMemcpy(out.data(), dO, (size_t)M * N * 4, hipMemcpyDeviceToHost));
            double maxerr = 0;
            if (epi == FEPI_SWIGLU) {
                for (int m = 0; m < M; ++m) for (int hid = 0; hid < N / 2; ++hid) {
                    const int na = (hid / 16) * 32 + hid % 16, nc = na + 16;
                    const double a = ref[(size_t)m * N + na], c = ref[(size_t)m * N + nc], want = a / (1.0 + exp(-a)) * c;
                    maxerr = fmax(maxerr, fabs(out[(size_t)m * (N / 2) + hid] - want));
                }
            } else {
                for (size_t i = 0; i < ref.size(); ++i) maxerr = fmax(maxerr, fabs(out[i] - (ref[i] + (epi == FEPI_RESID ? R[i] : 0.0))));
            }
            const bool ok = maxerr < 2e-5;
            if (!ok) ++fails;
            bool same = true;
            if (epi == FEPI_PLAIN) { if (first.empty()) first = out; else same = memcmp(first.data(), out.data(), out.size() * 4) == 0; if (!same) ++fails; }
            if (epi == FEPI_SWIGLU) { if (first_sw.empty()) first_sw = out; else same = memcmp(first_sw.data(), out.data(), (size_t)M * (N / 2) * 4) == 0; if (!same) ++fails; }
            printf("gemm cfg %d epi %d: max|err| %.3g %s%s\n", cfg, epi, maxerr, ok ? "ok" : "FAIL", epi != FEPI_RESID ? (same ? " (bits = first cfg)" : " BITS DIFFER ACROSS CFG") : "");
        }
    }
    // batch invariance: rows 0..4 computed alone (M = 5) must carry the bits they have inside M = 37
    {
        GemmFP p; memset(&p, 0, sizeof(p)); p.W = dWp; p.X = dX; p.ldx = K; p.M = 5; p.N = N; p.K = K; p.out = dO; p.ldo = N;
        CK(hipMemset(dO, 0, (size_t)M * N * 4));
        car_launch_dec_gemm_f32_cfg(&p, FEPI_PLAIN, car_pick_gemm_f32_cfg(5, N, K, FEPI_PLAIN), 0); CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dO, (size_t)5 * N * 4, hipMemcpyDeviceToHost));
        const bool same = memcmp(first.data(), out.data(), (size_t)5 * N * 4) == 0;
        if (!same) ++fails;
        printf("gemm batch invariance (M=5 vs rows of M=37): %s\n", same ? "bit-identical" : "BITS DIFFER");
    }
    // QKV epilogue: dim 64? the kernel needs N = 3*dim, dim % 64 == 0 -> dim = 64 is too small for N = 128; use N = 192
    {
        const int dm = 64, N3 = 192, Hh = 1, S_max = 40, pos = 17;
        std::vector<float> W3((size_t)N3 * K), rope((size_t)S_max * 64);
        for (auto& v : W3) v = frand(); for (auto& v : rope) v = frand();
        float *dW3, *dW3p, *dq, *dk, *dv, *drope; int* dpos;
        CK(hipMalloc(&dW3, W3.size() * 4)); CK(hipMalloc(&dW3p, W3.size() * 4)); CK(hipMalloc(&dq, (size_t)M * 64 * 4)); CK(hipMalloc(&dk, (size_t)M * S_max * 64 * 4)); CK(hipMalloc(&dv, (size_t)M * S_max * 64 * 4));
        CK(hipMalloc(&drope, rope.size() * 4)); CK(hipMalloc(&dpos, 4));
        CK(hipMemcpy(dW3, W3.data(), W3.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(drope, rope.data(), rope.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dpos, &pos, 4, hipMemcpyHostToDevice));
        CK(hipMemset(dk, 0, (size_t)M * S_max * 64 * 4)); CK(hipMemset(dv, 0, (size_t)M * S_max * 64 * 4));
        car_launch_pack_frag_f32(dW3, dW3p, N3, K, 0);
        GemmFP p; memset(&p, 0, sizeof(p)); p.W = dW3p; p.X = dX; p.ldx = K; p.M = M; p.N = N3; p.K = K; p.qout = dq; p.kc = dk; p.vc = dv; p.rope = drope; p.pos = dpos; p.H = Hh; p.S_max = S_max; p.dim = dm;
        std::vector<float> q((size_t)M * 64), kk((size_t)M * S_max * 64), vv((size_t)M * S_max * 64), q0, k0;
        for (int cfg : {11, 21, 22, 24, 42, 44}) {     // the rotation must be the same bits in every tile instantiation (the contraction hazard of round 4)
            if (car_launch_dec_gemm_f32_cfg(&p, FEPI_QKV, cfg, 0)) { printf("qkv cfg %d rejected\n", cfg); ++fails; }
            CK(hipDeviceSynchronize()); CK(hipGetLastError());
            CK(hipMemcpy(q.data(), dq, q.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(kk.data(), dk, kk.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(vv.data(), dv, vv.size() * 4, hipMemcpyDeviceToHost));
            if (q0.empty()) { q0 = q; k0 = kk; }
            else { const bool same = !memcmp(q0.data(), q.data(), q.size() * 4) && !memcmp(k0.data(), kk.data(), kk.size() * 4); if (!same) ++fails; printf("gemm QKV cfg %d vs cfg 11: %s\n", cfg, same ? "bit-identical" : "BITS DIFFER"); }
        }
        double maxerr = 0;
        for (int m = 0; m < M; ++m) {
            double y[192];
            for (int n = 0; n < N3; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)X[(size_t)m * K + k] * W3[(size_t)n * K + k]; y[n] = s; }
            for (int pr = 0; pr < 32; ++pr) {
                const double cs = rope[((size_t)pos * 32 + pr) * 2], sn = rope[((size_t)pos * 32 + pr) * 2 + 1];
                const double q0 = y[2 * pr], q1 = y[2 * pr + 1], k0 = y[64 + 2 * pr], k1 = y[64 + 2 * pr + 1];
                maxerr = fmax(maxerr, fabs(q[(size_t)m * 64 + 2 * pr] - (q0 * cs - q1 * sn) * 0.125)); maxerr = fmax(maxerr, fabs(q[(size_t)m * 64 + 2 * pr + 1] - (q1 * cs + q0 * sn) * 0.125));
                maxerr = fmax(maxerr, fabs(kk[((size_t)m * S_max + pos) * 64 + 2 * pr] - (k0 * cs - k1 * sn))); maxerr = fmax(maxerr, fabs(kk[((size_t)m * S_max + pos) * 64 + 2 * pr + 1] - (k1 * cs + k0 * sn)));
            }
            for (int d = 0; d < 64; ++d) maxerr = fmax(maxerr, fabs(vv[((size_t)m * S_max + pos) * 64 + d] - y[128 + d]));
        }
        if (!(maxerr < 2e-5)) ++fails;
        printf("gemm QKV epilogue (RoPE, q scale, K/V rows at pos): max|err| %.3g %s\n", maxerr, maxerr < 2e-5 ? "ok" : "FAIL");
    }
}

static void gemm_timing() {
    struct Shape { const char* name; int N, K, epi; };
    const Shape shapes[] = {{"wqkv", 3840, 1280, FEPI_PLAIN}, {"wo", 1280, 1280, FEPI_RESID}, {"w1|w3", 7168, 1280, FEPI_SWIGLU}, {"w2", 1280, 3584, FEPI_RESID}, {"logits", 16384, 1280, FEPI_PLAIN}};
    const int NL = 6;
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int M : {16, 64, 192, 384, 768}) for (const Shape& s : shapes) {
        const size_t wsz = (size_t)s.N * s.K;
        float *W, *X, *O; CK(hipMalloc(&W, wsz * NL * 4)); CK(hipMalloc(&X, (size_t)M * s.K * 4)); CK(hipMalloc(&O, (size_t)M * s.N * 4));
        hipLaunchKernelGGL(fillf_kernel, dim3(4096), dim3(256), 0, 0, W, wsz * NL, 11u, 0.1f);
        hipLaunchKernelGGL(fillf_kernel, dim3(256), dim3(256), 0, 0, X, (size_t)M * s.K, 12u, 2.0f);
        CK(hipMemset(O, 0, (size_t)M * s.N * 4));
        CK(hipDeviceSynchronize());
        const double gflop = 2.0 * M * s.N * s.K / 1e9;
        const int pick = car_pick_gemm_f32_cfg(M, s.N, s.K, s.epi);
        printf("M=%-3d %-6s N=%-5d K=%-4d %6.2f GFLOP pick %d:", M, s.name, s.N, s.K, gflop, pick);
        for (int cfg : {44, 42, 24, 22, 41, 21}) {
            if ((M <= 16 && cfg % 10 > 1) || (M > 16 && M <= 64 && cfg % 10 < 2) || (M > 64 && cfg % 10 < 2)) continue;
            auto launch = [&](int it) {
                GemmFP p; memset(&p, 0, sizeof(p)); p.W = W + wsz * (it % NL); p.X = X; p.ldx = s.K; p.M = M; p.N = s.N; p.K = s.K; p.out = O; p.ldo = s.epi == FEPI_SWIGLU ? s.N / 2 : s.N; p.R = O;
                p.w_nt = ((M + 15) / 16 + cfg % 10 - 1) / (cfg % 10) == 1;
                if (car_launch_dec_gemm_f32_cfg(&p, s.epi, cfg, 0)) { printf(" cfg %d rejected", cfg); }
            };
            for (int i = 0; i < 2; ++i) launch(i);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, 0));
            const int reps = 24;
            for (int i = 0; i < reps; ++i) launch(i);
            CK(hipEventRecord(t1, 0)); CK(hipEventSynchronize(t1)); CK(hipGetLastError());
            float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
            const double us = ms * 1000.0 / reps;
            printf("  [%d%s] %6.1f us %5.1f TF", cfg, cfg == pick ? "*" : "", us, gflop / us * 1e3);
        }
        printf("\n"); fflush(stdout);
        CK(hipFree(W)); CK(hipFree(X)); CK(hipFree(O));
    }
}

static void attn_check_and_timing() {
    const int H = 20, S_max = 1144, T = 120, dim = H * 64;
    const int nsm = (S_max + AF_SPLIT - 1) / AF_SPLIT;
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int b : {2, 64, 192, 384}) {
        const size_t kvn = (size_t)b * H * S_max * 64;
        float *kc, *vc, *q, *part, *out; int* dpos; unsigned char* dmask;
        CK(hipMalloc(&kc, kvn * 4)); CK(hipMalloc(&vc, kvn * 4)); CK(hipMalloc(&q, (size_t)b * dim * 4)); CK(hipMalloc(&part, (size_t)b * H * nsm * 66 * 4)); CK(hipMalloc(&out, (size_t)b * dim * 4));
        CK(hipMalloc(&dpos, 4)); CK(hipMalloc(&dmask, (size_t)b * T));
        hipLaunchKernelGGL(fillf_kernel, dim3(8192), dim3(256), 0, 0, kc, kvn, 21u, 2.0f);
        hipLaunchKernelGGL(fillf_kernel, dim3(8192), dim3(256), 0, 0, vc, kvn, 22u, 2.0f);
        hipLaunchKernelGGL(fillf_kernel, dim3(64), dim3(256), 0, 0, q, (size_t)b * dim, 23u, 0.5f);
        std::vector<unsigned char> mask((size_t)b * T);
        for (int i = 0; i < b; ++i) { const int len = 8 + (i * 7) % 33; for (int j = 0; j < T; ++j) mask[(size_t)i * T + j] = j >= T - len; }     // left-padded prompts
        CK(hipMemcpy(dmask, mask.data(), mask.size(), hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        for (int pos : {120, 200, 631, 1142}) {
            CK(hipMemcpy(dpos, &pos, 4, hipMemcpyHostToDevice));
            AttnFP p; memset(&p, 0, sizeof(p)); p.q = q; p.kc = kc; p.vc = vc; p.pos = dpos; p.mask = dmask; p.part = part; p.out = out; p.H = H; p.S_max = S_max; p.T = T; p.dim = dim; p.nsplit_max = nsm;
            car_launch_dec_attn_f32(&p, b, 0); CK(hipDeviceSynchronize()); CK(hipGetLastError());
            // host check of sequence b-1, head 3
            const int bi = b - 1, h = 3;
            std::vector<float> K_((size_t)(pos + 1) * 64), V_((size_t)(pos + 1) * 64), q_(64), o_(64);
            CK(hipMemcpy(K_.data(), kc + ((size_t)bi * H + h) * S_max * 64, K_.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(V_.data(), vc + ((size_t)bi * H + h) * S_max * 64, V_.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(q_.data(), q + ((size_t)bi * H + h) * 64, 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(o_.data(), out + (size_t)bi * dim + h * 64, 256, hipMemcpyDeviceToHost));
            std::vector<double> sc(pos + 1); double mx = -1e300;
            for (int j = 0; j <= pos; ++j) { const bool ok = !(j < T && !mask[(size_t)bi * T + j]); double s = 0; for (int d = 0; d < 64; ++d) s += (double)q_[d] * K_[(size_t)j * 64 + d]; sc[j] = ok ? s : -1e300; if (ok) mx = fmax(mx, s); }
            double L = 0; std::vector<double> O(64, 0.0);
            for (int j = 0; j <= pos; ++j) if (sc[j] > -1e299) { const double w = exp(sc[j] - mx); L += w; for (int d = 0; d < 64; ++d) O[d] += w * V_[(size_t)j * 64 + d]; }
            double maxerr = 0; for (int d = 0; d < 64; ++d) maxerr = fmax(maxerr, fabs(o_[d] - O[d] / L));
            if (!(maxerr < 2e-5)) ++fails;
            // valid rows actually read
            double rows = 0; for (int i = 0; i < b; ++i) { int len = 8 + (i * 7) % 33; rows += (pos + 1 - T) + len; }
            const double bytes = rows * H * 512.0;
            for (int i = 0; i < 2; ++i) car_launch_dec_attn_f32(&p, b, 0);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, 0));
            const int reps = 10;
            for (int i = 0; i < reps; ++i) car_launch_dec_attn_f32(&p, b, 0);
            CK(hipEventRecord(t1, 0)); CK(hipEventSynchronize(t1)); CK(hipGetLastError());
            float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
            const double us = ms * 1000.0 / reps;
            printf("attn b=%-3d pos=%-4d: max|err| %.3g %s   %8.1f us  %6.2f GB -> %5.2f TB/s\n", b, pos, maxerr, maxerr < 2e-5 ? "ok" : "FAIL", us, bytes / 1e9, bytes / us / 1e6);
            fflush(stdout);
        }
        // batch invariance of the attention: sequence 0 alone vs inside the batch
        if (b == 64) {
            const int pos = 631; CK(hipMemcpy(dpos, &pos, 4, hipMemcpyHostToDevice));
            AttnFP p; memset(&p, 0, sizeof(p)); p.q = q; p.kc = kc; p.vc = vc; p.pos = dpos; p.mask = dmask; p.part = part; p.out = out; p.H = H; p.S_max = S_max; p.T = T; p.dim = dim; p.nsplit_max = nsm;
            std::vector<float> a(dim), c(dim);
            car_launch_dec_attn_f32(&p, b, 0); CK(hipDeviceSynchronize()); CK(hipMemcpy(a.data(), out, dim * 4, hipMemcpyDeviceToHost));
            car_launch_dec_attn_f32(&p, 1, 0); CK(hipDeviceSynchronize()); CK(hipMemcpy(c.data(), out, dim * 4, hipMemcpyDeviceToHost));
            const bool same = memcmp(a.data(), c.data(), dim * 4) == 0; if (!same) ++fails;
            printf("attn batch invariance (sequence 0 alone vs in a batch of 64): %s\n", same ? "bit-identical" : "BITS DIFFER");
        }
        CK(hipFree(kc)); CK(hipFree(vc)); CK(hipFree(q)); CK(hipFree(part)); CK(hipFree(out)); CK(hipFree(dpos)); CK(hipFree(dmask));
    }
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    gemm_correctness();
    if (!quick) gemm_timing();
    attn_check_and_timing();
    printf(fails ? "FAILED: %d checks\n" : "all checks passed\n", fails);
    return fails ? 1 : 0;
}
