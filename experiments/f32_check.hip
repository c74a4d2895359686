// experiments/f32_check.hip — the exact-mode (fp32) decode kernels of controlar_amd/csrc/decode_f32.hip: correctness against host fp64
// references, bit-equality across tile configurations and batch sizes (the batch-invariance contract), and isolated timings at the XL shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I controlar_amd/csrc experiments/f32_check.hip -o experiments/f32_check && experiments/f32_check
#define F32T_ALL_CONFIGS
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
static bool argc_only_attn = false;

static void gemm_correctness() {
    // small odd case on the host: M not a multiple of 16, every epilogue
    const int M = 37, N = 128, K = 96, H = 2, dim = 64 /* for QKV: N must be 3*dim */;
    (void)H; (void)dim;
    std::vector<float> X((size_t)M * K), W((size_t)N * K), R((size_t)M * N);
    for (auto& v : X) v = frand(); for (auto& v : W) v = frand(); for (auto& v : R) v = frand();
    float *dX, *dW, *dWp, *dO, *dR;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dWp, W.size() * 4)); CK(hipMalloc(&dO, (size_t)M * N * 4)); CK(hipMalloc(&dR, R.size() * 4));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice));
    car_launch_pack_frag_f32(dW, dWp, N, K, nullptr, 0);
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
        car_launch_pack_frag_f32(dW3, dW3p, N3, K, nullptr, 0);
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


// round 5: the LDS-tiled kernel dec_gemm_f32t against the register kernel — same canonical arithmetic, so every output must carry the same BITS
// (all epilogues, M not a multiple of the tile, both stage depths: len = K/128 even -> 2 k-blocks per stage, odd -> 1)
static void tiled_correctness() {
    for (int K : {256, 384, 1280}) for (int M : {37, 150}) {
        const int dm = 128, N = 3 * dm, Hh = 2, S_max = 24, pos = 11;      // N = 384 = 3*dim for the QKV epilogue; also used as a plain N
        std::vector<float> X((size_t)M * K), W((size_t)N * K), R((size_t)M * N), rope((size_t)S_max * 64);
        for (auto& v : X) v = frand(); for (auto& v : W) v = frand(); for (auto& v : R) v = frand(); for (auto& v : rope) v = frand();
        float *dX, *dW, *dWp, *dO, *dR, *dq, *dk, *dv, *drope; int* dpos;
        CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dWp, W.size() * 4)); CK(hipMalloc(&dO, (size_t)M * N * 4)); CK(hipMalloc(&dR, R.size() * 4));
        CK(hipMalloc(&dq, (size_t)M * dm * 4)); CK(hipMalloc(&dk, (size_t)M * Hh * S_max * 64 * 4)); CK(hipMalloc(&dv, (size_t)M * Hh * S_max * 64 * 4)); CK(hipMalloc(&drope, rope.size() * 4)); CK(hipMalloc(&dpos, 4));
        CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(drope, rope.data(), rope.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dpos, &pos, 4, hipMemcpyHostToDevice));
        car_launch_pack_frag_f32(dW, dWp, N, K, nullptr, 0);
        const size_t kvn = (size_t)M * Hh * S_max * 64;
        auto run = [&](int cfg, int epi, std::vector<float>& out) -> bool {
            GemmFP p; memset(&p, 0, sizeof(p)); p.W = dWp; p.X = dX; p.ldx = K; p.M = M; p.N = N; p.K = K; p.out = dO; p.ldo = epi == FEPI_SWIGLU ? N / 2 : N; p.R = dR;
            p.qout = dq; p.kc = dk; p.vc = dv; p.rope = drope; p.pos = dpos; p.H = Hh; p.S_max = S_max; p.dim = dm;
            CK(hipMemset(dO, 0, (size_t)M * N * 4)); CK(hipMemset(dq, 0, (size_t)M * dm * 4)); CK(hipMemset(dk, 0, kvn * 4)); CK(hipMemset(dv, 0, kvn * 4));
            if (car_launch_dec_gemm_f32_cfg(&p, epi, cfg, 0)) return false;
            CK(hipDeviceSynchronize()); CK(hipGetLastError());
            if (epi == FEPI_QKV) { out.resize((size_t)M * dm + 2 * kvn); CK(hipMemcpy(out.data(), dq, (size_t)M * dm * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(out.data() + (size_t)M * dm, dk, kvn * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(out.data() + (size_t)M * dm + kvn, dv, kvn * 4, hipMemcpyDeviceToHost)); }
            else { out.resize((size_t)M * N); CK(hipMemcpy(out.data(), dO, (size_t)M * N * 4, hipMemcpyDeviceToHost)); }
            return true;
        };
        for (int epi : {FEPI_PLAIN, FEPI_RESID, FEPI_SWIGLU, FEPI_QKV}) {
            std::vector<float> ref, out;
            if (!run(22, epi, ref)) { printf("tiled check: reference cfg 22 rejected\n"); ++fails; continue; }
            if (epi == FEPI_PLAIN) {      // the reference itself against fp64
                double maxerr = 0;
                for (int m = 0; m < M; m += 7) for (int n = 0; n < N; n += 5) { double s = 0; for (int k = 0; k < K; ++k) s += (double)X[(size_t)m * K + k] * W[(size_t)n * K + k]; maxerr = fmax(maxerr, fabs(ref[(size_t)m * N + n] - s)); }
                if (!(maxerr < 1e-4)) { ++fails; printf("tiled check K=%d: cfg 22 vs fp64 %.3g FAIL\n", K, maxerr); }
            }
            for (int cfg : {1221, 1241, 1421, 1222, 1124, 1214, 1122, 1212, 1114, 1118}) {
                if (!run(cfg, epi, out)) { if (!((cfg % 1000 == 241 || cfg % 1000 == 421 || cfg == 2221) && (K / 128) % 2)) { printf("tiled cfg %d epi %d K=%d rejected\n", cfg, epi, K); ++fails; } continue; }
                const bool same = out.size() == ref.size() && memcmp(out.data(), ref.data(), out.size() * 4) == 0;
                if (!same) { ++fails; size_t bad = 0, first = 0; for (size_t i = 0; i < out.size(); ++i) if (memcmp(&out[i], &ref[i], 4)) { if (!bad) first = i; ++bad; }
                    printf("tiled cfg %d epi %d K=%d M=%d: BITS DIFFER (%zu of %zu, first at %zu: %.9g vs %.9g)\n", cfg, epi, K, M, bad, out.size(), first, out[first], ref[first]); }
            }
        }
        // NX: on-the-fly RMSNorm.  W' = W·diag(nw) packed with the column scale, X = raw rows; against fp64 of sum_k (x_k rstd nw_k) W_nk, and bit-equal across kernels
        {
            std::vector<float> nw(K); for (auto& v : nw) v = 1.0f + frand();
            float *dnw, *dWn; CK(hipMalloc(&dnw, K * 4)); CK(hipMalloc(&dWn, W.size() * 4)); CK(hipMemcpy(dnw, nw.data(), K * 4, hipMemcpyHostToDevice));
            car_launch_pack_frag_f32(dW, dWn, N, K, dnw, 0);
            const float eps = 1e-5f;
            auto runx = [&](int cfg, int epi, std::vector<float>& out) -> bool {
                GemmFP p; memset(&p, 0, sizeof(p)); p.W = dWn; p.X = dX; p.ldx = K; p.M = M; p.N = N; p.K = K; p.out = dO; p.ldo = epi == FEPI_SWIGLU ? N / 2 : N; p.normx = 1; p.neps = eps;
                p.qout = dq; p.kc = dk; p.vc = dv; p.rope = drope; p.pos = dpos; p.H = Hh; p.S_max = S_max; p.dim = dm;
                CK(hipMemset(dO, 0, (size_t)M * N * 4)); CK(hipMemset(dq, 0, (size_t)M * dm * 4)); CK(hipMemset(dk, 0, kvn * 4)); CK(hipMemset(dv, 0, kvn * 4));
                if (car_launch_dec_gemm_f32_cfg(&p, epi, cfg, 0)) return false;
                CK(hipDeviceSynchronize()); CK(hipGetLastError());
                if (epi == FEPI_QKV) { out.resize((size_t)M * dm + 2 * kvn); CK(hipMemcpy(out.data(), dq, (size_t)M * dm * 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(out.data() + (size_t)M * dm, dk, kvn * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(out.data() + (size_t)M * dm + kvn, dv, kvn * 4, hipMemcpyDeviceToHost)); }
                else { out.resize((size_t)M * N); CK(hipMemcpy(out.data(), dO, (size_t)M * N * 4, hipMemcpyDeviceToHost)); }
                return true;
            };
            for (int epi : {FEPI_PLAIN, FEPI_SWIGLU, FEPI_QKV}) {
                std::vector<float> ref, out;
                if (!runx(22, epi, ref)) { printf("NX reference cfg 22 rejected\n"); ++fails; continue; }
                if (epi == FEPI_PLAIN) {
                    double maxerr = 0;
                    for (int m = 0; m < M; m += 5) { double ss = 0; for (int k = 0; k < K; ++k) ss += (double)X[(size_t)m * K + k] * X[(size_t)m * K + k];
                        const double rstd = 1.0 / sqrt(ss / K + eps);
                        for (int n = 0; n < N; n += 7) { double sum = 0; for (int k = 0; k < K; ++k) sum += (double)X[(size_t)m * K + k] * rstd * nw[k] * W[(size_t)n * K + k]; maxerr = fmax(maxerr, fabs(ref[(size_t)m * N + n] - sum)); } }
                    if (!(maxerr < 2e-4)) ++fails;
                    printf("NX gemm (on-the-fly RMSNorm) K=%d M=%d vs fp64: max|err| %.3g %s\n", K, M, maxerr, maxerr < 2e-4 ? "ok" : "FAIL");
                }
                for (int cfg : {21, 42, 1221, 1241, 1421, 1222, 1124, 1214, 1122, 1212, 1114, 1118}) {
                    if (!runx(cfg, epi, out)) { if (!((cfg == 1241 || cfg == 1421 || cfg == 2221) && (K / 128) % 2)) { printf("NX cfg %d epi %d K=%d rejected\n", cfg, epi, K); ++fails; } continue; }
                    const bool same = out.size() == ref.size() && memcmp(out.data(), ref.data(), out.size() * 4) == 0;
                    if (!same) { ++fails; printf("NX cfg %d epi %d K=%d M=%d: BITS DIFFER\n", cfg, epi, K, M); }
                }
            }
            CK(hipFree(dnw)); CK(hipFree(dWn));
        }
        printf("tiled kernels vs register kernel, K=%d M=%d: all epilogues compared (plain and NX)\n", K, M);
        CK(hipFree(dX)); CK(hipFree(dW)); CK(hipFree(dWp)); CK(hipFree(dO)); CK(hipFree(dR)); CK(hipFree(dq)); CK(hipFree(dk)); CK(hipFree(dv)); CK(hipFree(drope)); CK(hipFree(dpos));
    }
}

static void gemm_timing() {
    struct Shape { const char* name; int N, K, epi; };
    const Shape shapes[] = {{"wqkv", 3840, 1280, FEPI_PLAIN}, {"wo", 1280, 1280, FEPI_RESID}, {"w1|w3", 7168, 1280, FEPI_SWIGLU}, {"w2", 1280, 3584, FEPI_RESID}, {"logits", 16384, 1280, FEPI_PLAIN}};
    const int NL = 6;
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int M : {64, 128, 192, 384, 768}) for (const Shape& s : shapes) {
        if (argc_only_attn) break;
        const size_t wsz = (size_t)s.N * s.K;
        float *W, *X, *O; CK(hipMalloc(&W, wsz * NL * 4)); CK(hipMalloc(&X, (size_t)M * s.K * 4)); CK(hipMalloc(&O, (size_t)M * s.N * 4));
        hipLaunchKernelGGL(fillf_kernel, dim3(4096), dim3(256), 0, 0, W, wsz * NL, 11u, 0.1f);
        hipLaunchKernelGGL(fillf_kernel, dim3(256), dim3(256), 0, 0, X, (size_t)M * s.K, 12u, 2.0f);
        CK(hipMemset(O, 0, (size_t)M * s.N * 4));
        CK(hipDeviceSynchronize());
        const double gflop = 2.0 * M * s.N * s.K / 1e9;
        const int pick = car_pick_gemm_f32_cfg(M, s.N, s.K, s.epi);
        printf("M=%-3d %-6s N=%-5d K=%-4d %6.2f GFLOP pick %d:", M, s.name, s.N, s.K, gflop, pick);
        for (int cfg : {22, 21, 1221, 1241, 1421, 1222, 1124, 1214, 1122, 1212, 1114, 1118}) {
            if (cfg < 1000 && ((M <= 16 && cfg % 10 > 1) || (M > 16 && cfg % 10 < 2))) continue;
            if ((cfg == 122 || cfg == 222) && M < 64) continue;
            if (cfg >= 1000 && M < 64) continue;
            const int nx = 0;
            auto launch = [&](int it) {
                GemmFP p; memset(&p, 0, sizeof(p)); p.W = W + wsz * (it % NL); p.X = X; p.ldx = s.K; p.M = M; p.N = s.N; p.K = s.K; p.out = O; p.ldo = s.epi == FEPI_SWIGLU ? s.N / 2 : s.N; p.R = O;
                p.normx = nx; p.neps = 1e-5f;
                p.w_nt = cfg < 1000 && ((M + 15) / 16 + cfg % 10 - 1) / (cfg % 10) == 1;
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
        if (s.epi != FEPI_RESID && M >= 64) {      // the same shapes with the on-the-fly RMSNorm (the form the engine runs for wqkv / w1|w3 / logits)
            printf("\n      with NX:                                ");
            for (int cfg : {22, 1221, 1212, 1214}) {
                auto launch = [&](int it) {
                    GemmFP p; memset(&p, 0, sizeof(p)); p.W = W + wsz * (it % NL); p.X = X; p.ldx = s.K; p.M = M; p.N = s.N; p.K = s.K; p.out = O; p.ldo = s.epi == FEPI_SWIGLU ? s.N / 2 : s.N;
                    p.normx = 1; p.neps = 1e-5f;
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
            car_launch_dec_attn_f32_ex(&p, b, 0, 0); CK(hipDeviceSynchronize()); CK(hipGetLastError());
            {   // the one-launch form must give the bits of split + combine
                std::vector<float> o0((size_t)b * dim), o1((size_t)b * dim);
                CK(hipMemcpy(o0.data(), out, o0.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemset(out, 0, o0.size() * 4));
                for (int form : {1, 3, 4}) {
                    CK(hipMemset(out, 0, o0.size() * 4));
                    car_launch_dec_attn_f32_ex(&p, b, form, 0); CK(hipDeviceSynchronize()); CK(hipGetLastError());
                    CK(hipMemcpy(o1.data(), out, o1.size() * 4, hipMemcpyDeviceToHost));
                    if (memcmp(o0.data(), o1.data(), o0.size() * 4)) { ++fails; printf("attn b=%d pos=%d: one-launch form %d BITS DIFFER from split + combine\n", b, pos, form); }
                }
            }
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
            double usv[4];
            for (int fused : {0, 1, 3}) {
                for (int i = 0; i < 2; ++i) car_launch_dec_attn_f32_ex(&p, b, fused, 0);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(t0, 0));
                const int reps = 10;
                for (int i = 0; i < reps; ++i) car_launch_dec_attn_f32_ex(&p, b, fused, 0);
                CK(hipEventRecord(t1, 0)); CK(hipEventSynchronize(t1)); CK(hipGetLastError());
                float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
                usv[fused] = ms * 1000.0 / reps;
            }
            const double us = usv[0];
            printf("attn b=%-3d pos=%-4d: max|err| %.3g %s   split+combine %8.1f us  %6.2f GB -> %5.2f TB/s | one launch %8.1f us -> %5.2f TB/s | 12-wave WGs %8.1f us -> %5.2f TB/s\n", b, pos, maxerr, maxerr < 2e-5 ? "ok" : "FAIL", us, bytes / 1e9, bytes / us / 1e6, usv[1], bytes / usv[1] / 1e6, usv[3], bytes / usv[3] / 1e6);
            fflush(stdout);
        }
        // batch invariance of the attention: sequence 0 alone vs inside the batch
        if (b == 64) {
            const int pos = 631; CK(hipMemcpy(dpos, &pos, 4, hipMemcpyHostToDevice));
            AttnFP p; memset(&p, 0, sizeof(p)); p.q = q; p.kc = kc; p.vc = vc; p.pos = dpos; p.mask = dmask; p.part = part; p.out = out; p.H = H; p.S_max = S_max; p.T = T; p.dim = dim; p.nsplit_max = nsm;
            std::vector<float> a(dim), c(dim);
            car_launch_dec_attn_f32_ex(&p, b, 1, 0); CK(hipDeviceSynchronize()); CK(hipMemcpy(a.data(), out, dim * 4, hipMemcpyDeviceToHost));
            car_launch_dec_attn_f32_ex(&p, 1, 0, 0); CK(hipDeviceSynchronize()); CK(hipMemcpy(c.data(), out, dim * 4, hipMemcpyDeviceToHost));
            const bool same = memcmp(a.data(), c.data(), dim * 4) == 0; if (!same) ++fails;
            printf("attn batch invariance (sequence 0 alone vs in a batch of 64): %s\n", same ? "bit-identical" : "BITS DIFFER");
        }
        CK(hipFree(kc)); CK(hipFree(vc)); CK(hipFree(q)); CK(hipFree(part)); CK(hipFree(out)); CK(hipFree(dpos)); CK(hipFree(dmask));
    }
}

// `f32_check one <cfg> <M> <N> <K> <epi> [reps]`: one configuration, launched `reps` times (for rocprofv3 --pmc / --kernel-trace)
static int run_one(int argc, char** argv) {
    if (argc < 7) { printf("usage: f32_check one cfg M N K epi [reps]\n"); return 2; }
    const int cfg = atoi(argv[2]), M = atoi(argv[3]), N = atoi(argv[4]), K = atoi(argv[5]), epi = atoi(argv[6]), reps = argc > 7 ? atoi(argv[7]) : 20;
    const int NL = 6; const size_t wsz = (size_t)N * K;
    float *W, *X, *O; CK(hipMalloc(&W, wsz * NL * 4)); CK(hipMalloc(&X, (size_t)M * K * 4)); CK(hipMalloc(&O, (size_t)M * N * 4));
    hipLaunchKernelGGL(fillf_kernel, dim3(4096), dim3(256), 0, 0, W, wsz * NL, 11u, 0.1f);
    hipLaunchKernelGGL(fillf_kernel, dim3(256), dim3(256), 0, 0, X, (size_t)M * K, 12u, 2.0f);
    CK(hipMemset(O, 0, (size_t)M * N * 4)); CK(hipDeviceSynchronize());
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    auto launch = [&](int it) {
        GemmFP p; memset(&p, 0, sizeof(p)); p.W = W + wsz * (it % NL); p.X = X; p.ldx = K; p.M = M; p.N = N; p.K = K; p.out = O; p.ldo = epi == FEPI_SWIGLU ? N / 2 : N; p.R = O;
        if (car_launch_dec_gemm_f32_cfg(&p, epi, cfg, 0)) { printf("cfg %d rejected\n", cfg); exit(2); }
    };
    launch(0); launch(1); CK(hipDeviceSynchronize());
    CK(hipEventRecord(t0, 0)); for (int i = 0; i < reps; ++i) launch(i); CK(hipEventRecord(t1, 0)); CK(hipEventSynchronize(t1)); CK(hipGetLastError());
    float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
    printf("cfg %d M=%d N=%d K=%d epi %d: %.1f us  %.1f TF\n", cfg, M, N, K, epi, ms * 1000.0 / reps, 2.0 * M * N * K / (ms * 1e-3 / reps) / 1e12);
#ifdef CAR_STAMP
    if (cfg >= 1000) {      // one stamped launch: workgroup lifetimes, phases of wave 0, residency per CU
        const int nwg = 8192; long long* st; CK(hipMalloc(&st, (size_t)nwg * 16 * 8)); CK(hipMemset(st, 0, (size_t)nwg * 16 * 8));
        GemmFP p; memset(&p, 0, sizeof(p)); p.W = W; p.X = X; p.ldx = K; p.M = M; p.N = N; p.K = K; p.out = O; p.ldo = epi == FEPI_SWIGLU ? N / 2 : N; p.R = O; p.stamp = st;
        car_launch_dec_gemm_f32_cfg(&p, epi, cfg, 0); CK(hipDeviceSynchronize());
        std::vector<long long> h((size_t)nwg * 16); CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        long long w0 = -1, w1 = 0; int n = 0; double life = 0, lifec = 0, sync = 0, iss = 0, epi_c = 0; std::vector<int> per_cu(8 * 64, 0);
        for (int i = 0; i < nwg; ++i) { const long long* o = &h[(size_t)i * 16]; if (!o[1]) continue; ++n;
            if (w0 < 0 || o[0] < w0) w0 = o[0]; if (o[1] > w1) w1 = o[1];
            life += (o[1] - o[0]) / 100.0; lifec += o[4] - o[2]; sync += o[5]; iss += o[6]; epi_c += o[4] - o[3];
            const int cu = (int)((o[7] >> 8) & 15), se = (int)((o[7] >> 13) & 7), xcc = (int)(o[8] & 15); per_cu[(xcc * 8 + se) * 8 + (cu & 7)]++; }
        // concurrency: max number of workgroups alive at once on the busiest (xcc, se, cu) is not recoverable from counts alone; print the spread of WGs per CU id
        int used = 0, mx = 0; for (int v : per_cu) { if (v) ++used; if (v > mx) mx = v; }
        printf("  stamped launch: %d workgroups, span %.1f us, mean lifetime %.1f us = %.0f cycles (clock %.2f GHz); of the lifetime: sync(wait+barrier) %.1f %%, DMA issue %.1f %%, epilogue %.1f %%; CU slots used %d, max WGs on one %d\n",
               n, (w1 - w0) / 100.0, life / n, lifec / n, lifec / n / (life / n) / 1e3, 100 * sync / lifec, 100 * iss / lifec, 100 * epi_c / lifec, used, mx);
    }
#endif
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "one")) return run_one(argc, argv);
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    gemm_correctness();
    tiled_correctness();
    if (!quick) gemm_timing();
    attn_check_and_timing();
    printf(fails ? "FAILED: %d checks\n" : "all checks passed\n", fails);
    return fails ? 1 : 0;
}
