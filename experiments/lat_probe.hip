// experiments/lat_probe.hip — WHERE does a decode layer's time go?  The product's decode kernels (controlar_amd/csrc/decode2.hip compiled with -DCAR_STAMP:
// every workgroup writes 100-MHz wall-clock stamps of its phases) run as the layer chain engine_generate.hip builds — captured into a hipGraph of NL distinct layers,
// replayed — and the stamps of one replay are reduced to a timeline per kernel of a middle layer:
//   gap     first workgroup entry of this kernel minus the last workgroup exit of the previous one (the dependent-kernel boundary)
//   ramp    median entry minus first entry (dispatch spread over the grid)
//   pro     entry -> prologue done (fused norm)              wait   prologue done -> first operand stage consumed (the first HBM / L2 round trip)
//   loop    first stage -> main loop done                    fold   loop done -> K-slices folded (LDS + barrier)
//   epi     fold done -> all stores issued and retired        span   first entry -> last exit of the kernel
// (medians over workgroups; the stamps cost a few hundred ns per kernel, so the layer total printed beside the un-instrumented product timing calibrates them).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCAR_STAMP -I controlar_amd/csrc experiments/lat_probe.hip -o experiments/lat_probe && experiments/lat_probe [rows=2] [pos=631]
#include "../controlar_amd/csrc/decode2.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
template <typename T> static T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); return p; }

__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        const unsigned a = (x & 0x807fu) | (((x >> 7) & 0x3f) + 64) << 7, b2 = ((x >> 16) & 0x807fu) | ((((x >> 23) & 0x3f) + 64) << 7);
        p[i] = a | (b2 << 16); }
}

struct Dims { int M, D, Fh, H, SA, T, pos; };
struct Bufs { bf16_t *W, *kv, *h, *xn, *att, *mid, *q, *nw; float *part, *rope, *ssq; int *dpos; long long* stamp; size_t per_layer, kvper; };
static bool g_normx = true;
static int g_pf_mode = 0, g_pf_attn = 192, g_pf_lin = 160, g_attn_small = 1, g_overlap = 0, g_pf_kv = 0;
static unsigned* g_cnt = nullptr; static unsigned* g_err = nullptr; static int g_prev_wgs = 0; static hipStream_t g_side = nullptr;      // early launch: per-kernel arrival counters (8 shards x 32 uints)
struct KInfo { std::string name; int wgs; int main; };      // main: workgroups that do the kernel's work (the rest are L2 run-ahead helpers)
static std::vector<KInfo> g_k;          // kernels in launch order (slot = index)

static void layer(const Dims& d, const Bufs& b, int l, int NL, hipStream_t st, int opt) {
    const int D = d.D, Fh = d.Fh, M = d.M;
    bf16_t* w = b.W + b.per_layer * (l % NL);
    bf16_t *wqkv = w, *wo = w + (size_t)3 * D * D, *w13 = wo + (size_t)D * D, *w2 = w13 + (size_t)2 * Fh * D;
    bf16_t* kc = b.kv + b.kvper * 2 * (l % NL); bf16_t* vc = kc + b.kvper;
    const bool fuse = M <= 8, nx = g_normx && M <= 128;        // nx: engine_generate.hip's normalise-on-the-fly flow (every layer but 0 / the control-add layers)
    auto slot = [&](const std::string& n, int wgs, int helpers = 0) { g_k.push_back({n, wgs + helpers, wgs}); return (int)g_k.size() - 1; };
    auto gemm = [&](const char* nm, const bf16_t* W, const bf16_t* X, int N, int K, int epi, GemmDP p) {
        p.W = W; p.X = X; p.M = M; p.N = N; p.K = K;
        int cfg = car_pick_gemm_cfg(M, N, K, epi);
        const int I = cfg / 100, J = (cfg / 10) % 10, Mb = (M + 15) / 16; p.w_nt = (Mb + J - 1) / J == 1;
        char nb[96]; snprintf(nb, sizeof(nb), "%s cfg %d", nm, cfg);
        if (((N / (16 * I)) * ((Mb + J - 1) / J)) % 8) p.pf_wgs = 0;      // the helper's XCD arithmetic assumes the main grid is a multiple of 8
        p.stamp = b.stamp; p.stamp_slot = slot(nb, (N / (16 * I)) * ((Mb + J - 1) / J), p.pf_wgs);
        if (p.ssq_out) p.ssq_ld = N / (16 * (I >= 2 ? 2 : 1));
        if (p.ssq_in) p.ssq_np = D / (16 * (car_pick_gemm_cfg(M, D, D, EPI_RESID) / 100 >= 2 ? 2 : 1));      // (the probe only times: wo's partial count serves both consumers; the buffer holds the maximum)
        hipStream_t ks = st;
#ifdef CAR_EARLY_LAUNCH
        if (g_overlap) {
            const int kidx = p.stamp_slot;
            ks = (kidx & 1) ? g_side : st;
            if (kidx > 0) { p.dep = g_cnt + (size_t)(kidx - 1) * 256; p.dep_n = g_prev_wgs; }
            p.done = g_cnt + (size_t)kidx * 256; p.hs_err = g_err;
            g_prev_wgs = (N / (16 * I)) * ((Mb + J - 1) / J);
        }
#endif
        if (car_launch_dec_gemm_cfg(&p, epi, cfg, ks)) { printf("cfg %d rejected (N=%d K=%d)\n", cfg, N, K); exit(3); }
    };
    GemmDP z; memset(&z, 0, sizeof(z));
    auto norm = [&](const bf16_t* hin) { Norm2P n; memset(&n, 0, sizeof(n)); n.h_in = hin; n.xn = b.xn; n.w = b.nw; n.D = D; n.eps = 1e-5f; n.stamp = b.stamp; n.stamp_slot = slot("rmsnorm2", (M + 3) / 4); car_launch_rmsnorm2(&n, M, st); };
    {
        GemmDP q = z; q.qout = b.q; q.kc = kc; q.vc = vc; q.rope = b.rope; q.pos = b.dpos; q.H = d.H; q.SA = d.SA; q.dim = D;
        if (nx) { q.nw = b.nw; q.neps = 1e-5f; q.nh_in = b.h; q.ssq_in = b.ssq; }
        else if (fuse) { q.nw = b.nw; q.neps = 1e-5f; q.nh_in = b.h; } else norm(b.h);
        gemm(nx ? "normx+wqkv" : (fuse ? "norm+wqkv" : "wqkv"), wqkv, b.xn, 3 * D, D, EPI_QKV, q);
    }
    {
        Attn2P a; memset(&a, 0, sizeof(a)); a.q = b.q; a.kc = kc; a.vc = vc; a.pos = b.dpos; a.out = b.att; a.part = b.part; a.H = d.H; a.SA = d.SA; a.T = d.T; a.dim = D; a.out_packed = 1;
        int ns = 1; while (M * d.H * ns < 1024 && ns < 16) ns *= 2;
        const bool one = (long)M * d.H <= 240;
        if (one) ns = 1;
        if (opt > 0) ns = opt;
        a.nsplit = ns;
        const int av = (one && ns == 1) ? (g_attn_small ? 162 : 160) : ((ns == 1 && M < 128) ? 20 : 40);
        char nb[96]; snprintf(nb, sizeof(nb), "attention v%d nsplit %d (+combine)", av, ns);
        if (g_pf_mode && ns == 1 && (d.H * M) % 8 == 0) {      // helpers need the 1-D grid: one item per workgroup + the helpers behind them
            a.n_seq = M; a.pf_wgs = g_pf_attn; a.pgrid = d.H * M + a.pf_wgs;
            a.pf_p0 = wo; a.pf_b0 = (unsigned)((size_t)D * D * 2); a.pf_p1 = w13; a.pf_b1 = (unsigned)((size_t)2 * Fh * D * 2);
        }
        a.stamp = b.stamp; a.stamp_slot = slot(nb, d.H * M * ns, a.pf_wgs);
        hipStream_t ks = st;
#ifdef CAR_EARLY_LAUNCH
        if (g_overlap) {
            if (av != 162) { printf("early launch needs the small-batch attention (variant 162)\n"); exit(3); }
            const int kidx = a.stamp_slot;
            ks = (kidx & 1) ? g_side : st;
            if (kidx > 0) { a.dep = g_cnt + (size_t)(kidx - 1) * 256; a.dep_n = g_prev_wgs; }
            a.done = g_cnt + (size_t)kidx * 256; a.hs_err = g_err;
            g_prev_wgs = d.H * M;
        }
#endif
        car_launch_dec_attn2_var(&a, M, av, 0, ks);
    }
    { GemmDP q = z; q.h = b.h; if (nx) q.ssq_out = b.ssq;
      if (g_pf_mode == 2) { q.pf_wgs = g_pf_lin; q.pf_p0 = w2; q.pf_b0 = (unsigned)((size_t)D * Fh * 2); }
      gemm(nx ? "wo+ssq" : "wo", wo, b.att, D, D, EPI_RESID, q); }
    {
        GemmDP q = z; q.outp = b.mid;
        if (nx) { q.nw = b.nw; q.neps = 1e-5f; q.nh_in = b.h; q.ssq_in = b.ssq; }
        else if (fuse) { q.nw = b.nw; q.neps = 1e-5f; q.nh_in = b.h; q.pos = b.dpos; } else norm(b.h);
        gemm(nx ? "normx+w1|w3" : (fuse ? "norm+w1|w3" : "w1|w3"), w13, b.xn, 2 * Fh, D, EPI_SWIGLU, q);
    }
    { GemmDP q = z; q.h = b.h; if (nx) q.ssq_out = b.ssq;
      if (g_pf_mode == 1 || g_pf_mode == 2) { q.pf_wgs = g_pf_lin; q.pf_p0 = b.W + b.per_layer * ((l + 1) % NL); q.pf_b0 = (unsigned)((size_t)3 * D * D * 2);
          if (g_pf_kv && (d.H * M) % 8 == 0) { bf16_t* kcn = b.kv + b.kvper * 2 * ((l + 1) % NL); q.pf_kc = kcn; q.pf_vc = kcn + b.kvper; q.pf_pos = b.dpos; q.pf_items = d.H * M; q.pf_SA = d.SA; q.pf_kvb = 2; } }
      gemm(nx ? "w2+ssq" : "w2", w2, b.mid, D, Fh, EPI_RESID, q); }
}

int main(int argc, char** argv) {
    Dims d; d.M = argc > 1 ? atoi(argv[1]) : 2; d.D = 1280; d.Fh = 3584; d.H = 20; d.T = 120; d.pos = argc > 2 ? atoi(argv[2]) : 631; d.SA = 1152;
    const int opt = argc > 3 ? atoi(argv[3]) : 0;        // > 0: force the attention split count
    if (argc > 4) g_normx = atoi(argv[4]) != 0;         // 0: the prologue-norm / separate-norm flow of round 3
    const int NL = argc > 5 ? atoi(argv[5]) : 12, REPS = 20;      // NL distinct layers per graph (4: the weights stay in the Infinity Cache)
    if (argc > 6) g_pf_mode = atoi(argv[6]);           // L2 run-ahead helper workgroups (decode2_params.h CAR_PF_FIELDS): see the mode line printed below
    if (argc > 7) g_pf_attn = atoi(argv[7]);
    if (argc > 8) g_pf_lin = atoi(argv[8]);
    if (argc > 9) g_attn_small = atoi(argv[9]);
    if (argc > 10) g_overlap = atoi(argv[10]);
    if (argc > 11) g_pf_kv = atoi(argv[11]);            // 1: w2's helpers also touch the KV prefixes of the next layer's attention
#ifndef CAR_EARLY_LAUNCH
    if (g_overlap) { printf("early launch: build with -DCAR_EARLY_LAUNCH\n"); return 2; }
#endif          // 1: early launch — the chain alternates between two streams, dependencies through arrival counters (decode2_params.h CAR_HS_FIELDS)        // 0: the round-3 one-launch attention (variant 160) instead of round 6's dec_attn2s_kernel (162)
    const int M16 = (d.M + 15) / 16 * 16;
    Bufs b; memset(&b, 0, sizeof(b));
    b.per_layer = (size_t)(3 * d.D * d.D + d.D * d.D + 2 * d.Fh * d.D + d.D * d.Fh);
    b.W = dalloc<bf16_t>(b.per_layer * NL);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned*)b.W, b.per_layer * NL / 2, 12345u);
    b.kvper = (size_t)d.M * d.H * d.SA * 64;
    b.kv = dalloc<bf16_t>(b.kvper * 2 * NL);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned*)b.kv, b.kvper * 2 * NL / 2, 999u);
    b.h = dalloc<bf16_t>((size_t)M16 * d.D); b.xn = dalloc<bf16_t>((size_t)M16 * d.D); b.att = dalloc<bf16_t>((size_t)M16 * d.D);
    b.mid = dalloc<bf16_t>((size_t)M16 * d.Fh); b.q = dalloc<bf16_t>((size_t)M16 * d.D); b.nw = dalloc<bf16_t>(d.D);
    for (bf16_t* p : {b.h, b.xn, b.att, b.q}) hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, (unsigned*)p, (size_t)M16 * d.D / 2, 7u);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, (unsigned*)b.mid, (size_t)M16 * d.Fh / 2, 8u);
    hipLaunchKernelGGL(fill_kernel, dim3(8), dim3(256), 0, 0, (unsigned*)b.nw, (size_t)d.D / 2, 9u);
    b.part = dalloc<float>((size_t)M16 * d.H * 16 * 66);
    b.ssq = dalloc<float>((size_t)M16 * (d.D / 16)); { std::vector<float> one((size_t)M16 * (d.D / 16), 16.0f); CK(hipMemcpy(b.ssq, one.data(), one.size() * 4, hipMemcpyHostToDevice)); }
    b.rope = dalloc<float>((size_t)d.SA * 64); CK(hipMemset(b.rope, 0, (size_t)d.SA * 64 * 4));
    b.dpos = dalloc<int>(1); CK(hipMemcpy(b.dpos, &d.pos, 4, hipMemcpyHostToDevice));
    const size_t nstamp = (size_t)NL * 8 * 2048 * 8;
    b.stamp = dalloc<long long>(nstamp); CK(hipMemset(b.stamp, 0, nstamp * 8));
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const double wbytes = (double)b.per_layer * 2, kvbytes = (double)d.M * d.H * (d.pos + 1) * 256.0;
    printf("rows %d, position %d, %d distinct layers per graph: %.1f MB of weights + %.1f MB of KV rows per layer; HBM floor %.1f us per layer at 6.3 TB/s\n",
           d.M, d.pos, NL, wbytes / 1e6, kvbytes / 1e6, (wbytes + kvbytes) / 6.3e6);
    hipGraph_t graph = nullptr; hipGraphExec_t ex = nullptr;
    hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    if (g_overlap) {
        CK(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));
        g_cnt = dalloc<unsigned>((size_t)NL * 8 * 256 + 64); g_err = g_cnt + (size_t)NL * 8 * 256;
        CK(hipMemset(g_cnt, 0, ((size_t)NL * 8 * 256 + 64) * 4));
    }
    // early launch: TWO single-branch graphs, one per stream (a two-branch graph is submitted by the host node by node, branch after branch — ~9.5 us per node,
    // measured: the branches never ran side by side), launched back to back every replay; the counters are cleared and the streams joined around them by events
    hipGraph_t graphB = nullptr; hipGraphExec_t exB = nullptr;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    if (g_overlap) CK(hipStreamBeginCapture(g_side, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < NL; ++l) layer(d, b, l, NL, st, opt);
    CK(hipStreamEndCapture(st, &graph));
    if (g_overlap) { CK(hipStreamEndCapture(g_side, &graphB)); CK(hipGraphInstantiate(&exB, graphB, nullptr, nullptr, 0)); printf("EARLY LAUNCH: kernels alternate between two streams (one single-branch graph each), dependencies through arrival counters\n"); }
    if (g_pf_mode) printf("L2 run-ahead helpers: mode %d (1: attention hosts wo + w1|w3, w2 hosts the next wqkv; 2: + wo hosts w2; 3: attention hosts wo + w1|w3 + nothing else), %d helper workgroups beside the attention, %d beside wo / w2\n", g_pf_mode, g_pf_attn, g_pf_lin);
    CK(hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0));
    auto replay = [&]() {
        if (!g_overlap) { CK(hipGraphLaunch(ex, st)); return; }
        CK(hipMemsetAsync(g_cnt, 0, (size_t)NL * 8 * 256 * 4, st)); CK(hipEventRecord(ef, st)); CK(hipStreamWaitEvent(g_side, ef, 0));
        CK(hipGraphLaunch(ex, st)); CK(hipGraphLaunch(exB, g_side));
        CK(hipEventRecord(ej, g_side)); CK(hipStreamWaitEvent(st, ej, 0));
    };
    for (int i = 0; i < 3; ++i) replay();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(t0, st));
    for (int i = 0; i < REPS; ++i) replay();
    CK(hipEventRecord(t1, st)); CK(hipEventSynchronize(t1));
    float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1)); CK(hipGetLastError());
    if (g_overlap) { unsigned e = 0; CK(hipMemcpy(&e, g_err, 4, hipMemcpyDeviceToHost)); printf("early-launch error word: %u (0 = no wait gave up)\n", e); }
    const int KPL = (int)g_k.size() / NL;
    printf("instrumented chain: %.2f us per layer (%d kernels per layer + combine where split)\n", ms * 1000.0 / (REPS * NL), KPL);
    std::vector<long long> hs(nstamp);
    CK(hipMemcpy(hs.data(), b.stamp, nstamp * 8, hipMemcpyDeviceToHost));
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%-34s %5s | %6s %6s | %6s %6s %6s %6s %6s | %6s   (us; 100 MHz stamps)\n", "kernel", "wgs", "gap", "ramp", "pro", "wait", "loop", "fold", "epi", "span");
    for (int L : {NL > 6 ? 5 : 1, NL > 6 ? 6 : 2}) {
        double prev_exit = -1;
        { // last exit of the previous layer's last kernel
            const int s = L * KPL - 1; long long mx = 0; const int n = std::min(g_k[s].wgs, 2048);
            for (int w = 0; w < n; ++w) mx = std::max(mx, hs[((size_t)s * 2048 + w) * 8 + 5]);
            prev_exit = (double)mx;
        }
        double layer_start = prev_exit;
        for (int k = 0; k < KPL; ++k) {
            const int s = L * KPL + k; const int n = std::min(g_k[s].wgs, 2048);
            long long first = 0, last = 0; std::vector<double> e0, ph[5], hd, iss;
            for (int w = 0; w < n; ++w) {
                const long long* t = &hs[((size_t)s * 2048 + w) * 8];
                if (!t[0]) continue;
                if (!first || t[0] < first) first = t[0];
                last = std::max(last, t[5] ? t[5] : t[0]);
                if (w >= g_k[s].main) { hd.push_back((double)((t[5] ? t[5] : t[0]) - t[0]) / 100.0); continue; }      // helper: lifetime only
                e0.push_back((double)t[0]);
                if (t[6]) iss.push_back((double)(t[6] - t[0]) / 100.0);
                long long prevt = t[0];
                for (int i = 1; i <= 5; ++i) { if (t[i]) { ph[i - 1].push_back((double)(t[i] - prevt) / 100.0); prevt = t[i]; } else ph[i - 1].push_back(0.0); }
            }
            const double ramp = (med(e0) - (double)first) / 100.0;
            printf("%-34s %5d | %6.2f %6.2f | %6.2f %6.2f %6.2f %6.2f %6.2f | %6.2f\n", g_k[s].name.c_str(), g_k[s].wgs, ((double)first - prev_exit) / 100.0, ramp,
                   med(ph[0]), med(ph[1]), med(ph[2]), med(ph[3]), med(ph[4]), (double)(last - first) / 100.0);
            if (!iss.empty()) printf("%34s       prologue loads all issued %.2f us after entry (median)\n", "", med(iss));
            if (!hd.empty()) { std::sort(hd.begin(), hd.end()); printf("%34s %5zu helper workgroups: lifetime median %.2f, max %.2f us\n", "", hd.size(), hd[hd.size() / 2], hd.back()); }
            prev_exit = (double)last;
        }
        printf("  layer %d: %.2f us from the previous layer's last exit to this layer's last exit\n", L, (prev_exit - layer_start) / 100.0);
    }
    return 0;
}
