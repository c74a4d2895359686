// experiments/xk_fresh.hip — round 6: what does the FIRST load of a dependent kernel cost, by what it reads?
// A decode layer at 2 rows is five kernels whose first MFMA waits 1.4-2.4 us after kernel entry (experiments/lat_probe) although a cold 3 MB burst over 240
// workgroups costs 0.5 us of kernel time (experiments/xk_cache, part 3).  Candidates: (a) reading lines the PREVIOUS kernel just wrote from another XCD
// (dirty in a remote L2 until the end-of-kernel write-back), (b) scalar-load chains, (c) TLB.  Chain per iteration: producer P (80 workgroups write a 5 KB
// activation buffer + 640 B of statistics, by plain / write-through sc1 / non-temporal stores) -> consumer C (240 workgroups x 512 lanes): stamp, every lane loads
// 16 B of the activation buffer, wait, stamp, 8 x 1 KiB per wave of cold weights, wait, stamp.  Medians over workgroups of (first-load latency, weight burst).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/xk_fresh.hip -o experiments/xk_fresh && experiments/xk_fresh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>      // 0 plain, 1 sc1 write-through (relaxed agent atomic store), 2 non-temporal, 3: do not write at all
__global__ __launch_bounds__(512) void producer(unsigned long long* act, int n8, unsigned long long v) {
    const int i = blockIdx.x * 16 + (threadIdx.x & 15);      // 80 workgroups x 16 lanes x 8 B = 10 KB
    if (threadIdx.x >= 16 || i >= n8) return;
    if (MODE == 0) act[i] = v + i;
    else if (MODE == 1) __hip_atomic_store(act + i, v + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 2) __builtin_nontemporal_store(v + i, act + i);
}
template <int LD>        // 0 plain loads of the activation, 1 sc1 (relaxed agent atomic) loads
__global__ __launch_bounds__(512) void consumer(const unsigned long long* act, int n8, const u32x4* w, long long* stamps, int slot, unsigned* sink) {
    const int wg = blockIdx.x;
    long long t0 = 0, t1 = 0, t2 = 0;
    if (threadIdx.x == 0) t0 = (long long)wall_clock64();
    unsigned long long a;
    const int ai = (threadIdx.x * 2) % n8;
    if (LD == 0) a = act[ai]; else a = __hip_atomic_load(act + ai, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(a));
    if (threadIdx.x == 0) t1 = (long long)wall_clock64();
    const u32x4* q = w + (size_t)wg * (512 * 8) + threadIdx.x;
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(q + u * 512);
    unsigned acc = (unsigned)a;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][3];
    asm volatile("" ::"v"(acc));
    if (threadIdx.x == 0) { t2 = (long long)wall_clock64(); long long* s = stamps + ((size_t)slot * 256 + wg) * 4; s[0] = t0; s[1] = t1; s[2] = t2; }
    if (acc == 0x9e3779b9u) *sink = acc;
}
__global__ void fill_kernel(unsigned* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t st = (size_t)gridDim.x * blockDim.x; for (; i < n; i += st) p[i] = (unsigned)i * 2654435761u; }

int main() {
    const int NIT = 24;
    const size_t WB = (size_t)240 * 512 * 8 * 16;          // 15.7 MB of "weights" per consumer launch
    unsigned* wbuf; CK(hipMalloc(&wbuf, WB * NIT));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, wbuf, WB * NIT / 4);
    const int n8 = 1280;                                    // 10 KB activation buffer
    unsigned long long* act; CK(hipMalloc(&act, (size_t)n8 * 8 * NIT)); CK(hipMemset(act, 0, (size_t)n8 * 8 * NIT));
    long long* stamps; CK(hipMalloc(&stamps, (size_t)NIT * 256 * 4 * 8));
    unsigned* sink; CK(hipMalloc(&sink, 4));
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto run = [&](const char* name, int pmode, int ld, bool same_act) {
        CK(hipMemset(stamps, 0, (size_t)NIT * 256 * 4 * 8));
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < NIT; ++i) {
            unsigned long long* a = act + (same_act ? 0 : (size_t)i * n8);
            switch (pmode) {
                case 0: hipLaunchKernelGGL((producer<0>), dim3(80), dim3(512), 0, st, a, n8, (unsigned long long)i); break;
                case 1: hipLaunchKernelGGL((producer<1>), dim3(80), dim3(512), 0, st, a, n8, (unsigned long long)i); break;
                case 2: hipLaunchKernelGGL((producer<2>), dim3(80), dim3(512), 0, st, a, n8, (unsigned long long)i); break;
                default: hipLaunchKernelGGL((producer<3>), dim3(80), dim3(512), 0, st, a, n8, (unsigned long long)i); break;
            }
            const u32x4* w = (const u32x4*)((const char*)wbuf + (size_t)i * WB);
            if (ld) hipLaunchKernelGGL((consumer<1>), dim3(240), dim3(512), 0, st, a, n8, w, stamps, i, sink);
            else hipLaunchKernelGGL((consumer<0>), dim3(240), dim3(512), 0, st, a, n8, w, stamps, i, sink);
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ex, st));
        CK(hipStreamSynchronize(st));
        std::vector<long long> hs((size_t)NIT * 256 * 4);
        CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> l1, l2, span;
        for (int i = 4; i < NIT; ++i) {
            long long first = 0, last = 0;
            for (int w = 0; w < 240; ++w) { const long long* s = &hs[((size_t)i * 256 + w) * 4]; if (!s[0]) continue; l1.push_back((s[1] - s[0]) / 100.0); l2.push_back((s[2] - s[1]) / 100.0);
                if (!first || s[0] < first) first = s[0]; last = std::max(last, s[2]); }
            span.push_back((last - first) / 100.0);
        }
        auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
        auto p90 = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() * 9 / 10]; };
        printf("%-86s first load: median %.2f p90 %.2f us | weight burst %.2f p90 %.2f us | kernel span %.2f us\n", name, med(l1), p90(l1), med(l2), p90(l2), med(span));
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
    };
    run("activation NOT written by the predecessor, distinct buffers (cold HBM lines)", 3, 0, false);
    run("activation NOT written, same buffer every iteration (warm)", 3, 0, true);
    run("predecessor wrote it with plain stores; plain loads", 0, 0, false);
    run("predecessor wrote it with plain stores, same buffer every iteration; plain loads", 0, 0, true);
    run("predecessor wrote it write-through (sc1); plain loads", 1, 0, false);
    run("predecessor wrote it write-through (sc1), same buffer every iteration; plain loads", 1, 0, true);
    run("predecessor wrote it non-temporal; plain loads", 2, 0, false);
    run("predecessor wrote it with plain stores; sc1 loads", 0, 1, false);
    run("predecessor wrote it write-through (sc1); sc1 loads", 1, 1, true);
    return 0;
}
