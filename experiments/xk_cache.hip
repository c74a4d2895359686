// experiments/xk_cache.hip — round 6: what survives a kernel boundary on MI355X, and what a short dependent weight-streaming kernel can pull.
//   1. warm vs cold: a streaming-read kernel over S bytes, preceded by a kernel that read (a) the SAME S bytes with the same workgroup -> address mapping
//      (L2 of the same XCD, then Infinity Cache), (b) a disjoint region (cold: HBM).  S = 2 ... 128 MB.  If (a) at S <= 32 MB runs at L2 speed, the per-XCD
//      L2 survives the boundary; if it only beats (b) by the Infinity-Cache margin, the boundary invalidates L2 and only the memory-side cache helps.
//   2. "touch" prefetch: a kernel that loads ONE dword per 128-byte line (32x less data through the CU) of region R, then the streaming kernel over R:
//      does a line-touch leave the region warm in the Infinity Cache?
//   3. short-burst bandwidth: the time of ONE dependent streaming kernel of S bytes (graph of 40 distinct regions back to back) for 240 workgroups x 8 waves,
//      all loads in flight at once — the shape of a small-batch decode linear — cold and Infinity-Cache-warm.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/xk_cache.hip -o experiments/xk_cache && experiments/xk_cache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// every workgroup streams its contiguous slice; U 16-byte loads in flight per lane
template <int NT>
__global__ __launch_bounds__(512) void stream_kernel(const u32x4* p, size_t n16, unsigned* sink) {
    const size_t per = n16 / gridDim.x;                       // 16-byte units per workgroup
    const u32x4* q = p + per * blockIdx.x;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i < per; i += (size_t)blockDim.x * 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const size_t j = i + (size_t)u * blockDim.x; v[u] = (u32x4){0, 0, 0, 0}; if (j < per) v[u] = NT ? __builtin_nontemporal_load(q + j) : q[j]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][3];
    }
    if (acc == 0x9e3779b9u) *sink = acc;
}
// one dword per 128-byte line
__global__ __launch_bounds__(256) void touch_kernel(const unsigned* p, size_t nlines, unsigned* sink) {
    unsigned acc = 0;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlines; i += st * 8) {
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const size_t j = i + u * st; v[u] = 0; if (j < nlines) v[u] = p[j * 32]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    if (acc == 0x9e3779b9u) *sink = acc;
}
__global__ void fill_kernel(unsigned* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t st = (size_t)gridDim.x * blockDim.x; for (; i < n; i += st) p[i] = (unsigned)i * 2654435761u; }

int main() {
    const size_t TOTAL = (size_t)3 << 30;                      // 3 GB: far beyond the 256 MB Infinity Cache
    unsigned* buf; CK(hipMalloc(&buf, TOTAL));
    unsigned* sink; CK(hipMalloc(&sink, 4));
    hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, buf, TOTAL / 4);
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](int reps, auto&& body) {       // body(i) enqueues iteration i; returns us per iteration from a captured graph
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < reps; ++i) body(i);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st)); for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ex, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
        return (double)ms * 1000.0 / (3.0 * reps);
    };
    printf("== 1. the same streaming kernel (256 workgroups x 512 lanes, 8 x 16 B per lane in flight) back to back: same region (warm) vs rotating regions (cold)\n");
    printf("%8s | %10s %10s | %10s %10s | %12s\n", "S (MB)", "warm us", "TB/s", "cold us", "TB/s", "empty-chain us");
    const double empty = timed(40, [&](int) { hipLaunchKernelGGL((stream_kernel<0>), dim3(256), dim3(512), 0, st, (const u32x4*)buf, (size_t)0, sink); });
    for (size_t mb : {1, 2, 4, 8, 16, 24, 32, 48, 64, 128, 192}) {
        const size_t S = mb << 20; const int nreg = (int)std::min<size_t>(TOTAL / S, 40);
        const double warm = timed(40, [&](int) { hipLaunchKernelGGL((stream_kernel<0>), dim3(256), dim3(512), 0, st, (const u32x4*)buf, S / 16, sink); });
        const double cold = timed(40, [&](int i) { hipLaunchKernelGGL((stream_kernel<0>), dim3(256), dim3(512), 0, st, (const u32x4*)((char*)buf + (size_t)(i % nreg) * S), S / 16, sink); });
        printf("%8zu | %10.2f %10.2f | %10.2f %10.2f | %12.2f%s\n", mb, warm, S / (warm - empty) / 1e6, cold, S / (cold - empty) / 1e6, empty, (size_t)nreg * S < ((size_t)512 << 20) ? "  (cold set < 512 MB: partly cached)" : "");
    }
    printf("== 1b. the same with non-temporal loads (what the decode linears use for single-use weights)\n");
    for (size_t mb : {8, 32, 128}) {
        const size_t S = mb << 20; const int nreg = (int)std::min<size_t>(TOTAL / S, 40);
        const double warm = timed(40, [&](int) { hipLaunchKernelGGL((stream_kernel<1>), dim3(256), dim3(512), 0, st, (const u32x4*)buf, S / 16, sink); });
        const double cold = timed(40, [&](int i) { hipLaunchKernelGGL((stream_kernel<1>), dim3(256), dim3(512), 0, st, (const u32x4*)((char*)buf + (size_t)(i % nreg) * S), S / 16, sink); });
        printf("%8zu | %10.2f %10.2f | %10.2f %10.2f\n", mb, warm, S / (warm - empty) / 1e6, cold, S / (cold - empty) / 1e6);
    }
    printf("== 2. line-touch prefetch (one dword per 128-byte line, G workgroups x 256 lanes) of region i+1 in front of the stream over region i+1 (regions rotate: cold without the touch)\n");
    for (size_t mb : {8, 16, 40}) {
        const size_t S = mb << 20; const int nreg = (int)std::min<size_t>(TOTAL / S, 40);
        for (int G : {32, 128}) {
            const double both = timed(40, [&](int i) {
                const char* r = (const char*)buf + (size_t)(i % nreg) * S;
                hipLaunchKernelGGL(touch_kernel, dim3(G), dim3(256), 0, st, (const unsigned*)r, S / 128, sink);
                hipLaunchKernelGGL((stream_kernel<1>), dim3(256), dim3(512), 0, st, (const u32x4*)r, S / 16, sink); });
            const double touch = timed(40, [&](int i) { const char* r = (const char*)buf + (size_t)(i % nreg) * S; hipLaunchKernelGGL(touch_kernel, dim3(G), dim3(256), 0, st, (const unsigned*)r, S / 128, sink); });
            printf("S %3zu MB, touch grid %3d: touch alone %8.2f us (%.2f TB/s of lines), touch + stream %8.2f us -> stream after touch %8.2f us\n", mb, G, touch, S / (touch - empty) / 1e6, both, both - touch);
        }
    }
    printf("== 3. one short dependent burst: 240 workgroups x 512 lanes, everything in flight at once (S / 240 per workgroup), 40 distinct regions per graph\n");
    printf("%8s | %10s %10s | %12s %10s\n", "S (MB)", "cold us", "TB/s", "IC-warm us", "TB/s");
    for (size_t mb : {3, 6, 10, 18, 36}) {
        const size_t S = mb << 20;
        const int nreg = 40;
        const double cold = timed(40, [&](int i) { hipLaunchKernelGGL((stream_kernel<1>), dim3(240), dim3(512), 0, st, (const u32x4*)((char*)buf + (size_t)(i % nreg) * ((size_t)48 << 20)), S / 16, sink); });
        const int nw = (int)std::max<size_t>(1, ((size_t)160 << 20) / S);      // the rotating set stays under 160 MB: Infinity-Cache resident, L2 (32 MB) exceeded when nw*S > 32 MB
        const double warm = timed(40, [&](int i) { hipLaunchKernelGGL((stream_kernel<1>), dim3(240), dim3(512), 0, st, (const u32x4*)((char*)buf + (size_t)(i % nw) * S), S / 16, sink); });
        printf("%8zu | %10.2f %10.2f | %12.2f %10.2f   (warm set %d regions)\n", mb, cold, S / (cold - empty) / 1e6, warm, S / (warm - empty) / 1e6, nw);
    }
    return 0;
}
