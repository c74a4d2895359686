// experiments/queue_check.hip — do kernels from two HIP streams run CONCURRENTLY on this box at all?  (Follow-up of ws_check: an attention
// kernel and the other chain's linears on two streams take exactly their serial sum.)  Spin kernels of a fixed wall time and a small grid:
// two of them on two streams take 1x the time if the hardware queues overlap, 2x if the device runs one kernel at a time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/queue_check.hip -o experiments/queue_check && experiments/queue_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void spin_kernel(long long ticks, int* sink) {          // wall_clock64: constant 100 MHz counter
    const long long t0 = wall_clock64();
    int n = 0;
    while (wall_clock64() - t0 < ticks) ++n;
    if (n == -1) *sink = n;
}

int main() {
    hipStream_t s[4]; for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t t0, t1, j[4]; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); for (auto& x : j) CK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    int* sink; CK(hipMalloc(&sink, 4));
    const long long us100 = 100 * 100;                              // 100 us at 100 MHz
    auto run = [&](int nstreams, int grid, int reps) {
        for (int w = 0; w < 2; ++w) for (int i = 0; i < nstreams; ++i) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, s[i], us100, sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(t0, s[0]));
        for (int i = 1; i < nstreams; ++i) CK(hipStreamWaitEvent(s[i], t0, 0));
        for (int r = 0; r < reps; ++r) for (int i = 0; i < nstreams; ++i) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, s[i], us100, sink);
        for (int i = 1; i < nstreams; ++i) { CK(hipEventRecord(j[i], s[i])); CK(hipStreamWaitEvent(s[0], j[i], 0)); }
        CK(hipEventRecord(t1, s[0])); CK(hipEventSynchronize(t1));
        float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
        printf("%d stream(s) x %d spin kernels of 100 us, grid %5d x 256 threads: %8.1f us total = %.2f x one stream's %d kernels\n", nstreams, reps, grid, ms * 1000.f, ms * 1000.f / (reps * 100.f), reps);
    };
    for (int grid : {64, 1024, 8192}) { run(1, grid, 10); run(2, grid, 10); run(4, grid, 10); }
    return 0;
}
