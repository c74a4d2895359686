// experiments/l2bw.hip — what can a CU pull from L2 / MALL, and through which path?  Test infrastructure (round 3): the LDS-tiled decode GEMM
// (controlar_amd/csrc/decode3.hip) and dec_gemm both sit at ~35 GB/s per CU of operand traffic; this harness measures the ceiling.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/l2bw.hip -o experiments/l2bw && experiments/l2bw
// Every workgroup (256 threads = 4 waves) reads `per_wg` bytes as 1-KiB wave chunks (16 B per lane) from a region of `region` bytes:
//   mode SHARED : all workgroups walk the SAME region in the same order (the X operand of a GEMM: broadcast reads)
//   mode ROTATE : the same region, each workgroup starting at its own offset
//   mode PRIVATE: each workgroup has its own slice of a large buffer (streaming from HBM / MALL)
// paths: VGPR (global_load_dwordx4, U loads in flight per wave) and LDS-DMA (global_load_lds_dwordx4 into a ring, U in flight per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int U>
__global__ __launch_bounds__(256) void read_vgpr(const u32x4* __restrict__ buf, size_t region_chunks, size_t per_wave_chunks, size_t wg_stride_chunks, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t c = ((size_t)blockIdx.x * wg_stride_chunks + (size_t)wave * per_wave_chunks) % region_chunks;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = 0; i < per_wave_chunks; i += U) {
        u32x4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t cc = c + u; if (cc >= region_chunks) cc -= region_chunks; r[u] = buf[cc * 64 + lane]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc[0] ^= r[u][0]; acc[1] += r[u][1]; acc[2] ^= r[u][2]; acc[3] += r[u][3]; }
        c += U; if (c >= region_chunks) c -= region_chunks;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

template <int U>
__global__ __launch_bounds__(256) void read_dma(const u32x4* __restrict__ buf, size_t region_chunks, size_t per_wave_chunks, size_t wg_stride_chunks, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) u32x4 ring[];      // [4 waves][2 * U][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t c = ((size_t)blockIdx.x * wg_stride_chunks + (size_t)wave * per_wave_chunks) % region_chunks;
    u32x4* mine = ring + (size_t)wave * 2 * U * 64;
    // two half-rings of U chunks: issue half h while half 1-h is "consumed" (one ds_read per chunk keeps the LDS side honest)
    u32x4 acc = {0, 0, 0, 0};
    auto issue = [&](int h) {
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t cc = c + u; if (cc >= region_chunks) cc -= region_chunks;
            __builtin_amdgcn_global_load_lds((gptr_t*)(buf + cc * 64 + lane), (lptr_t*)(mine + (size_t)(h * U + u) * 64), 16, 0, 0); }
        c += U; if (c >= region_chunks) c -= region_chunks;
    };
    issue(0);
    int h = 0;
    for (size_t i = 0; i < per_wave_chunks; i += U) {
        if (i + U < per_wave_chunks) { issue(1 - h); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; u += 4) { const u32x4 v = mine[(size_t)(h * U + u) * 64 + lane]; acc[0] ^= v[0]; acc[1] += v[1]; }
        h = 1 - h;
    }
    if ((acc[0] ^ acc[1]) == 0x12345678u) out[0] = 1;
}

int main() {
    const size_t BIG = (size_t)1 << 30;                 // 1 GiB
    u32x4* buf; unsigned* out;
    CK(hipMalloc(&buf, BIG)); CK(hipMalloc(&out, 64)); CK(hipMemset(buf, 0x5a, BIG)); CK(hipMemset(out, 0, 64));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    struct Case { const char* name; size_t region; int mode; };       // mode 0 shared, 1 rotate, 2 private
    const Case cases[] = {{"shared 256 KB  (broadcast)", 256 << 10, 0}, {"shared 2 MB    (broadcast)", 2 << 20, 0}, {"rotate 2 MB    (L2)", 2 << 20, 1},
                          {"rotate 16 MB   (MALL)", 16 << 20, 1}, {"private stream (HBM)", BIG, 2}};
    const int wgs_list[] = {64, 256, 512, 1024};
    const size_t per_wg = 1 << 20;                      // 1 MiB per workgroup
    for (const Case& cs : cases) {
        for (int wgs : wgs_list) {
            const size_t region_chunks = cs.region >> 10, per_wave_chunks = (per_wg >> 10) / 4;
            size_t stride = cs.mode == 0 ? 0 : (cs.mode == 1 ? 37 * 4 + 1 : (per_wg >> 10));
            if (cs.mode == 2 && (size_t)wgs * per_wg > BIG) continue;
            printf("%-28s %4d WGs x 1 MiB:", cs.name, wgs);
            auto run = [&](auto kern, size_t sh, const char* tag) {
                for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), sh, 0, buf, region_chunks, per_wave_chunks, stride, out);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(t0, 0));
                for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), sh, 0, buf, region_chunks, per_wave_chunks, stride, out);
                CK(hipEventRecord(t1, 0)); CK(hipEventSynchronize(t1)); CK(hipGetLastError());
                float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
                const double us = ms * 1000.0 / 5, tb = (double)wgs * per_wg / us / 1e6;
                const int cus = wgs < 256 ? wgs : 256;
                printf("  %s %6.1f us %5.2f TB/s (%4.0f GB/s/CU)", tag, us, tb, tb * 1000.0 / cus);
            };
            run(read_vgpr<4>, 0, "vgpr U=4 ");
            run(read_vgpr<16>, 0, "vgpr U=16");
            run(read_dma<4>, (size_t)4 * 2 * 4 * 1024, "dma U=4 ");
            run(read_dma<8>, (size_t)4 * 2 * 8 * 1024, "dma U=8 ");
            printf("\n"); fflush(stdout);
        }
    }
    return 0;
}
