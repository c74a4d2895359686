// experiments/gemm_mid.hip — what do the LDS-tiled GEMMs of the compute-bound stages (controlar_amd/csrc/gemm.hip) achieve at the DECODE
// linears' shapes of a large chain (M = 384 / 768 rows)?  dec_gemm's K-split register tiles take 20.0 / 17.3 / 5.9 / 10.4 us for
// wqkv / w1|w3 / wo / w2 at 384 rows (profiles/r02_ws_check_half_period.txt) = 15 % of the MFMA peak at best; the step's linears are
// 1.15 TFLOP per 768 rows = 0.5 ms at the peak against 3.2 ms measured.  Plain GEMMs only (no decode epilogues): a yes/no on the tiling
// before a fused-epilogue kernel is written.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I controlar_amd/csrc experiments/gemm_mid.hip -o experiments/gemm_mid && experiments/gemm_mid
#include "../controlar_amd/csrc/gemm.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        const unsigned a = (x & 0x807fu) | (((x >> 7) & 0x3f) + 64) << 7, b2 = ((x >> 16) & 0x807fu) | ((((x >> 23) & 0x3f) + 64) << 7);
        p[i] = a | (b2 << 16); }
}

int main() {
    struct Shape { const char* name; int N, K; };
    const Shape shapes[] = {{"wqkv", 3840, 1280}, {"w1|w3", 7168, 1280}, {"wo", 1280, 1280}, {"w2", 1280, 3584}};
    const int NL = 8;
    void* zero; CK(hipMalloc(&zero, 256)); CK(hipMemset(zero, 0, 256));
    const size_t sh = (size_t)G2_NS * G2_STAGE * 2;
    CK(hipFuncSetAttribute((const void*)gemm_bf16_glds_kernel<AMODE_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int M : {384, 768}) for (const Shape& s : shapes) {
        const size_t wsz = (size_t)s.N * s.K;
        bf16_t *W, *A, *C; CK(hipMalloc(&W, wsz * NL * 2)); CK(hipMalloc(&A, (size_t)M * s.K * 2)); CK(hipMalloc(&C, (size_t)M * s.N * 2));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned*)W, wsz * NL / 2, 11u);
        hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, (unsigned*)A, (size_t)M * s.K / 2, 12u);
        CK(hipDeviceSynchronize());
        const double gflop = 2.0 * M * s.N * s.K / 1e9;
        printf("M=%-3d %-6s N=%-5d K=%-4d %5.1f GFLOP:", M, s.name, s.N, s.K, gflop);
        for (int kind = 0; kind < 2; ++kind) {
            auto launch = [&](int it) {
                GemmP p; memset(&p, 0, sizeof(p)); p.A = A; p.W = W + wsz * (it % NL); p.C = C; p.lda = s.K; p.ldw = s.K; p.ldc = s.N; p.M = M; p.N = s.N; p.K = s.K;
                p.alpha = 1.f; p.nb0 = 1; p.nb1 = 1; p.zero = zero;
                if (kind == 0) hipLaunchKernelGGL(gemm_bf16_kernel<AMODE_PLAIN>, dim3((s.N + BN - 1) / BN, (M + BM - 1) / BM, 1), dim3(256), 0, 0, p);
                else hipLaunchKernelGGL(gemm_bf16_glds_kernel<AMODE_PLAIN>, dim3((s.N + BN - 1) / BN, (M + G2_BM - 1) / G2_BM, 1), dim3(512), sh, 0, p);
            };
            for (int i = 0; i < 3; ++i) launch(i);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, 0));
            for (int i = 0; i < 48; ++i) launch(i);
            CK(hipEventRecord(t1, 0)); CK(hipEventSynchronize(t1)); CK(hipGetLastError());
            float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
            const double us = ms * 1000.0 / 48;
            printf("  %s %6.1f us = %4.0f TFLOP/s (%d workgroups)", kind == 0 ? "128x128x32 register-staged" : "256x128x64 LDS-DMA", us, gflop / us * 1e3,
                   kind == 0 ? ((s.N + BN - 1) / BN) * ((M + BM - 1) / BM) : ((s.N + BN - 1) / BN) * ((M + G2_BM - 1) / G2_BM));
        }
        printf("\n"); fflush(stdout);
        CK(hipFree(W)); CK(hipFree(A)); CK(hipFree(C));
    }
    return 0;
}
