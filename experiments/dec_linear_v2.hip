// EXPERIMENT (round-2 candidate, NOT part of libcontrolar_hip.so, not yet run on hardware).
//
// dec_linear v2: the decode-step weight-streaming skinny GEMM without split-K partials.
//   out[m][n] = epi( sum_k X[m][k] * W[n][k] ),  m < b <= 64 (one chain), bf16 in, fp32 accumulate.
// Motivation (profiles/r01_pmc_FETCH_SIZE_b256.txt): v1 covers the chip by splitting K across
// workgroups and leaves fp32 partials [KS][b][N] for the consumer to sum — 1.55x the algorithmic weight bytes per layer and one
// extra kernel (swiglu_parts) per layer.  With 2-4 concurrent chains per step the chip is covered by the chains, so a
// workgroup can own its 64 output rows over the WHOLE K:
//   * X is restaged through LDS in K-chunks of 256 (double-buffered, one barrier per chunk), shared by the 4 waves;
//   * every wave streams its 16-row block of the MFMA-fragment-packed weights (same image as v1: engine.hip pack_decode_bf16),
//     8 x 1 KiB non-temporal loads per chunk, prefetched one chunk ahead;
//   * epilogues run in-kernel: EPI_BF16 (wqkv -> dec_attn reads bf16 directly), EPI_SWIGLU (w1|w3 -> mid, the
//     block-16 interleave puts the (a, c) pair in adjacent waves), EPI_F32 (logits / KS > 1 partials as in v1).
// Expected per chain-layer: -11 MB of partial traffic, -1 kernel, wqkv/w13 at 60/112 workgroups per chain.
//
// Validate with experiments/test_dec_linear_v2.cpp (host reference included) before wiring it into engine.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ inline float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ inline bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ inline float silu_f(float x) { return x / (1.0f + expf(-x)); }

enum { EPI_F32 = 0, EPI_BF16 = 1, EPI_SWIGLU = 2 };

struct Lin2P {
    const bf16_t* W;      // packed [N/16][K/32][64 lanes][8]
    const bf16_t* X;      // [b][K] bf16
    void* out;            // EPI_F32: float [KS][b][N];  EPI_BF16: bf16 [b][N];  EPI_SWIGLU: bf16 [b][N/2]
    int b, N, K, KS;
};

constexpr int CK = 256;               // K-chunk staged in LDS
constexpr int LDX = CK + 8;           // padded row (bf16 elements): 528 B rows -> conflict-free 16-lane b128 reads

template <int NB, int EPI>
__global__ __launch_bounds__(256) void dec_linear_v2_kernel(Lin2P p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t xs[];     // [2][16*NB][LDX]  (+ epilogue exchange reuses buffer 0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KC = p.K / p.KS, nch = KC / CK;                       // KC must be a multiple of CK
    const int ks = blockIdx.y, k0 = ks * KC;
    const int rb = blockIdx.x * 4 + wave;
    const bool active = rb * 16 < p.N;
    const u32x4* wp = (const u32x4*)p.W + ((long)rb * (p.K / 32) + (k0 / 32)) * 64 + lane;
    constexpr int ROWS = 16 * NB, CPR = CK / 8, NLD = ROWS * CPR / 256;   // 16-byte pieces per thread per chunk (NB=4: 8)

    u32x4 wa[8], wb[8];
    const u32x4 zw = (u32x4){0u, 0u, 0u, 0u};
    uint4 xr[NLD];
    auto wload = [&](u32x4 (&w)[8], int ch) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { w[i] = zw; if (active) w[i] = __builtin_nontemporal_load(wp + (long)(ch * 8 + i) * 64); }
    };
    auto xload = [&](int ch) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * 256, m = c / CPR, kc = (c - m * CPR) * 8;
            xr[i] = make_uint4(0, 0, 0, 0);
            if (m < p.b) xr[i] = *(const uint4*)(p.X + (long)m * p.K + k0 + ch * CK + kc);
        }
    };
    auto xstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * 256, m = c / CPR, kc = (c - m * CPR) * 8;
            *(uint4*)(xs + (buf * ROWS + m) * LDX + kc) = xr[i];
        }
    };

    f32x4 acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int xo = (lane & 15) * LDX + (lane >> 4) * 8;
    auto compute = [&](const u32x4 (&w)[8], int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x8 a = *(const bf16x8*)&w[i];
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const bf16x8 x = *(const bf16x8*)(xs + (buf * ROWS + n * 16) * LDX + xo + i * 32);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, x, acc[n], 0, 0, 0);
            }
        }
    };

    // prologue: chunk 0 weights in flight, X chunk 0 staged
    wload(wa, 0);
    xload(0); xstore(0);
    __syncthreads();
    for (int ch = 0; ch < nch; ch += 2) {
        // even chunk in (wa, buf 0); prefetch odd chunk into (wb, regs)
        if (ch + 1 < nch) { wload(wb, ch + 1); xload(ch + 1); }
        compute(wa, 0);
        if (ch + 1 < nch) xstore(1);
        __syncthreads();
        if (ch + 1 >= nch) break;
        if (ch + 2 < nch) { wload(wa, ch + 2); xload(ch + 2); }
        compute(wb, 1);
        if (ch + 2 < nch) xstore(0);
        __syncthreads();
    }

    // D[row = (lane>>4)*4 + r][col = lane&15]: n = rb*16 + (lane>>4)*4 + r, m = nb*16 + (lane&15)
    const int n0 = rb * 16 + (lane >> 4) * 4;
    if (EPI == EPI_F32) {
        if (!active) return;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int m = n * 16 + (lane & 15);
            if (m < p.b) *(f32x4*)((float*)p.out + ((long)ks * p.b + m) * p.N + n0) = acc[n];
        }
    } else if (EPI == EPI_BF16) {
        if (!active) return;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int m = n * 16 + (lane & 15);
            if (m < p.b) {
                uint2 o; o.x = (unsigned)f2bf(acc[n][0]) | ((unsigned)f2bf(acc[n][1]) << 16); o.y = (unsigned)f2bf(acc[n][2]) | ((unsigned)f2bf(acc[n][3]) << 16);
                *(uint2*)((bf16_t*)p.out + (long)m * p.N + n0) = o;
            }
        }
    } else {
        // SwiGLU: row-blocks alternate w1 | w3 (block-16 interleave): wave 2j holds a = w1 x, wave 2j+1 holds c = w3 x for the same
        // 16 hidden units.  Odd waves park c (rounded to bf16, as the reference's Linear output) in LDS; even waves finish.
        float* ex = (float*)xs;                       // [2 pairs][NB][64 lanes][4]  (all MFMA reads of xs are behind the last barrier)
        const int pair = wave >> 1;
        if (wave & 1) {
#pragma unroll
            for (int n = 0; n < NB; ++n) *(f32x4*)(ex + ((pair * NB + n) * 64 + lane) * 4) = acc[n];
        }
        __syncthreads();
        if (!(wave & 1) && active) {
            const int h0 = (rb >> 1) * 16 + (lane >> 4) * 4;      // hidden unit index of r = 0
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int m = n * 16 + (lane & 15);
                const f32x4 cc = *(const f32x4*)(ex + ((pair * NB + n) * 64 + lane) * 4);
                if (m < p.b) {
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a1 = bf2f(f2bf(acc[n][r])), c3 = bf2f(f2bf(cc[r]));     // gpt_t2i.py:217 rounding points
                        o[r] = bf2f(f2bf(silu_f(a1))) * c3;
                    }
                    uint2 q; q.x = (unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16); q.y = (unsigned)f2bf(o[2]) | ((unsigned)f2bf(o[3]) << 16);
                    *(uint2*)((bf16_t*)p.out + (long)m * (p.N / 2) + h0) = q;
                }
            }
        }
    }
}

template <int EPI>
static void launch_v2(const Lin2P& p, hipStream_t st) {
    dim3 g((p.N + 63) / 64, p.KS);
    const int NB = (p.b + 15) / 16;
    static bool attr = false;      // NB = 4 needs 67,584 B of dynamic LDS (> the 64 KiB default cap)
    if (!attr) { (void)hipFuncSetAttribute((const void*)dec_linear_v2_kernel<4, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024); attr = true; }
    if (NB <= 1) hipLaunchKernelGGL((dec_linear_v2_kernel<1, EPI>), g, dim3(256), (size_t)2 * 16 * LDX * 2, st, p);
    else if (NB == 2) hipLaunchKernelGGL((dec_linear_v2_kernel<2, EPI>), g, dim3(256), (size_t)2 * 32 * LDX * 2, st, p);
    else hipLaunchKernelGGL((dec_linear_v2_kernel<4, EPI>), g, dim3(256), (size_t)2 * 64 * LDX * 2, st, p);
}
extern "C" void exp_launch_dec_linear_v2(const Lin2P* p, int epi, hipStream_t st) {
    if (epi == EPI_BF16) launch_v2<EPI_BF16>(*p, st);
    else if (epi == EPI_SWIGLU) launch_v2<EPI_SWIGLU>(*p, st);
    else launch_v2<EPI_F32>(*p, st);
}
