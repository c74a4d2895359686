// car_common.h — shared device/host helpers for libcontrolar_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

typedef uint16_t bf16_t;   // raw bf16 bits

// A/B and profiling switches (CAR_* environment variables: DESIGN.md §4, tools/*_sweep.py, experiments/) exist only in the DEVELOPMENT build of the library
// (libcontrolar_hip_dev.so, compiled with -DCAR_DEV_KNOBS by build.sh).  The shipped libcontrolar_hip.so contains neither the getenv calls nor the names.
#include <stdlib.h>
#ifdef CAR_DEV_KNOBS
extern "C" int g_car_knob_hits;      // switches actually found set since the counter was last cleared (car_stats.dev_knobs_active; defined in engine.hip)
static inline const char* car_knob_get(const char* name) { const char* v = getenv(name); if (v) __atomic_fetch_add(&g_car_knob_hits, 1, __ATOMIC_RELAXED); return v; }
#define CAR_KNOB(name) car_knob_get(name)
#else
#define CAR_KNOB(name) ((const char*)nullptr)
#endif

__host__ __device__ inline float bf2f(bf16_t v) {
    union { uint32_t u; float f; } c; c.u = ((uint32_t)v) << 16; return c.f;
}
// round-to-nearest-even, identical to torch's float->bfloat16
__host__ __device__ inline bf16_t f2bf(float f) {
    union { uint32_t u; float f; } c; c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// Element-type traits: T = float (exact mode) or bf16_t (fast mode)
template <typename T> struct ET;
template <> struct ET<float> {
    static constexpr int mode = 0;
    __host__ __device__ static inline float ld(const float* p) { return *p; }
    __host__ __device__ static inline void st(float* p, float v) { *p = v; }
    __host__ __device__ static inline float rnd(float v) { return v; }
};
template <> struct ET<bf16_t> {
    static constexpr int mode = 1;
    __host__ __device__ static inline float ld(const bf16_t* p) { return bf2f(*p); }
    __host__ __device__ static inline void st(bf16_t* p, float v) { *p = f2bf(v); }
    __host__ __device__ static inline float rnd(float v) { return bf2f(f2bf(v)); }
};

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

// A decode kernel's parameter block is 150-300 bytes = 3-5 cache lines of the kernarg segment.  hipcc places each s_load next to its first use with an
// s_waitcnt in front of the next, so a cold kernel start walked the lines one scalar-cache miss after the other (five dependent misses in dec_gemm's prologue:
// ISA of round 5).  Touch every line up front — independent s_loads, one wait — and the later field loads hit the scalar cache.
template <int NLINES>
__device__ __forceinline__ void car_kernarg_prefetch() {
    const __attribute__((address_space(4))) unsigned* ka = (const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned t[NLINES];
#pragma unroll
    for (int i = 0; i < NLINES; ++i) t[i] = ka[i * 16];
#pragma unroll
    for (int i = 0; i < NLINES; ++i) asm volatile("" ::"s"(t[i]));
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum for blockDim.x multiple of 64 (<= 1024); `sm` has >= 17 floats
__device__ inline float block_sum(float v, float* sm) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (l == 0) sm[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sm[i];   // fixed order: deterministic
    return r;
}
__device__ inline float block_max(float v, float* sm) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (l == 0) sm[w] = v;
    __syncthreads();
    float r = sm[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, sm[i]);
    return r;
}

__device__ inline float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ inline float gelu_tanh_f(float x) {
    const float k = 0.79788456080286535588f;   // sqrt(2/pi)
    return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}
__device__ inline float silu_f(float x) { return x / (1.0f + expf(-x)); }

enum { ACT_NONE = 0, ACT_GELU_ERF = 1, ACT_GELU_TANH = 2, ACT_SILU = 3 };
enum { BIAS_NONE = 0, BIAS_N = 1, BIAS_M = 2 };
enum { AMODE_PLAIN = 0, AMODE_CONV3 = 1, AMODE_CONV3S2 = 2 };

// Generic GEMM descriptor: C[z][m,n] = epi( alpha * sum_k A[z][m,k] * W[z][n,k] )   (both K-contiguous)
struct GemmP {
    const void* A; const void* W; void* C;
    const void* bias; const void* scale; const void* R;   // bias: T per n or per m; scale: T per n; R: residual T [m,n]
    int M, N, K;
    long lda, ldw, ldc, ldr;
    long sA0, sA1, sW0, sW1, sC0, sC1, sR0, sR1;          // batch strides: z = z0*nb1 + z1
    int nb0, nb1;
    float alpha;
    int bias_mode, act, out_f32, swiglu;
    // AMODE_CONV3: A is NHWC [B,Hin,Win,Cin]; output grid Ho x Wo (= Hin<<ups); K = 9*Cin; M = B*Ho*Wo
    int Ho, Wo, Cin, ups;
    int patch;            // AMODE_CONV3, bf16 kernels: GEMM rows enumerate the output in 16x16 spatial patches (Ho, Wo multiples of 16) so that a
                          // tile's nine taps re-read an 18x18 halo that stays in L2, instead of three full image rows
    const void* zero;     // >= 16 zero bytes in device memory: source of padded / out-of-range chunks of the LDS-DMA loader (set by car_launch_gemm)
    float* gn_part;       // conv3_halo64_kernel only (car_conv3_halo64_ok): if set, the epilogue also writes the GroupNorm stage-1 partials of its OUTPUT
                          // tensor — per (image, 16x16 tile, channel) sum and sum of squares of the stored bf16 values, layout [B][Ho*Wo/256][2][N] =
                          // what gn_partial_vec_kernel writes with one 256-pixel chunk per tile — so the next GroupNorm skips its read-only pass
};
