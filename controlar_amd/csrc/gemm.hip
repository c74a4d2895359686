// gemm.hip — the MFMA workhorse of the compute-bound stages (ViT linears, control MLPs,
// prefill, VQ convs/attention):  C[z] = epi(alpha * A[z] · W[z]^T), both operands K-contiguous.
//   * bf16 path: 128x128x32 tiles, 4 waves (2x2), each wave 4x4 fragments of
//     v_mfma_f32_16x16x32_bf16, register-staged double-buffered LDS (80-byte padded rows).
//   * fp32 path ("exact" mode): 64x64x16 tiles of plain v_fma_f32 (k-ordered fp32 chain).
//   * AMODE_CONV3 turns the A loader into an implicit-GEMM 3x3 gather over NHWC activations
//     with zero padding and an optional fused nearest x2 upsample (vq_model.py:375-379).
// The epilogue reproduces the reference's rounding points (one rounding per torch op).
#include "car_common.h"

template <typename T>
__device__ inline float epi_value(const GemmP& p, const T* bias, const T* scale, const T* R, long zR, int m, int n, float v) {
    v *= p.alpha;
    if (p.bias_mode == BIAS_N) v += ET<T>::ld(bias + n);
    else if (p.bias_mode == BIAS_M) v += ET<T>::ld(bias + m);
    v = ET<T>::rnd(v);
    if (p.act == ACT_GELU_ERF) v = ET<T>::rnd(gelu_erf_f(v));
    else if (p.act == ACT_GELU_TANH) v = ET<T>::rnd(gelu_tanh_f(v));
    else if (p.act == ACT_SILU) v = ET<T>::rnd(silu_f(v));
    if (scale) v = ET<T>::rnd(v * ET<T>::ld(scale + n));
    if (R) v = ET<T>::rnd(v + ET<T>::ld(R + zR + (long)m * p.ldr + n));
    return v;
}

// ---- A-operand row descriptor (per thread, constant over the K loop)
struct ARow { long base; int y, x; bool ok; };

struct Geo { int M, Cin, Ho, Wo, ups; long lda; };
template <int AMODE>
__device__ inline ARow make_arow(const Geo p, int m) {
    ARow r; r.ok = m < p.M; r.base = 0; r.y = 0; r.x = 0;
    if (AMODE == AMODE_PLAIN) { r.base = (long)m * p.lda; }
    else {
        const int hw = p.Ho * p.Wo;
        const int b = m / hw, rem = m - b * hw;
        r.y = rem / p.Wo; r.x = rem - r.y * p.Wo;
        if (AMODE == AMODE_CONV3S2) r.base = (long)b * (p.Ho * 2) * (p.Wo * 2);
        else r.base = (long)b * (p.Ho >> p.ups) * (p.Wo >> p.ups);   // in pixels
    }
    return r;
}
// element offset of A[m, k] (k multiple of the chunk width), or -1 if the chunk is zero padding
template <int AMODE>
__device__ inline long a_off(const Geo p, const ARow r, int k) {
    if (!r.ok) return -1;
    if (AMODE == AMODE_PLAIN) return r.base + k;
    const int tap = k / p.Cin, c = k - tap * p.Cin;
    if (AMODE == AMODE_CONV3S2) {
        // Downsample (vq_model.py:382-396): F.pad(x, (0,1,0,1)) then conv3x3 stride 2, no padding: taps (2y+ty, 2x+tx), zero past the edge
        const int Hin = p.Ho * 2, Win = p.Wo * 2, yy = 2 * r.y + tap / 3, xx = 2 * r.x + tap % 3;
        if (yy >= Hin || xx >= Win) return -1;
        return (r.base + (long)yy * Win + xx) * p.Cin + c;
    }
    const int yy = r.y + tap / 3 - 1, xx = r.x + tap % 3 - 1;
    if (yy < 0 || yy >= p.Ho || xx < 0 || xx >= p.Wo) return -1;
    return (r.base + (long)(yy >> p.ups) * (p.Wo >> p.ups) + (xx >> p.ups)) * p.Cin + c;
}

// =========================================================================== bf16 MFMA
#define BM 128
#define BN 128
#define BKK 32
#define LDS_LD 40   // bf16 elements per padded row (80 B)

template <int AMODE>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (BM + BN) * LDS_LD];
#define SA(buf) (smem + (buf) * (BM + BN) * LDS_LD)
#define SB(buf) (smem + (buf) * (BM + BN) * LDS_LD + BM * LDS_LD)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z, z0 = z / p.nb1, z1 = z - z0 * p.nb1;
    const bf16_t* A = (const bf16_t*)p.A + z0 * p.sA0 + z1 * p.sA1;
    const bf16_t* W = (const bf16_t*)p.W + z0 * p.sW0 + z1 * p.sW1;
    const long zC = z0 * p.sC0 + z1 * p.sC1, zR = z0 * p.sR0 + z1 * p.sR1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // load assignment: 2 A chunks + 2 W chunks of 16 B per thread per k-tile
    const int lrow0 = tid >> 2, lrow1 = lrow0 + 64, lkc = (tid & 3) * 8;
    const Geo geo = { p.M, p.Cin, p.Ho, p.Wo, p.ups, p.lda };
    const ARow ar0 = make_arow<AMODE>(geo, m0 + lrow0), ar1 = make_arow<AMODE>(geo, m0 + lrow1);
    const bool wok0 = (n0 + lrow0) < p.N, wok1 = (n0 + lrow1) < p.N;
    const bf16_t* w0p = W + (long)(n0 + lrow0) * p.ldw + lkc;
    const bf16_t* w1p = W + (long)(n0 + lrow1) * p.ldw + lkc;

    uint4 ra0, ra1, rb0, rb1;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    auto gload = [&](int kt) {
        const int k = kt * BKK + lkc;
        long o0 = a_off<AMODE>(geo, ar0, k), o1 = a_off<AMODE>(geo, ar1, k);
        ra0 = zero4; ra1 = zero4; rb0 = zero4; rb1 = zero4;
        if (o0 >= 0) ra0 = *(const uint4*)(A + o0);
        if (o1 >= 0) ra1 = *(const uint4*)(A + o1);
        if (wok0) rb0 = *(const uint4*)(w0p + (long)kt * BKK);
        if (wok1) rb1 = *(const uint4*)(w1p + (long)kt * BKK);
    };
    auto sstore = [&](int buf) {
        *(uint4*)(SA(buf) + lrow0 * LDS_LD + lkc) = ra0;
        *(uint4*)(SA(buf) + lrow1 * LDS_LD + lkc) = ra1;
        *(uint4*)(SB(buf) + lrow0 * LDS_LD + lkc) = rb0;
        *(uint4*)(SB(buf) + lrow1 * LDS_LD + lkc) = rb1;
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BKK;
    gload(0); sstore(0);
    __syncthreads();
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        bf16x8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(SA(buf) + (wm * 64 + i * 16 + fr) * LDS_LD + fk);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8*)(SB(buf) + (wn * 64 + j * 16 + fr) * LDS_LD + fk);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: each wave stages 16x64 fp32 strips of its accumulators through LDS so that the
    // (large) per-element epilogue runs in a rolled loop and global stores are row-contiguous.
    const bf16_t* bias = (const bf16_t*)p.bias; const bf16_t* scale = (const bf16_t*)p.scale; const bf16_t* R = (const bf16_t*)p.R;
    float* strip = (float*)smem + wave * (16 * 68);
    const int swiglu = p.swiglu, out_f32 = p.out_f32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 q = acc[i][j];
            float* d = strip + ((lane >> 4) * 4) * 68 + j * 16 + fr;
            d[0] = q[0]; d[68] = q[1]; d[136] = q[2]; d[204] = q[3];
        }
        __builtin_amdgcn_wave_barrier();
        const int mb = m0 + wm * 64 + i * 16, nb = n0 + wn * 64;
        if (swiglu) {
            // column blocks of 16 alternate w1 | w3 (packed at load time): out[m, n/2] = silu(a) * c
            const int c = lane & 31, src = (c >> 4) * 32 + (c & 15);
            for (int rr = (lane >> 5); rr < 16; rr += 2) {
                const int m = mb + rr, n = nb + src;
                if (m < p.M && n < p.N) {
                    const float a1 = bf2f(f2bf(strip[rr * 68 + src])), c3 = bf2f(f2bf(strip[rr * 68 + src + 16]));
                    const float sl = bf2f(f2bf(silu_f(a1)));
                    ((bf16_t*)p.C)[zC + (long)m * p.ldc + (nb >> 1) + c] = f2bf(sl * c3);
                }
            }
        } else {
            for (int rr = 0; rr < 16; ++rr) {
                const int m = mb + rr, n = nb + lane;
                if (m < p.M && n < p.N) {
                    const float v = epi_value<bf16_t>(p, bias, scale, R, zR, m, n, strip[rr * 68 + lane]);
                    if (out_f32) ((float*)p.C)[zC + (long)m * p.ldc + n] = v;
                    else ((bf16_t*)p.C)[zC + (long)m * p.ldc + n] = f2bf(v);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// =========================================================================== fp32 exact
template <int AMODE>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) float sA[2][16][68];
    __shared__ __attribute__((aligned(16))) float sB[2][16][68];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int z = blockIdx.z, z0 = z / p.nb1, z1 = z - z0 * p.nb1;
    const float* A = (const float*)p.A + z0 * p.sA0 + z1 * p.sA1;
    const float* W = (const float*)p.W + z0 * p.sW0 + z1 * p.sW1;
    const long zC = z0 * p.sC0 + z1 * p.sC1, zR = z0 * p.sR0 + z1 * p.sR1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int lrow = tid >> 2, lkc = (tid & 3) * 4;
    const Geo geo = { p.M, p.Cin, p.Ho, p.Wo, p.ups, p.lda };
    const ARow ar = make_arow<AMODE>(geo, m0 + lrow);
    const bool wok = (n0 + lrow) < p.N;
    const float* wp = W + (long)(n0 + lrow) * p.ldw + lkc;
    float4 ra, rb;
    const float4 zero4 = make_float4(0, 0, 0, 0);
    auto gload = [&](int kt) {
        long o = a_off<AMODE>(geo, ar, kt * 16 + lkc);
        ra = zero4; rb = zero4;
        if (o >= 0) ra = *(const float4*)(A + o);
        if (wok) rb = *(const float4*)(wp + (long)kt * 16);
    };
    auto sstore = [&](int buf) {
        sA[buf][lkc + 0][lrow] = ra.x; sA[buf][lkc + 1][lrow] = ra.y; sA[buf][lkc + 2][lrow] = ra.z; sA[buf][lkc + 3][lrow] = ra.w;
        sB[buf][lkc + 0][lrow] = rb.x; sB[buf][lkc + 1][lrow] = rb.y; sB[buf][lkc + 2][lrow] = rb.z; sB[buf][lkc + 3][lrow] = rb.w;
    };
    float acc[4][4] = {};
    const int nk = p.K / 16;
    gload(0); sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 a = *(const float4*)&sA[buf][k][ty * 4];
            const float4 b = *(const float4*)&sB[buf][k][tx * 4];
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }
    const float* bias = (const float*)p.bias; const float* scale = (const float*)p.scale; const float* R = (const float*)p.R;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m < p.M && n < p.N)
                ((float*)p.C)[zC + (long)m * p.ldc + n] = epi_value<float>(p, bias, scale, R, zR, m, n, acc[i][j]);
        }
}

// host launchers -------------------------------------------------------------------------
extern "C" void car_launch_gemm(int mode, int amode, const GemmP* pp, hipStream_t st) {
    GemmP p = *pp;
    if (p.nb0 <= 0) p.nb0 = 1;
    if (p.nb1 <= 0) p.nb1 = 1;
    if (mode == 1) {
        dim3 g((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.nb0 * p.nb1);
        if (amode == AMODE_PLAIN) hipLaunchKernelGGL(gemm_bf16_kernel<AMODE_PLAIN>, g, dim3(256), 0, st, p);
        else if (amode == AMODE_CONV3S2) hipLaunchKernelGGL(gemm_bf16_kernel<AMODE_CONV3S2>, g, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(gemm_bf16_kernel<AMODE_CONV3>, g, dim3(256), 0, st, p);
    } else {
        dim3 g((p.N + 63) / 64, (p.M + 63) / 64, p.nb0 * p.nb1);
        if (amode == AMODE_PLAIN) hipLaunchKernelGGL(gemm_f32_kernel<AMODE_PLAIN>, g, dim3(256), 0, st, p);
        else if (amode == AMODE_CONV3S2) hipLaunchKernelGGL(gemm_f32_kernel<AMODE_CONV3S2>, g, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(gemm_f32_kernel<AMODE_CONV3>, g, dim3(256), 0, st, p);
    }
}
