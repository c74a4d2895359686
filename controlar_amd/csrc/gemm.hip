// gemm.hip — the MFMA workhorse of the compute-bound stages (ViT linears, control MLPs,
// prefill, VQ convs/attention):  C[z] = epi(alpha * A[z] · W[z]^T), both operands K-contiguous.
//   * bf16 path: 128x128x32 tiles, 4 waves (2x2), each wave 4x4 fragments of
//     v_mfma_f32_16x16x32_bf16, register-staged double-buffered LDS (80-byte padded rows).
//   * fp32 path ("exact" mode): 64x64x16 tiles of plain v_fma_f32 (k-ordered fp32 chain).
//   * AMODE_CONV3 turns the A loader into an implicit-GEMM 3x3 gather over NHWC activations
//     with zero padding and an optional fused nearest x2 upsample (vq_model.py:375-379).
// The epilogue reproduces the reference's rounding points (one rounding per torch op).
#include "car_common.h"
#include <cstdio>
#include <cstdlib>

template <typename T>
__device__ __forceinline__ float epi_value(const GemmP& p, const T* bias, const T* scale, const T* R, long zR, int m, long mrow, int n, float v) {
    v *= p.alpha;
    if (p.bias_mode == BIAS_N) v += ET<T>::ld(bias + n);
    else if (p.bias_mode == BIAS_M) v += ET<T>::ld(bias + m);
    v = ET<T>::rnd(v);
    if (p.act == ACT_GELU_ERF) v = ET<T>::rnd(gelu_erf_f(v));
    else if (p.act == ACT_GELU_TANH) v = ET<T>::rnd(gelu_tanh_f(v));
    else if (p.act == ACT_SILU) v = ET<T>::rnd(silu_f(v));
    if (scale) v = ET<T>::rnd(v * ET<T>::ld(scale + n));
    if (R) v = ET<T>::rnd(v + ET<T>::ld(R + zR + mrow * p.ldr + n));
    return v;
}

// ---- A-operand row descriptor (per thread, constant over the K loop)
struct ARow { long base; int y, x; bool ok; };

struct Geo { int M, Cin, Ho, Wo, ups; long lda; int patch; };
// patch order: m = ((b * (Ho/16) + ty) * (Wo/16) + tx) * 256 + py * 16 + px  ->  pixel (b, ty*16 + py, tx*16 + px)
__device__ inline void patch_decode(int Ho, int Wo, int m, int& b, int& y, int& x) {
    const int tw = Wo >> 4, th = Ho >> 4, tile = m >> 8, within = m & 255;
    b = tile / (tw * th); const int t2 = tile - b * (tw * th), ty = t2 / tw, tx = t2 - ty * tw;
    y = ty * 16 + (within >> 4); x = tx * 16 + (within & 15);
}
// row of C / R that GEMM row m addresses (the NHWC pixel index under patch order, m itself otherwise)
__device__ __forceinline__ long out_row(const GemmP& p, int m) {
    if (!p.patch) return m;
    int b, y, x; patch_decode(p.Ho, p.Wo, m, b, y, x);
    return ((long)b * p.Ho + y) * p.Wo + x;
}
template <int AMODE>
__device__ inline ARow make_arow(const Geo p, int m) {
    ARow r; r.ok = m < p.M; r.base = 0; r.y = 0; r.x = 0;
    if (AMODE == AMODE_PLAIN) { r.base = (long)m * p.lda; }
    else {
        const int hw = p.Ho * p.Wo;
        int b = m / hw; const int rem = m - b * hw;
        r.y = rem / p.Wo; r.x = rem - r.y * p.Wo;
        if (AMODE == AMODE_CONV3 && p.patch) patch_decode(p.Ho, p.Wo, m, b, r.y, r.x);
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

// ---- one 16 x 64 fp32 strip (ld 68) of a wave's accumulators -> C, through the epilogue.  Vector form: a lane owns 8 consecutive
// columns of one row (16-byte bias / scale / residual loads, one 16-byte bf16 store or two float4 stores) whenever the row
// strides and N allow it; the element-wise form (lane = column) remains for ragged shapes (e.g. the 1025-wide score matrices).
__device__ __forceinline__ void ld8bf(const bf16_t* p, float (&v)[8]) {
    const uint4 u = *(const uint4*)p; const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
}
enum { EP_VEC = 0, EP_SCALAR = 1, EP_SWIGLU = 2 };
// gate of the paired-column epilogue: GemmP::swiglu 1 = SiLU (LlamaGen FeedForward), 2 = tanh-GELU (T5 gated-gelu, "gelu_new")
__device__ __forceinline__ float gate_act(int kind, float a) { return kind == 2 ? gelu_tanh_f(a) : silu_f(a); }
__device__ __forceinline__ int strip_path(const GemmP& p, long zC, long zR) {
    if (p.swiglu) return EP_SWIGLU;
    const bool vec = (p.N & 7) == 0 && (p.ldc & 7) == 0 && (!p.R || (p.ldr & 7) == 0) && p.bias_mode != BIAS_M &&
                     ((zC | zR) & 7) == 0 && ((uintptr_t)p.C & 15) == 0 && (!p.R || ((uintptr_t)p.R & 15) == 0);
    return vec ? EP_VEC : EP_SCALAR;
}
template <int PATH>
__device__ __forceinline__ void store_strip(const GemmP& p, const float* strip, int mb, int nb, long zC, long zR, int lane) {
    const bf16_t* bias = (const bf16_t*)p.bias; const bf16_t* scale = (const bf16_t*)p.scale; const bf16_t* R = (const bf16_t*)p.R;
    if (PATH == EP_SWIGLU) {
        // column blocks of 16 alternate w1 | w3 (packed at load time): out[m, n/2] = silu(a) * c ; lane = (row, 8 of the 32 outputs)
        const int rr = lane >> 2, c = (lane & 3) * 8, src = (c >> 4) * 32 + (c & 15), m = mb + rr;
        if (m < p.M && nb + src + 24 <= p.N && (p.ldc & 7) == 0 && (zC & 7) == 0) {
            unsigned o[4];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float a0 = bf2f(f2bf(strip[rr * 68 + src + e])), c0 = bf2f(f2bf(strip[rr * 68 + src + 16 + e]));
                const float a1 = bf2f(f2bf(strip[rr * 68 + src + e + 1])), c1 = bf2f(f2bf(strip[rr * 68 + src + 17 + e]));
                o[e >> 1] = (unsigned)f2bf(bf2f(f2bf(gate_act(p.swiglu, a0))) * c0) | ((unsigned)f2bf(bf2f(f2bf(gate_act(p.swiglu, a1))) * c1) << 16);
            }
            *(uint4*)((bf16_t*)p.C + zC + out_row(p, m) * p.ldc + (nb >> 1) + c) = make_uint4(o[0], o[1], o[2], o[3]);
        } else if (m < p.M) {
            for (int e = 0; e < 8; ++e) {
                const int n = nb + src + e;
                if (n + 16 < p.N) {
                    const float a1 = bf2f(f2bf(strip[rr * 68 + src + e])), c3 = bf2f(f2bf(strip[rr * 68 + src + 16 + e]));
                    ((bf16_t*)p.C)[zC + out_row(p, m) * p.ldc + (nb >> 1) + c + e] = f2bf(bf2f(f2bf(gate_act(p.swiglu, a1))) * c3);
                }
            }
        }
        return;
    }
    if (PATH == EP_VEC) {
        const int ec = (lane & 7) * 8, n = nb + ec;
        float bv[8], sv[8];
        if (n < p.N) {
            if (p.bias_mode == BIAS_N) ld8bf(bias + n, bv);
            if (scale) ld8bf(scale + n, sv);
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int rr = pass * 8 + (lane >> 3), m = mb + rr;
            if (m < p.M && n < p.N) {
                const long mr = out_row(p, m);
                const float4 s0 = *(const float4*)(strip + rr * 68 + ec), s1 = *(const float4*)(strip + rr * 68 + ec + 4);
                float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, rv8[8];
                if (R) ld8bf(R + zR + mr * p.ldr + n, rv8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = v[e] * p.alpha;
                    if (p.bias_mode == BIAS_N) x += bv[e];
                    x = bf2f(f2bf(x));
                    if (p.act == ACT_GELU_ERF) x = bf2f(f2bf(gelu_erf_f(x)));
                    else if (p.act == ACT_GELU_TANH) x = bf2f(f2bf(gelu_tanh_f(x)));
                    else if (p.act == ACT_SILU) x = bf2f(f2bf(silu_f(x)));
                    if (scale) x = bf2f(f2bf(x * sv[e]));
                    if (R) x = bf2f(f2bf(x + rv8[e]));
                    v[e] = x;
                }
                if (p.out_f32) {
                    float* d = (float*)p.C + zC + mr * p.ldc + n;
                    *(float4*)d = make_float4(v[0], v[1], v[2], v[3]); *(float4*)(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    uint4 o;
                    o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16); o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                    o.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16); o.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
                    *(uint4*)((bf16_t*)p.C + zC + mr * p.ldc + n) = o;
                }
            }
        }
        return;
    }
    for (int rr = 0; rr < 16; ++rr) {
        const int m = mb + rr, n = nb + lane;
        if (m < p.M && n < p.N) {
            const long mr = out_row(p, m);
            const float v = epi_value<bf16_t>(p, bias, scale, R, zR, m, mr, n, strip[rr * 68 + lane]);
            if (p.out_f32) ((float*)p.C)[zC + mr * p.ldc + n] = v;
            else ((bf16_t*)p.C)[zC + mr * p.ldc + n] = f2bf(v);
        }
    }
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
    const Geo geo = { p.M, p.Cin, p.Ho, p.Wo, p.ups, p.lda, p.patch };
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
    float* strip = (float*)smem + wave * (16 * 68);
    const int path = strip_path(p, zC, zR);       // chosen once: each path's strip loop stays small enough to unroll (acc stays in registers)
#define STRIP_LOOP(PATH)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
            const f32x4 q = acc[i][j];                                                            \
            float* d = strip + ((lane >> 4) * 4) * 68 + j * 16 + fr;                              \
            d[0] = q[0]; d[68] = q[1]; d[136] = q[2]; d[204] = q[3];                              \
        }                                                                                         \
        __builtin_amdgcn_wave_barrier();                                                          \
        store_strip<PATH>(p, strip, m0 + wm * 64 + i * 16, n0 + wn * 64, zC, zR, lane);           \
        __builtin_amdgcn_wave_barrier();                                                          \
    }
    if (path == EP_VEC) { STRIP_LOOP(EP_VEC) } else if (path == EP_SWIGLU) { STRIP_LOOP(EP_SWIGLU) } else { STRIP_LOOP(EP_SCALAR) }
#undef STRIP_LOOP
}

// =========================================================================== bf16 MFMA, LDS-DMA staged (large-M GEMMs / convs)
// 256x128x64 tiles, 8 waves (4x2) x (4x4) fragments of v_mfma_f32_16x16x32_bf16, THREE 48-KiB LDS stages filled by
// global_load_lds (16 B per lane: no staging registers, no ds_write pass).  Stage t+2 is issued while stage t is consumed and
// stage t+1 is still in flight: the per-wave wait is a counted vmcnt (one stage = 6 DMA pieces per wave may stay outstanding)
// followed by a raw s_barrier — __syncthreads() would drain the DMA queue (cdna_hip_programming.md §5).  A stage row is
// 64 k = 8 chunks of 16 B; chunk c of row r is stored at slot c ^ (r & 7): the DMA destination is lane-linear, so the swizzle is
// applied to the per-lane GLOBAL address, and the fragment reads (16 rows x one chunk per ds_read_b128 group) spread over the
// bank groups.  Padded taps of the implicit-GEMM conv, rows >= M and weight rows >= N read a zero page.
// Requirements: K % 64 == 0, 16-byte aligned rows; conv: Cin % 64 == 0.  Used when the grid has >= 1 tile per CU.
#define G2_BM 256
#define G2_BK 64
#define G2_NS 3
#define G2_STAGE ((G2_BM + BN) * G2_BK)          // bf16 elements per stage
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int AMODE>
__global__ __launch_bounds__(512) void gemm_bf16_glds_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem2[];                   // 3 x 48 KiB
#define S2A(buf) (smem2 + (buf) * G2_STAGE)
#define S2B(buf) (smem2 + (buf) * G2_STAGE + G2_BM * G2_BK)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z, z0 = z / p.nb1, z1 = z - z0 * p.nb1;
    const bf16_t* A = (const bf16_t*)p.A + z0 * p.sA0 + z1 * p.sA1;
    const bf16_t* W = (const bf16_t*)p.W + z0 * p.sW0 + z1 * p.sW1;
    const long zC = z0 * p.sC0 + z1 * p.sC1, zR = z0 * p.sR0 + z1 * p.sR1;
    const int m0 = blockIdx.y * G2_BM, n0 = blockIdx.x * BN;
    const Geo geo = { p.M, p.Cin, p.Ho, p.Wo, p.ups, p.lda, p.patch };
    // loader: wave w, pass i covers tile rows i*64 + w*8 .. +8; lane = (row-in-8, slot); global chunk = slot ^ row-in-8
    const int lr = lane >> 3, gchunk = ((lane & 7) ^ lr) * 8;
    ARow ar[4]; const bf16_t* wrow[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) ar[i] = make_arow<AMODE>(geo, m0 + i * 64 + wave * 8 + lr);
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int r = i * 64 + wave * 8 + lr; wrow[i] = (n0 + r) < p.N ? W + (long)(n0 + r) * p.ldw + gchunk : nullptr; }
    const bf16_t* zero = (const bf16_t*)p.zero;
    // K position of the NEXT stage to issue, kept incrementally (stages are issued in increasing k order): no divisions per step.
    // conv: k = tap * Cin + c0, tap = (dy+1)*3 + (dx+1); Cin % 64 == 0 so a stage never straddles two taps.
    int k_next = 0, c0 = 0, dy = -1, dx = -1;
    const int Wi = p.Wo >> p.ups;
    auto issue = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* src = zero;
            if (AMODE == AMODE_PLAIN) { if (ar[i].ok) src = A + ar[i].base + k_next + gchunk; }
            else {
                const int yy = ar[i].y + dy, xx = ar[i].x + dx;
                if (ar[i].ok && yy >= 0 && yy < p.Ho && xx >= 0 && xx < p.Wo)
                    src = A + (ar[i].base + (long)(yy >> p.ups) * Wi + (xx >> p.ups)) * p.Cin + c0 + gchunk;
            }
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(S2A(buf) + (i * 64 + wave * 8) * G2_BK), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16_t* src = wrow[i] ? wrow[i] + k_next : zero;
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(S2B(buf) + (i * 64 + wave * 8) * G2_BK), 16, 0, 0);
        }
        k_next += G2_BK; c0 += G2_BK;
        if (AMODE != AMODE_PLAIN && c0 == p.Cin) { c0 = 0; if (++dx == 2) { dx = -1; ++dy; } }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = p.K / G2_BK;
    const int fr = lane & 15, fq = lane >> 4;
    issue(0);
    if (nk > 1) issue(1);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's 6 pieces of stage kt have landed; stage kt+1 (6 more) may still be in flight
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();             // ... so have all other waves'; and everyone is done reading the stage consumed at kt-1
        if (kt + 2 < nk) issue(buf >= 1 ? buf - 1 : G2_NS - 1);              // stage kt+2 -> buffer (kt+2) % 3 == (kt-1) % 3, free now
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int r = wm * 64 + i * 16 + fr; a[i] = *(const bf16x8*)(S2A(buf) + r * G2_BK + (((kk * 4 + fq) ^ (r & 7)) << 3)); }
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int r = wn * 64 + j * 16 + fr; b[j] = *(const bf16x8*)(S2B(buf) + r * G2_BK + (((kk * 4 + fq) ^ (r & 7)) << 3)); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        buf = buf + 1 == G2_NS ? 0 : buf + 1;
    }
    __syncthreads();                              // the epilogue strips reuse the stage memory

    // ---- epilogue (same as gemm_bf16_kernel): 16x64 fp32 strips through LDS, rolled per-element epilogue, row-contiguous stores
    float* strip = (float*)smem2 + wave * (16 * 68);
    const int path = strip_path(p, zC, zR);       // chosen once: each path's strip loop stays small enough to unroll (acc stays in registers)
#define STRIP_LOOP(PATH)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
            const f32x4 q = acc[i][j];                                                            \
            float* d = strip + ((lane >> 4) * 4) * 68 + j * 16 + fr;                              \
            d[0] = q[0]; d[68] = q[1]; d[136] = q[2]; d[204] = q[3];                              \
        }                                                                                         \
        __builtin_amdgcn_wave_barrier();                                                          \
        store_strip<PATH>(p, strip, m0 + wm * 64 + i * 16, n0 + wn * 64, zC, zR, lane);           \
        __builtin_amdgcn_wave_barrier();                                                          \
    }
    if (path == EP_VEC) { STRIP_LOOP(EP_VEC) } else if (path == EP_SWIGLU) { STRIP_LOOP(EP_SWIGLU) } else { STRIP_LOOP(EP_SCALAR) }
#undef STRIP_LOOP
}

// =========================================================================== 3x3 conv with an LDS-resident halo (VQ decoder)
// The nine taps of a 3x3 convolution re-read every input pixel nine times; as an implicit GEMM fed from global memory that is
// 590 KB of L2/fabric traffic per 128x128x1152 tile, and the fabric (~6 TB/s) — not the matrix cores — bounds the kernel at
// ~15 % of the MFMA peak (profiles/r02_*trace*: 376 TFLOP/s on the 128->128 conv at 512x512).  Here a workgroup owns one
// 16x16 output patch x 128 output channels: the patch's input halo (18x18 pixels, or 10x10 under the folded nearest x2
// upsample, vq_model.py:375-379) x 128 channels is DMA'd into LDS once per 128-channel chunk and every tap's A fragments are
// read from it; only the weights stream (3-stage LDS-DMA ring, counted vmcnt + raw s_barrier as gemm_bf16_glds_kernel).
// 8 waves: wave (wm, wn) owns output rows py = 4*wm..+4 of the patch (16 pixels each = one A fragment) x 64 output channels.
// LDS: halo [HD*HD pixels][128 ch] bf16, chunk c (16 B) of pixel h at slot c ^ (h & 15); weight stage [128 n][64 k], chunk c of
// row r at slot c ^ (r & 7).  Requirements: Cin % 128 == 0, Cout % 128 == 0 handled by the n grid, Ho % 16 == 0, Wo % 16 == 0.
#define CH_HALO_MAX (18 * 18)
template <int UPS>
__global__ __launch_bounds__(512) void conv3_halo_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem3[];
    constexpr int HD = (16 >> UPS) + 2, HP = HD * HD;
    bf16_t* halo = smem3;                                   // [CH_HALO_MAX][128]
    bf16_t* wst = smem3 + CH_HALO_MAX * 128;                // 3 x [128][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * BN, tile = blockIdx.y;
    const int tw = p.Wo >> 4, th = p.Ho >> 4;
    const int b = tile / (tw * th), t2 = tile - b * (tw * th), ty = t2 / tw, tx = t2 - ty * tw;
    const int Hin = p.Ho >> UPS, Win = p.Wo >> UPS;
    const int hy0 = ((16 * ty - 1) >> UPS), hx0 = ((16 * tx - 1) >> UPS);      // arithmetic shift: -1 >> 1 == -1
    const bf16_t* A = (const bf16_t*)p.A + (long)b * Hin * Win * p.Cin;
    const bf16_t* W = (const bf16_t*)p.W;
    const bf16_t* zero = (const bf16_t*)p.zero;
    const int fr = lane & 15, fq = lane >> 4;
    // weight loader: wave w, pass i covers rows i*64 + w*8 .. +8 of the 128 n rows
    const int lr = lane >> 3, gchunk = ((lane & 7) ^ lr) * 8;
    const bf16_t* wrow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int r = i * 64 + wave * 8 + lr; wrow[i] = (n0 + r) < p.N ? W + (long)(n0 + r) * p.ldw + gchunk : nullptr; }
    auto issue_w = [&](int buf, int kglob) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16_t* src = wrow[i] ? wrow[i] + kglob : zero;
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(wst + buf * (BN * G2_BK) + (i * 64 + wave * 8) * G2_BK), 16, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nchunk = p.Cin >> 7;
    for (int ch = 0; ch < nchunk; ++ch) {
        // ---- halo of this 128-channel chunk: one wave instruction = 4 pixels x 256 B
        __syncthreads();                                   // everyone is done with the previous chunk's halo and weight stages
        for (int q = wave; q * 4 < HP; q += 8) {
            const int pix = q * 4 + (lane >> 4), slot = lane & 15;
            const int hy = pix / HD, hx = pix - hy * HD, iy = hy0 + hy, ix = hx0 + hx;
            const bf16_t* src = zero;
            if (pix < HP && iy >= 0 && iy < Hin && ix >= 0 && ix < Win) src = A + ((long)iy * Win + ix) * p.Cin + ch * 128 + ((slot ^ (pix & 15)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(halo + q * 4 * 128), 16, 0, 0);
        }
        // weight ring for this chunk: k-steps s = tap*2 + c64 (18 of them); global k = tap*Cin + ch*128 + c64*64
        auto kglob = [&](int s_) { return (s_ >> 1) * p.Cin + ch * 128 + (s_ & 1) * 64; };
        issue_w(0, kglob(0)); issue_w(1, kglob(1));
        int buf = 0;
        for (int s_ = 0; s_ < 18; ++s_) {
            // weights of step s_ landed (2 pieces per wave per stage; the next stage may stay in flight); at s_ == 0 this also covers the halo
            if (s_ == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (s_ + 1 < 18) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s_ + 2 < 18) issue_w(buf >= 1 ? buf - 1 : 2, kglob(s_ + 2));
            const int tap = s_ >> 1, c64 = s_ & 1, dy = tap / 3 - 1, dx = tap % 3 - 1;
            const bf16_t* ws = wst + buf * (BN * G2_BK);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 a[4], bb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int h = (((16 * ty + wm * 4 + i + dy) >> UPS) - hy0) * HD + (((16 * tx + fr + dx) >> UPS) - hx0);
                    a[i] = *(const bf16x8*)(halo + h * 128 + (((c64 * 8 + kk * 4 + fq) ^ (h & 15)) << 3));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int r = wn * 64 + j * 16 + fr; bb[j] = *(const bf16x8*)(ws + r * G2_BK + (((kk * 4 + fq) ^ (r & 7)) << 3)); }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bb[j], acc[i][j], 0, 0, 0);
            }
            buf = buf + 1 == 3 ? 0 : buf + 1;
        }
    }
    __syncthreads();
    // ---- epilogue: out = rnd(rnd(acc + bias) + R), 16-byte loads/stores: lane = (pixel rr of the patch row, 8 consecutive channels)
    const bf16_t* bias = (const bf16_t*)p.bias; const bf16_t* R = (const bf16_t*)p.R;
    float* strip = (float*)smem3 + wave * (16 * 68);
    const int ec = (lane & 7) * 8, en = n0 + wn * 64 + ec;
    float bv[8];
    {
        const uint4 u = bias ? *(const uint4*)(bias + en) : make_uint4(0, 0, 0, 0);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { bv[2 * e] = __uint_as_float(w[e] << 16); bv[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 q = acc[i][j];
            float* d = strip + ((lane >> 4) * 4) * 68 + j * 16 + fr;
            d[0] = q[0]; d[68] = q[1]; d[136] = q[2]; d[204] = q[3];
        }
        __builtin_amdgcn_wave_barrier();
        const int oy = 16 * ty + wm * 4 + i;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int rr = pass * 8 + (lane >> 3);
            const long mr = ((long)b * p.Ho + oy) * p.Wo + 16 * tx + rr;
            const float4 s0 = *(const float4*)(strip + rr * 68 + ec), s1 = *(const float4*)(strip + rr * 68 + ec + 4);
            float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e] + bv[e]));
            if (R) {
                const uint4 u = *(const uint4*)(R + mr * p.ldr + en);
                const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] = v[2 * e] + __uint_as_float(w[e] << 16); v[2 * e + 1] = v[2 * e + 1] + __uint_as_float(w[e] & 0xffff0000u); }
            }
            uint4 o;
            o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16); o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
            o.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16); o.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
            *(uint4*)((bf16_t*)p.C + mr * p.ldc + en) = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// conv3_halo64_kernel (round 3): the same tile (16x16 output pixels x 128 output channels, 8 waves, the same accumulation order per output:
// 64-channel groups ascending, taps ascending inside a group — the OLD kernel walks taps inside 128-channel groups, so results agree to fp32
// summation order, not bit for bit) with HALF the LDS: the halo is staged in 64-channel groups ([HP][64] bf16 = 41 KB instead of 83 KB) and the weight
// ring has two stages (one tap of 64 channels = 16 KiB in flight).  75 KB per workgroup puts TWO workgroups on a CU: conv3_halo_kernel's single
// 131-KB workgroup leaves the matrix cores idle during its halo load, at every weight-stage wait and through its epilogue (27 us per tile against
// 8 us of MFMA work); with two, one workgroup's waits sit under the other's MFMAs.
// LDS: halo pixel h = 8 chunks of 16 B, chunk c at slot c ^ (h & 7); weight stage [128 n][64 k], chunk c of row r at slot c ^ (r & 7).
// Requirements: Cin % 64 == 0, Ho % 16 == 0, Wo % 16 == 0.
#define CH64_HALO_PIX 328                                   // 18 x 18 = 324 rounded up to whole 8-pixel DMA pieces
template <int UPS, int GNP>
__global__ __launch_bounds__(512, 4) void conv3_halo64_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem4[];
    constexpr int HD = (16 >> UPS) + 2, HP = HD * HD;
    bf16_t* halo = smem4;                                   // [CH64_HALO_PIX][64]
    bf16_t* wst = smem4 + CH64_HALO_PIX * 64;               // 2 x [128][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * BN, tile = blockIdx.y;
    const int tw = p.Wo >> 4, th = p.Ho >> 4;
    const int b = tile / (tw * th), t2 = tile - b * (tw * th), ty = t2 / tw, tx = t2 - ty * tw;
    const int Hin = p.Ho >> UPS, Win = p.Wo >> UPS;
    const int hy0 = ((16 * ty - 1) >> UPS), hx0 = ((16 * tx - 1) >> UPS);      // arithmetic shift: -1 >> 1 == -1
    const bf16_t* A = (const bf16_t*)p.A + (long)b * Hin * Win * p.Cin;
    const bf16_t* W = (const bf16_t*)p.W;
    const bf16_t* zero = (const bf16_t*)p.zero;
    const int fr = lane & 15, fq = lane >> 4;
    // weight loader: wave w, pass i covers rows i*64 + w*8 .. +8 of the 128 n rows
    const int lr = lane >> 3, gchunk = ((lane & 7) ^ lr) * 8;
    const bf16_t* wrow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int r = i * 64 + wave * 8 + lr; wrow[i] = (n0 + r) < p.N ? W + (long)(n0 + r) * p.ldw + gchunk : nullptr; }
    auto issue_w = [&](int buf, int kglob) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16_t* src = wrow[i] ? wrow[i] + kglob : zero;
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(wst + buf * (BN * G2_BK) + (i * 64 + wave * 8) * G2_BK), 16, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // halo pixel index of output pixel (row 4*wm + i, column fr) under tap (dy, dx), hoisted per tap
    const int ngrp = p.Cin >> 6;
    for (int cg = 0; cg < ngrp; ++cg) {
        __syncthreads();                                   // everyone is done with the previous group's halo and weight stages
        // ---- halo of this 64-channel group: one wave instruction = 8 pixels x 128 B
        for (int q = wave; q * 8 < HP; q += 8) {
            const int pix = q * 8 + (lane >> 3), slot = lane & 7;
            const int hy = pix / HD, hx = pix - hy * HD, iy = hy0 + hy, ix = hx0 + hx;
            const bf16_t* src = zero;
            if (pix < HP && iy >= 0 && iy < Hin && ix >= 0 && ix < Win) src = A + ((long)iy * Win + ix) * p.Cin + cg * 64 + ((slot ^ (pix & 7)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(halo + q * 8 * 64), 16, 0, 0);
        }
        issue_w(0, cg * 64);                               // tap 0 of this group: global k = tap * Cin + cg * 64
        int buf = 0;
        for (int tap = 0; tap < 9; ++tap) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of stage `tap` (and, at tap 0, of the halo) have landed
            __builtin_amdgcn_s_barrier();                         // ... so have everyone's; and everyone is done reading the other buffer
            if (tap + 1 < 9) issue_w(buf ^ 1, (tap + 1) * p.Cin + cg * 64);
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const bf16_t* ws = wst + buf * (BN * G2_BK);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 a[4], bb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int h = (((16 * ty + wm * 4 + i + dy) >> UPS) - hy0) * HD + (((16 * tx + fr + dx) >> UPS) - hx0);
                    a[i] = *(const bf16x8*)(halo + h * 64 + (((kk * 4 + fq) ^ (h & 7)) << 3));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int r = wn * 64 + j * 16 + fr; bb[j] = *(const bf16x8*)(ws + r * G2_BK + (((kk * 4 + fq) ^ (r & 7)) << 3)); }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bb[j], acc[i][j], 0, 0, 0);
            }
            buf ^= 1;
        }
    }
    __syncthreads();
    // ---- epilogue (as conv3_halo_kernel): out = rnd(rnd(acc + bias) + R), 16-byte loads/stores: lane = (pixel rr of the patch row, 8 consecutive channels)
    const bf16_t* bias = (const bf16_t*)p.bias; const bf16_t* R = (const bf16_t*)p.R;
    float* strip = (float*)smem4 + wave * (16 * 68);
    const int ec = (lane & 7) * 8, en = n0 + wn * 64 + ec;
    float bv[8];
    {
        const uint4 u = bias ? *(const uint4*)(bias + en) : make_uint4(0, 0, 0, 0);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { bv[2 * e] = __uint_as_float(w[e] << 16); bv[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
    }
    float gs[8], gq[8];                         // GroupNorm stage-1 partials of this lane's 8 channels over its 8 pixels (p.gn_part)
#pragma unroll
    for (int e = 0; e < 8; ++e) { gs[e] = 0.f; gq[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 q = acc[i][j];
            float* d = strip + ((lane >> 4) * 4) * 68 + j * 16 + fr;
            d[0] = q[0]; d[68] = q[1]; d[136] = q[2]; d[204] = q[3];
        }
        __builtin_amdgcn_wave_barrier();
        const int oy = 16 * ty + wm * 4 + i;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int rr = pass * 8 + (lane >> 3);
            const long mr = ((long)b * p.Ho + oy) * p.Wo + 16 * tx + rr;
            const float4 s0 = *(const float4*)(strip + rr * 68 + ec), s1 = *(const float4*)(strip + rr * 68 + ec + 4);
            float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e] + bv[e]));
            if (R) {
                const uint4 u = *(const uint4*)(R + mr * p.ldr + en);
                const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] = v[2 * e] + __uint_as_float(w[e] << 16); v[2 * e + 1] = v[2 * e + 1] + __uint_as_float(w[e] & 0xffff0000u); }
            }
            uint4 o;
            o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16); o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
            o.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16); o.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
            *(uint4*)((bf16_t*)p.C + mr * p.ldc + en) = o;
            if (GNP) {                           // statistics of the values as STORED (bf16), as the separate pass would read them back
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float r_ = bf2f(f2bf(v[e])); gs[e] += r_; gq[e] += r_ * r_; }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (GNP) {
        // fold in a fixed order: 8 lanes of a wave share a channel chunk (lane & 7) -> xor 8, 16, 32; then the 4 waves of a channel half through LDS
        float* red = (float*)smem4 + 8 * (16 * 68);          // [8 waves][64 ch][2], behind the epilogue strips
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o_ = 8; o_ < 64; o_ <<= 1) { gs[e] += __shfl_xor(gs[e], o_, 64); gq[e] += __shfl_xor(gq[e], o_, 64); }
        }
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[(wave * 64 + lane * 8 + e) * 2] = gs[e]; red[(wave * 64 + lane * 8 + e) * 2 + 1] = gq[e]; }
        }
        __syncthreads();
        if (tid < 128) {                         // channel n0 + tid: waves (wm, wn) with wn = tid / 64, wm = 0..3 in order
            const int wn2 = tid >> 6, cc = tid & 63;
            float s_ = 0.f, q_ = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) { const float* r2 = red + ((w4 * 2 + wn2) * 64 + cc) * 2; s_ += r2[0]; q_ += r2[1]; }
            if (n0 + tid < p.N) {
                float* o2 = p.gn_part + ((long)b * (tw * th) + t2) * 2 * p.N;
                o2[n0 + tid] = s_; o2[p.N + n0 + tid] = q_;
            }
        }
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
    const Geo geo = { p.M, p.Cin, p.Ho, p.Wo, p.ups, p.lda, p.patch };
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
                ((float*)p.C)[zC + (long)m * p.ldc + n] = epi_value<float>(p, bias, scale, R, zR, m, (long)m, n, acc[i][j]);
        }
}

// fp32 on the matrix cores: the same contract as gemm_f32_kernel (K % 16 == 0, K-contiguous operands, every AMODE, the generic epilogue) on
// v_mfma_f32_16x16x4_f32 — exact fp32 products and sums (an fmaf chain per instruction), 16 x the VALU kernel's arithmetic rate per CU.
// 128 x 128 tile, 4 waves of 64 x 64 (4 x 4 accumulator fragments), K step 16 through a double-buffered LDS stage (row stride 20 floats);
// one 16-byte LDS read per lane feeds four MFMAs: lane (r, q) holds k = 16kt + 4q .. 4q+3 of row r and step s multiplies component s of both
// operands (a fixed permutation of K inside each 16-block, the same for A and W — the order decode_f32.hip uses).  Every output element sums the
// k-blocks in order, whatever M: a row's result does not depend on the batch it is computed in.
typedef __attribute__((ext_vector_type(4))) float f4_t;
#define F32_LD 20
template <int AMODE>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) float sA[2][128 * F32_LD];
    __shared__ __attribute__((aligned(16))) float sB[2][128 * F32_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q4 = lane >> 4, c16 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z, z0 = z / p.nb1, z1 = z - z0 * p.nb1;
    const float* A = (const float*)p.A + z0 * p.sA0 + z1 * p.sA1;
    const float* W = (const float*)p.W + z0 * p.sW0 + z1 * p.sW1;
    const long zC = z0 * p.sC0 + z1 * p.sC1, zR = z0 * p.sR0 + z1 * p.sR1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int lrow = tid >> 1, lkc = (tid & 1) * 8;               // loader: one row, two 16-byte chunks of the 16-wide k step
    const Geo geo = { p.M, p.Cin, p.Ho, p.Wo, p.ups, p.lda, p.patch };
    const ARow ar = make_arow<AMODE>(geo, m0 + lrow);
    const bool wok = (n0 + lrow) < p.N;
    const float* wp = W + (long)(n0 + lrow) * p.ldw + lkc;
    float4 ra[2], rb[2];
    const float4 zero4 = make_float4(0, 0, 0, 0);
    auto gload = [&](int kt) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const long o = a_off<AMODE>(geo, ar, kt * 16 + lkc + c * 4);
            ra[c] = zero4; rb[c] = zero4;
            if (o >= 0) ra[c] = *(const float4*)(A + o);
            if (wok) rb[c] = *(const float4*)(wp + (long)kt * 16 + c * 4);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int c = 0; c < 2; ++c) { *(float4*)&sA[buf][lrow * F32_LD + lkc + c * 4] = ra[c]; *(float4*)&sB[buf][lrow * F32_LD + lkc + c * 4] = rb[c]; }
    };
    f4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4_t){0.f, 0.f, 0.f, 0.f};
    const int nk = p.K / 16;
    gload(0); sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        f4_t xa[4], wa[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[j] = *(const f4_t*)&sA[buf][(wm * 64 + j * 16 + c16) * F32_LD + q4 * 4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wa[i] = *(const f4_t*)&sB[buf][(wn * 64 + i * 16 + c16) * F32_LD + q4 * 4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i][s], xa[j][s], acc[i][j], 0, 0, 0);
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }
    const float* bias = (const float*)p.bias; const float* scale = (const float*)p.scale; const float* R = (const float*)p.R;
    const bool vec = (p.ldc & 3) == 0 && ((zC & 3) == 0) && (((uintptr_t)p.C & 15) == 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + c16;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + q4 * 4;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (n + r < p.N) ? epi_value<float>(p, bias, scale, R, zR, m, (long)m, n + r, acc[i][j][r]) : 0.f;
            float* dst = (float*)p.C + zC + (long)m * p.ldc + n;
            if (vec && n + 3 < p.N) *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
            else { for (int r = 0; r < 4; ++r) if (n + r < p.N) dst[r] = v[r]; }
        }
    }
}

// host launchers -------------------------------------------------------------------------
// ONE predicate decides whether a 3x3 convolution takes the LDS-halo kernels (conv3_halo64_kernel / conv3_halo_kernel) — shape, alignment, the A/B switches,
// the device ordinal and the zero page of the LDS-DMA loader: car_conv3_halo64_ok (what engine_vq.hip asks before it lets the next GroupNorm skip its
// read-only pass) and car_launch_gemm both call it, so the two cannot disagree.
static void* g_zero_page[16] = {};
static void* zero_page_for(int dev) {
    if (dev < 0 || dev >= 16) return nullptr;
    if (!g_zero_page[dev]) { if (hipMalloc(&g_zero_page[dev], 256) == hipSuccess) (void)hipMemset(g_zero_page[dev], 0, 256); else { g_zero_page[dev] = nullptr; (void)hipGetLastError(); } }
    return g_zero_page[dev];
}
static bool conv3_halo_plan(int mode, int amode, const GemmP& p, int* dev_out, void** zero_out) {
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev_out) *dev_out = dev;
    if (!(mode == 1 && amode == AMODE_CONV3 && p.Cin % 128 == 0 && (p.Ho & 15) == 0 && (p.Wo & 15) == 0 && (p.ups == 0 || p.ups == 1) && p.K == 9 * p.Cin && p.ldw % 8 == 0 &&
          (uintptr_t)p.A % 16 == 0 && (uintptr_t)p.W % 16 == 0 && !p.swiglu && !p.out_f32 && p.act == ACT_NONE && (p.nb0 <= 1) && (p.nb1 <= 1) && p.alpha == 1.0f &&
          p.N % 128 == 0 && !p.scale && p.bias_mode != BIAS_M && p.ldc % 8 == 0 && (!p.R || p.ldr % 8 == 0) && !CAR_KNOB("CAR_GEMM_V1") && !CAR_KNOB("CAR_NO_HALO"))) return false;
    void* z = zero_page_for(dev);
    if (zero_out) *zero_out = z;
    return z != nullptr;
}
// the 64-channel-group kernel (the only one whose epilogue writes GroupNorm partials, GemmP::gn_part)
extern "C" int car_conv3_halo64_ok(int mode, const GemmP* pp) {
    return conv3_halo_plan(mode, AMODE_CONV3, *pp, nullptr, nullptr) && !CAR_KNOB("CAR_CONV_HALO128") && !CAR_KNOB("CAR_GN_UNFUSED");
}
// returns 0, or -1 when GemmP::gn_part is set on a call that cannot take conv3_halo64_kernel (the caller did not ask car_conv3_halo64_ok): nothing is launched
extern "C" int car_launch_gemm(int mode, int amode, const GemmP* pp, hipStream_t st) {
    GemmP p = *pp;
    if (p.nb0 <= 0) p.nb0 = 1;
    if (p.nb1 <= 0) p.nb1 = 1;
    if (mode == 1) {
        if (amode != AMODE_CONV3 || (p.Ho & 15) || (p.Wo & 15) || CAR_KNOB("CAR_CONV_LINEAR")) p.patch = 0;
        dim3 g((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.nb0 * p.nb1);
        // dynamic-LDS attributes are per device (a one-process multi-GPU host launches on several)
        static bool attr_set[16] = {}, attr3[16] = {}, attr4[16] = {};
        int dev = 0; void* zero = nullptr;
        // 3x3 conv with the input halo resident in LDS: stride 1, optional folded x2 upsample
        if (conv3_halo_plan(mode, amode, p, &dev, &zero)) {
            p.zero = zero;
            const dim3 g3((p.N + BN - 1) / BN, (unsigned)((long)p.M / 256));
            const bool halo128 = CAR_KNOB("CAR_CONV_HALO128") != nullptr;
            if (p.gn_part && (halo128 || CAR_KNOB("CAR_GN_UNFUSED"))) return -1;
            if (!halo128) {          // default: the 75-KB form, two workgroups per CU (A/B switch: CAR_CONV_HALO128=1 -> the 131-KB kernel)
                const size_t sh4 = (size_t)(CH64_HALO_PIX * 64 + 2 * BN * G2_BK) * 2;
                if (!attr4[dev]) {
                    (void)hipFuncSetAttribute((const void*)conv3_halo64_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh4);
                    (void)hipFuncSetAttribute((const void*)conv3_halo64_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh4);
                    (void)hipFuncSetAttribute((const void*)conv3_halo64_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh4);
                    (void)hipFuncSetAttribute((const void*)conv3_halo64_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh4);
                    attr4[dev] = true;
                }
                if (p.ups == 0) { if (p.gn_part) hipLaunchKernelGGL((conv3_halo64_kernel<0, 1>), g3, dim3(512), sh4, st, p); else hipLaunchKernelGGL((conv3_halo64_kernel<0, 0>), g3, dim3(512), sh4, st, p); }
                else { if (p.gn_part) hipLaunchKernelGGL((conv3_halo64_kernel<1, 1>), g3, dim3(512), sh4, st, p); else hipLaunchKernelGGL((conv3_halo64_kernel<1, 0>), g3, dim3(512), sh4, st, p); }
                return 0;
            }
            const size_t sh3 = (size_t)(CH_HALO_MAX * 128 + 3 * BN * G2_BK) * 2;
            if (!attr3[dev]) {
                (void)hipFuncSetAttribute((const void*)conv3_halo_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh3);
                (void)hipFuncSetAttribute((const void*)conv3_halo_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh3);
                attr3[dev] = true;
            }
            if (p.ups == 0) hipLaunchKernelGGL(conv3_halo_kernel<0>, g3, dim3(512), sh3, st, p);
            else hipLaunchKernelGGL(conv3_halo_kernel<1>, g3, dim3(512), sh3, st, p);
            return 0;
        }
        if (p.gn_part) return -1;
        const bool al16 = ((uintptr_t)p.A % 16 == 0) && ((uintptr_t)p.W % 16 == 0) && p.ldw % 8 == 0 && p.sW0 % 8 == 0 && p.sW1 % 8 == 0 && p.sA0 % 8 == 0 && p.sA1 % 8 == 0;
        const long tiles = (long)((p.N + BN - 1) / BN) * ((p.M + G2_BM - 1) / G2_BM) * p.nb0 * p.nb1;
        (void)hipGetDevice(&dev);
        const bool ok2 = dev >= 0 && dev < 16 && p.K % G2_BK == 0 && p.K >= 512 && al16 && tiles >= 512 && !CAR_KNOB("CAR_GEMM_V1") &&
                         (amode == AMODE_PLAIN ? p.lda % 8 == 0 : (amode == AMODE_CONV3 && p.Cin % G2_BK == 0));
        if (ok2) {
            const size_t sh = (size_t)G2_NS * G2_STAGE * 2;
            if (!attr_set[dev]) {
                (void)hipFuncSetAttribute((const void*)gemm_bf16_glds_kernel<AMODE_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
                (void)hipFuncSetAttribute((const void*)gemm_bf16_glds_kernel<AMODE_CONV3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
                attr_set[dev] = true;
            }
            if ((zero = zero_page_for(dev)) != nullptr) {
                p.zero = zero;
                dim3 g2((p.N + BN - 1) / BN, (p.M + G2_BM - 1) / G2_BM, p.nb0 * p.nb1);
                if (amode == AMODE_PLAIN) hipLaunchKernelGGL(gemm_bf16_glds_kernel<AMODE_PLAIN>, g2, dim3(512), sh, st, p);
                else hipLaunchKernelGGL(gemm_bf16_glds_kernel<AMODE_CONV3>, g2, dim3(512), sh, st, p);
                return 0;
            }
        }
        if (amode == AMODE_PLAIN) hipLaunchKernelGGL(gemm_bf16_kernel<AMODE_PLAIN>, g, dim3(256), 0, st, p);
        else if (amode == AMODE_CONV3S2) hipLaunchKernelGGL(gemm_bf16_kernel<AMODE_CONV3S2>, g, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(gemm_bf16_kernel<AMODE_CONV3>, g, dim3(256), 0, st, p);
    } else {
        if (p.gn_part) return -1;                 // GroupNorm partials are a bf16 conv3_halo64_kernel feature
        p.patch = 0;                              // the exact-mode kernel enumerates pixels linearly
        // exact mode: fp32 MFMA tiles (16 x the VALU rate per CU); the round-1 VALU kernel serves the shapes whose strides break the loader's 16-byte chunks
        const bool mfma_ok = p.K % 16 == 0 && p.ldw % 4 == 0 && ((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0 && p.sA0 % 4 == 0 && p.sA1 % 4 == 0 && p.sW0 % 4 == 0 && p.sW1 % 4 == 0 &&
                             (amode == AMODE_PLAIN ? p.lda % 4 == 0 : p.Cin % 4 == 0);
        if (mfma_ok) {
            dim3 g((p.N + 127) / 128, (p.M + 127) / 128, p.nb0 * p.nb1);
            if (amode == AMODE_PLAIN) hipLaunchKernelGGL(gemm_f32_mfma_kernel<AMODE_PLAIN>, g, dim3(256), 0, st, p);
            else if (amode == AMODE_CONV3S2) hipLaunchKernelGGL(gemm_f32_mfma_kernel<AMODE_CONV3S2>, g, dim3(256), 0, st, p);
            else hipLaunchKernelGGL(gemm_f32_mfma_kernel<AMODE_CONV3>, g, dim3(256), 0, st, p);
            return 0;
        }
        dim3 g((p.N + 63) / 64, (p.M + 63) / 64, p.nb0 * p.nb1);
        if (amode == AMODE_PLAIN) hipLaunchKernelGGL(gemm_f32_kernel<AMODE_PLAIN>, g, dim3(256), 0, st, p);
        else if (amode == AMODE_CONV3S2) hipLaunchKernelGGL(gemm_f32_kernel<AMODE_CONV3S2>, g, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(gemm_f32_kernel<AMODE_CONV3>, g, dim3(256), 0, st, p);
    }
    return 0;
}
