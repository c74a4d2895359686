// decode_f32.hip — the decode step of the EXACT mode (CAR_F32: fp32 weights / activations / KV cache, greedy tokens bit-identical to the
// reference's fp32 CPU path, `--precision none` of sample_t2i.py:197) on the matrix cores.
//
//   dec_gemm_f32   out = epi(X · W^T) for all rows of the batch in one pass over the fragment-packed fp32 weights, on
//                  v_mfma_f32_16x16x4_f32 (exact fp32 products and sums: the instruction is an fmaf chain over its 4 k values).  The five
//                  nn.Linear calls of the decode branch: gpt_t2i.py:264 wqkv, :289 wo, :216-217 w1 / w3 / w2, :470 output.
//                  One 16-byte operand load per lane feeds FOUR MFMAs: lane (r, q) holds k = 16kb + 4q .. 4q+3 of row r, and step s of the
//                  k-block multiplies component s of both operands — the k values {16kb + 4q + s : q = 0..3}.  A permutation of K inside a
//                  16-block, identical for W and X, i.e. a fixed summation order.
//                  A workgroup owns a (16·I n) x (16·J m) tile over the whole K; its 8 waves split K in 8 fixed slices and fold through LDS
//                  in wave order.  The slice boundaries depend on K only, never on M: every output element is the same arithmetic whatever
//                  the batch (the exact mode is batch-invariant; tests/test_parity_gpu.py::test_exact_mode_is_batch_invariant).
//   dec_attn_f32   single-query attention over the fp32 cache, split-KV with boundaries fixed in absolute positions (batch-invariant),
//                  16-byte loads, 4 rows per wave instruction, 8 waves per SIMD resident.
//   pack_frag_f32  row-major fp32 [N][K] -> the fragment image (built once at car_finalize_weights).
#include "car_common.h"
#include "decode_f32_params.h"

typedef __attribute__((ext_vector_type(4))) float f4;

// No implicit fused multiply-adds in this file: `a*b - c*d` (the RoPE of the QKV epilogue) was contracted differently in two tile instantiations of the same
// template (fma(a, b, -(c*d)) in one, packed multiplies and a subtract in the other) — a last-bit difference between a sequence decoded alone (one m-block,
// 32 x 16 tiles) and in a large batch (32 x 32 tiles), caught by test_exact_mode_is_batch_invariant at the first position with a non-trivial rotation.
// Every product-sum below is now exactly what is written; the fmaf() calls are the only fused operations.
#pragma clang fp contract(off)

__global__ void pack_frag_f32_kernel(const float* src, float* dst, long N, long K) {
    const long nkb = K >> 4, nch = (N >> 4) * nkb * 64;          // one thread per 16-byte lane slot
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x; const long st = (long)gridDim.x * blockDim.x;
    for (; i < nch; i += st) {
        const int l = (int)(i & 63); const long ck = i >> 6, rb = ck / nkb, kb = ck - rb * nkb;
        *(float4*)(dst + i * 4) = *(const float4*)(src + (rb * 16 + (l & 15)) * K + kb * 16 + (l >> 4) * 4);
    }
}
extern "C" void car_launch_pack_frag_f32(const void* src, void* dst, long N, long K, hipStream_t st) {
    long n = (N >> 4) * (K >> 4) * 64; long g = (n + 255) / 256; if (g > 16384) g = 16384; if (g < 1) g = 1;
    hipLaunchKernelGGL(pack_frag_f32_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)src, (float*)dst, N, K);
}

#ifndef F32_WAVES
#define F32_WAVES 8
#endif

template <int I, int J, int EPI>
__global__ __launch_bounds__(F32_WAVES * 64) void dec_gemm_f32_kernel(GemmFP p) {
    extern __shared__ __attribute__((aligned(16))) float red_all[];   // [8 waves][I*J][64] f4
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q4 = lane >> 4, c16 = lane & 15;
    const int nkb = p.K >> 4, Mb = (p.M + 15) >> 4, MT = (Mb + J - 1) / J;
    // XCD-aware tile order: workgroup ids are dealt round-robin to the 8 XCDs; give each XCD a contiguous run of tiles so that the M tiles
    // sharing a weight row-block hit the same L2
    int t = blockIdx.x; const int total = gridDim.x;
    if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);
    const int nt = t / MT, mt = t - nt * MT;
    const int rb0 = nt * I, mb0 = mt * J;
    const int kb_lo = (int)((long)nkb * wave / F32_WAVES), kb_hi = (int)((long)nkb * (wave + 1) / F32_WAVES);   // functions of K only
    const f4* wp = (const f4*)p.W + (long)rb0 * nkb * 64 + lane;
    const float* xr[J]; bool xok[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { const int m = (mb0 + j) * 16 + c16; xok[j] = m < p.M; xr[j] = p.X + (long)(xok[j] ? m : 0) * p.ldx + q4 * 4; }

    f4 acc[I][J];
#pragma unroll
    for (int i = 0; i < I; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    constexpr int DEPTH = (I + J) >= 8 ? 3 : ((I + J) >= 4 ? 4 : 6);
    f4 wr[DEPTH][I], xv[DEPTH][J];
    const f4 z4 = (f4){0.f, 0.f, 0.f, 0.f};
    auto load = [&](f4 (&w)[I], f4 (&x)[J], int kb) {
#pragma unroll
        for (int i = 0; i < I; ++i) { const f4* a = wp + ((long)i * nkb + kb) * 64; w[i] = p.w_nt ? __builtin_nontemporal_load(a) : *a; }
#pragma unroll
        for (int j = 0; j < J; ++j) { x[j] = z4; if (xok[j]) x[j] = *(const f4*)(xr[j] + kb * 16); }
    };
    auto compute = [&](const f4 (&w)[I], const f4 (&x)[J]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < I; ++i)
#pragma unroll
                for (int j = 0; j < J; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][s], x[j][s], acc[i][j], 0, 0, 0);
    };
    const int nkw = kb_hi - kb_lo;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) if (d < nkw) load(wr[d], xv[d], kb_lo + d);
    for (int base = 0; base < nkw; base += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (base + d < nkw) {                                     // wave-uniform
                compute(wr[d], xv[d]);
                if (base + d + DEPTH < nkw) load(wr[d], xv[d], kb_lo + base + d + DEPTH);
            }
        }
    }
    // ---- fold the 8 K-slices in wave order through LDS
    f4* rv = (f4*)red_all;
#pragma unroll
    for (int i = 0; i < I; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) rv[((wave * I + i) * J + j) * 64 + lane] = acc[i][j];
    __syncthreads();
    auto fold = [&](int i, int j) -> f4 {
        f4 s = rv[((0 * I + i) * J + j) * 64 + lane];
#pragma unroll
        for (int w = 1; w < F32_WAVES; ++w) { const f4 v = rv[((w * I + i) * J + j) * 64 + lane]; s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
        return s;
    };
    // epilogue units: (pair of adjacent row-blocks, m-block) — the SwiGLU (a, c) pair must meet in one lane
    constexpr int IP = I >= 2 ? I / 2 : 1, IW = I >= 2 ? 2 : 1;
    for (int u = wave; u < IP * J; u += F32_WAVES) {
        const int ip = u / J, j = u - ip * J;
        const int m = (mb0 + j) * 16 + c16;
        if ((mb0 + j) >= Mb) continue;
        f4 v[IW];
#pragma unroll
        for (int ii = 0; ii < IW; ++ii) v[ii] = fold(ip * IW + ii, j);
        if (m >= p.M) continue;
        if (EPI == FEPI_SWIGLU) {
            // row-blocks alternate w1 | w3 (engine_weights.hip car_load_tensor): v[0] = a, v[1] = c of hidden block (rb0/2 + ip)
            f4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = silu_f(v[0][r]) * v[IW - 1][r];
            const int hid = ((rb0 >> 1) + ip) * 16 + q4 * 4;
            *(f4*)(p.out + (long)m * p.ldo + hid) = o;
        } else {
#pragma unroll
            for (int ii = 0; ii < IW; ++ii) {
                const int n0 = (rb0 + ip * IW + ii) * 16 + q4 * 4;
                const f4 a = v[ii];
                if (EPI == FEPI_PLAIN) {
                    *(f4*)(p.out + (long)m * p.ldo + n0) = a;
                } else if (EPI == FEPI_RESID) {
                    const f4 r = *(const f4*)(p.R + (long)m * p.ldo + n0);
                    *(f4*)(p.out + (long)m * p.ldo + n0) = (f4){r[0] + a[0], r[1] + a[1], r[2] + a[2], r[3] + a[3]};
                } else {   // FEPI_QKV
                    const int pos = *p.pos;
                    const int sec = n0 / p.dim, within = n0 - sec * p.dim, hh = within >> 6, d0 = within & 63;
                    const long row = ((long)m * p.H + hh);
                    if (sec == 2) {
                        *(f4*)(p.vc + (row * p.S_max + pos) * 64 + d0) = a;
                    } else {
                        const float4 cs = *(const float4*)(p.rope + ((long)pos * 32 + (d0 >> 1)) * 2);   // (cos, sin) of pairs d0/2, d0/2+1
                        const f4 r = (f4){a[0] * cs.x - a[1] * cs.y, a[1] * cs.x + a[0] * cs.y, a[2] * cs.z - a[3] * cs.w, a[3] * cs.z + a[2] * cs.w};
                        if (sec == 0) *(f4*)(p.qout + row * 64 + d0) = (f4){r[0] * 0.125f, r[1] * 0.125f, r[2] * 0.125f, r[3] * 0.125f};   // head_dim^-0.5 = 1/8 exactly
                        else *(f4*)(p.kc + (row * p.S_max + pos) * 64 + d0) = r;
                    }
                }
            }
        }
    }
}

template <int I, int J>
static int launch_f32_ij(const GemmFP& p, int epi, hipStream_t st) {
    const int Mb = (p.M + 15) / 16, MT = (Mb + J - 1) / J, NT = p.N / (16 * I);
    const dim3 g(NT * MT), b(F32_WAVES * 64);
    const size_t sh = (size_t)F32_WAVES * I * J * 64 * 16;
    static size_t attr[4][16] = {};
    int dev = 0; (void)hipGetDevice(&dev); if (dev < 0 || dev >= 16) return -1;
#define LG(E)                                                                                                                    \
    do {                                                                                                                         \
        if (sh > 48 * 1024 && sh > attr[E][dev]) {                                                                               \
            if (hipFuncSetAttribute((const void*)dec_gemm_f32_kernel<I, J, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) return -1; \
            attr[E][dev] = sh; }                                                                                                 \
        hipLaunchKernelGGL((dec_gemm_f32_kernel<I, J, E>), g, b, sh, st, p);                                                     \
    } while (0)
    if (epi == FEPI_PLAIN) LG(FEPI_PLAIN); else if (epi == FEPI_RESID) LG(FEPI_RESID); else if (epi == FEPI_SWIGLU) LG(FEPI_SWIGLU); else LG(FEPI_QKV);
#undef LG
    return 0;
}

// cfg = I*10 + J.  Any (I, J) gives the same bits per output element (the K slicing is fixed); the choice is throughput only.
extern "C" int car_launch_dec_gemm_f32_cfg(const GemmFP* p, int epi, int cfg, hipStream_t st) {
    const int I = cfg / 10;
    if (epi == FEPI_SWIGLU && I < 2) return -1;
    if (p->N % (16 * I) || p->K % 16 || (p->ldx & 3)) return -1;
    if (epi == FEPI_QKV && (p->dim % 64 || p->N != 3 * p->dim)) return -1;
    switch (cfg) {
#define CASE(I, J) case I * 10 + J: return launch_f32_ij<I, J>(*p, epi, st);
        CASE(1, 1) CASE(1, 2) CASE(1, 4) CASE(2, 1) CASE(2, 2) CASE(2, 4) CASE(4, 1) CASE(4, 2) CASE(4, 4)
#undef CASE
        default: return -1;
    }
}

// Tile choice from the MI355X sweep of experiments/f32_check.hip (profiles/r04_f32_check_v1_8waves.txt; XL shapes, M = 16 .. 768): the kernel is
// bound by the fp32 matrix pipe (one 16x16x4 MFMA per 32 cycles per SIMD; 80-92 TFLOP/s reached of the 155 peak), and the 32 x 32 tile — 100 VGPRs
// and 32 KiB of LDS, so TWO 8-wave workgroups share a CU and one's prologue / fold / epilogue sit under the other's MFMAs — is within 5 % of the best
// configuration for every shape and batch from 32 rows up (64 x 64 tiles: one workgroup per CU, 0.45-0.95 x).  One m-block: 32 x 16.
extern "C" int car_pick_gemm_f32_cfg(int M, int N, int K, int epi) {
    (void)K; (void)epi;
    const int Mb = (M + 15) / 16;
    const int I = N % 32 == 0 ? 2 : 1;
    return I * 10 + (Mb >= 2 ? 2 : 1);
}

// =============================================================================================== attention
// One workgroup (4 waves) per (head, sequence, split).  Split s covers the cache positions [s*AF_SPLIT, (s+1)*AF_SPLIT) ∩ [0, pos]: the
// boundaries are absolute, so the per-split online-softmax states and their fold are the same arithmetic for any batch.  A 16-lane group
// reads one 256-byte row with 16 bytes per lane (a wave instruction moves 1 KiB); 4 K rows + 4 V rows per lane are requested before the
// first is used.  Wave w takes the rows j0 + 4·UNR·(4i + w) + 4u + grp of its split: AF_SPLIT / 4 rows per wave per split.
__global__ __launch_bounds__(256) void dec_attn_f32_kernel(AttnFP p) {
    __shared__ float red[4][4][66];           // per wave, per row group: m, l, o[64]
    const int h = blockIdx.x, b = blockIdx.y, split = blockIdx.z;
    const int pos = *p.pos;
    const int j0 = split * AF_SPLIT;
    if (j0 > pos) return;                                       // empty split (uniform per workgroup): the combine never reads it
    const int j1 = min(pos + 1, j0 + AF_SPLIT);                 // the new token's row is in the cache (FEPI_QKV wrote it)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, sub = lane & 15;
    const long sbase = ((long)b * p.H + h) * p.S_max * 64;
    const float* kc = p.kc + sbase + sub * 4;
    const float* vc = p.vc + sbase + sub * 4;
    const f4 q = *(const f4*)(p.q + ((long)b * p.H + h) * 64 + sub * 4);
    const unsigned char* mk = p.mask ? p.mask + (long)b * p.T : nullptr;

    float m = -INFINITY, l = 0.f;
    f4 o = (f4){0.f, 0.f, 0.f, 0.f};
    // rows in flight per lane per stream: 2 K + 2 V (4 KiB per wave).  Fewer registers beat deeper unrolling: 64 VGPRs keep 8 waves per SIMD resident
    // (UNR 4: 100 VGPRs, 4 waves per SIMD, 374 us per layer at 384 sequences, position 631; UNR 2: 336 us = 6.27 TB/s, the chip's copy rate; UNR 1: 334;
    // UNR 8 spills: 1270) — profiles/r04_f32_attention_split_sweep.txt
#ifndef AF_UNR
#define AF_UNR 2
#endif
    constexpr int UNR = AF_UNR;
    for (int base = j0 + wave * (4 * UNR); base < j1; base += 16 * UNR) {
        f4 kv[UNR], vv[UNR]; bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int j = base + u * 4 + grp;
            ok[u] = j < j1 && !(mk && j < p.T && j != pos && !mk[j]);          // the diagonal is always allowed (generate.py:190-193)
            if (ok[u]) { kv[u] = __builtin_nontemporal_load((const f4*)(kc + (long)j * 64)); vv[u] = __builtin_nontemporal_load((const f4*)(vc + (long)j * 64)); }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float s = 0.f;
            if (ok[u]) { s = fmaf(q[0], kv[u][0], s); s = fmaf(q[1], kv[u][1], s); s = fmaf(q[2], kv[u][2], s); s = fmaf(q[3], kv[u][3], s); }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) s += __shfl_xor(s, off, 64);
            if (ok[u]) {
                const float mn = fmaxf(m, s), a = expf(m - mn), w = expf(s - mn);
                l = l * a + w;
                o[0] = fmaf(o[0], a, w * vv[u][0]); o[1] = fmaf(o[1], a, w * vv[u][1]); o[2] = fmaf(o[2], a, w * vv[u][2]); o[3] = fmaf(o[3], a, w * vv[u][3]);
                m = mn;
            }
        }
    }
    // ---- merge the 16 (wave, row group) states of this split in fixed order
    if (sub == 0) { red[wave][grp][0] = m; red[wave][grp][1] = l; }
    *(f4*)&red[wave][grp][2 + sub * 4] = o;
    __syncthreads();
    if (tid < 64) {
        float M = -INFINITY;
        for (int w = 0; w < 4; ++w) for (int g = 0; g < 4; ++g) M = fmaxf(M, red[w][g][0]);
        float L = 0.f, O = 0.f;
        for (int w = 0; w < 4; ++w) for (int g = 0; g < 4; ++g) {
            const float mm = red[w][g][0];
            if (mm > -INFINITY) { const float a = expf(mm - M); L += red[w][g][1] * a; O += red[w][g][2 + tid] * a; }
        }
        float* pt = p.part + (((long)b * p.H + h) * p.nsplit_max + split) * 66;
        if (tid == 0) { pt[0] = M; pt[1] = L; }
        pt[2 + tid] = O;
    }
}

// folds the non-empty splits (those starting at or before *pos) in position order
__global__ __launch_bounds__(64) void dec_attn_f32_combine_kernel(AttnFP p) {
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const int ns = *p.pos / AF_SPLIT + 1;
    const float* pt = p.part + ((long)b * p.H + h) * p.nsplit_max * 66;
    float M = -INFINITY;
    for (int s = 0; s < ns; ++s) M = fmaxf(M, pt[s * 66]);
    float L = 0.f, O = 0.f;
    for (int s = 0; s < ns; ++s) {
        const float mm = pt[s * 66];
        if (mm > -INFINITY) { const float a = expf(mm - M); L += pt[s * 66 + 1] * a; O += pt[s * 66 + 2 + d] * a; }
    }
    p.out[(long)b * p.dim + h * 64 + d] = O / L;
}

extern "C" void car_launch_dec_attn_f32(const AttnFP* p, int b, hipStream_t st) {
    hipLaunchKernelGGL(dec_attn_f32_kernel, dim3(p->H, b, p->nsplit_max), dim3(256), 0, st, *p);
    hipLaunchKernelGGL(dec_attn_f32_combine_kernel, dim3(p->H, b), dim3(64), 0, st, *p);
}
