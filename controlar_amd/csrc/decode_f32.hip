// decode_f32.hip — the decode step of the EXACT mode (CAR_F32: fp32 weights / activations / KV cache, greedy tokens bit-identical to the
// reference's fp32 CPU path, `--precision none` of sample_t2i.py:197) on the matrix cores.
//
//   dec_gemm_f32   out = epi(X · W^T) for all rows of the batch in one pass over the fragment-packed fp32 weights, on
//                  v_mfma_f32_16x16x4_f32 (exact fp32 products and sums: the instruction is an fmaf chain over its 4 k values).  The five
//                  nn.Linear calls of the decode branch: gpt_t2i.py:264 wqkv, :289 wo, :216-217 w1 / w3 / w2, :470 output.
//                  One 16-byte operand load per lane feeds FOUR MFMAs: lane (r, q) holds k = 16kb + 4q .. 4q+3 of row r, and step s of the
//                  k-block multiplies component s of both operands — the k values {16kb + 4q + s : q = 0..3}.  A permutation of K inside a
//                  16-block, identical for W and X, i.e. a fixed summation order.
//                  CANONICAL ARITHMETIC of one output element (every kernel below produces exactly these bits): K is cut into 8 slices whose
//                  boundaries depend on K only (k-block floor(nkb·w/8)); slice w is one MFMA chain A_w started from zero in k-block order;
//                  the slices are folded as the balanced tree ((A0+A1)+(A2+A3)) + ((A4+A5)+(A6+A7)).  Nothing depends on M, on the tile
//                  shape or on how the slices are dealt to waves: the exact mode is batch-invariant
//                  (tests/test_parity_gpu.py::test_exact_mode_is_batch_invariant; experiments/f32_check compares all kernels bit for bit).
//   dec_gemm_f32   (round 4) a workgroup owns a (16·I n) x (16·J m) tile; its 8 waves take one slice each, operands straight from global
//                  memory into registers, fold through LDS.  Small batches and shapes the tiled kernel cannot take.
//   dec_gemm_f32t  (round 5) LDS-shared operand tiles: a workgroup of WN x WM x KG waves owns a (32·WN n) x (32·WM m) tile; the W and X
//                  fragments of a k-block are DMA'd into an LDS ring once per workgroup (global_load_lds, 1 KiB per wave instruction,
//                  counted vmcnt + raw s_barrier) and every wave reads its 2 x 2 fragments from there; KG wave groups take 8/KG
//                  consecutive slices each (a subtree of the fold).  A half to a third of dec_gemm_f32's L2 traffic per MFMA.
//   NX variants    (round 5) the RMSNorm in front of wqkv / w1|w3 / output applied ON THE FLY: the norm weight is folded into the columns of the packed
//                  weight image (W' = W·diag(w), pack_frag_f32 `colscale`), the GEMM multiplies the RAW residual rows, accumulates each row's sum of
//                  squares from the X fragments it streams anyway (canonical order: per K slice and lane quarter q one fmaf chain over the slice's
//                  k-blocks and the quarter's 4 elements; slice sum (p0+p1)+(p2+p3); the 8 slice sums folded by the same tree as the products) and scales
//                  the folded sums by rstd = rsqrt(ssq/K + eps) in the epilogue: y = rstd · Σ_k h_k (w_k W_nk) = Σ_k ((h_k rstd) w_k) W_nk up to fp32
//                  round-off (gpt_t2i.py:190-199 RMSNorm, :264 / :216-217 / :470 the consumers).  Two dependent launches per layer less.
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

// `colscale` (or null): dst = W · diag(colscale) — the RMSNorm weight of the norm in front of this linear folded into its columns (NX kernels below)
__global__ void pack_frag_f32_kernel(const float* src, float* dst, long N, long K, const float* colscale) {
    const long nkb = K >> 4, nch = (N >> 4) * nkb * 64;          // one thread per 16-byte lane slot
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x; const long st = (long)gridDim.x * blockDim.x;
    for (; i < nch; i += st) {
        const int l = (int)(i & 63); const long ck = i >> 6, rb = ck / nkb, kb = ck - rb * nkb;
        const long k0 = kb * 16 + (l >> 4) * 4;
        float4 v = *(const float4*)(src + (rb * 16 + (l & 15)) * K + k0);
        if (colscale) { const float4 c = *(const float4*)(colscale + k0); v.x = v.x * c.x; v.y = v.y * c.y; v.z = v.z * c.z; v.w = v.w * c.w; }
        *(float4*)(dst + i * 4) = v;
    }
}
extern "C" void car_launch_pack_frag_f32(const void* src, void* dst, long N, long K, const void* colscale, hipStream_t st) {
    long n = (N >> 4) * (K >> 4) * 64; long g = (n + 255) / 256; if (g > 16384) g = 16384; if (g < 1) g = 1;
    hipLaunchKernelGGL(pack_frag_f32_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)src, (float*)dst, N, K, (const float*)colscale);
}

#define F32_WAVES 8      // = the number of K slices of the canonical arithmetic

__device__ inline f4 add4(const f4 a, const f4 b) { return (f4){a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]}; }
__device__ inline f4 scale4(const f4 a, const float s) { return (f4){a[0] * s, a[1] * s, a[2] * s, a[3] * s}; }
// NX: slice sum of squares of a row from its four lane-quarter chains, (p0 + p1) + (p2 + p3), left in all four lanes (additions commute bit for bit)
__device__ inline float quarter_sum(float pq) { const float a = pq + __shfl_xor(pq, 16, 64); return a + __shfl_xor(a, 32, 64); }
__device__ inline float rstd_of(float ssq, int K, float eps) { return rsqrtf(ssq / (float)K + eps); }

// epilogue of one unit = (IW adjacent weight row-blocks starting at rbA, output row m): shared by both GEMM kernels, so a rotation or a SwiGLU product
// is the same instruction sequence whatever kernel / tile produced the sums
template <int EPI, int IW>
__device__ inline void f32_epilogue(const GemmFP& p, const f4 (&v)[IW], int rbA, int m, int q4) {
    if (EPI == FEPI_SWIGLU) {
        // row-blocks alternate w1 | w3 (engine_weights.hip car_load_tensor): v[0] = a, v[1] = c of hidden block rbA/2
        f4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = silu_f(v[0][r]) * v[IW - 1][r];
        const int hid = (rbA >> 1) * 16 + q4 * 4;
        *(f4*)(p.out + (long)m * p.ldo + hid) = o;
    } else {
#pragma unroll
        for (int ii = 0; ii < IW; ++ii) {
            const int n0 = (rbA + ii) * 16 + q4 * 4;
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

template <int I, int J, int EPI, int NX>
__global__ __launch_bounds__(F32_WAVES * 64) void dec_gemm_f32_kernel(GemmFP p) {
    car_kernarg_prefetch<(sizeof(GemmFP) + 63 + 48) / 64>();      // (round 6) every line of the argument block requested at once instead of one scalar-cache miss per first use
    extern __shared__ __attribute__((aligned(16))) float red_all[];   // [8 waves][I*J][64] f4 (+ NX: [8 waves][J][16] slice sums of squares)
    if (p.w_nt & 2) __builtin_amdgcn_s_setprio(3);
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
    // k-blocks in flight per wave.  Deeper (6 / 8 for the 2 x 2 tile, 136-209 VGPRs) was tried for chains that run in the shadow of another chain's attention, where
    // a load takes about three times as long: SLOWER, the registers cost more residency than the depth buys (384 sequences, 3 chains: 17.55 -> 18.29 / 19.21 ms per step)
    constexpr int DEPTH = (I + J) >= 8 ? 3 : ((I + J) >= 4 ? 4 : 6);
    f4 wr[DEPTH][I], xv[DEPTH][J];
    const f4 z4 = (f4){0.f, 0.f, 0.f, 0.f};
    auto load = [&](f4 (&w)[I], f4 (&x)[J], int kb) {
#pragma unroll
        for (int i = 0; i < I; ++i) { const f4* a = wp + ((long)i * nkb + kb) * 64; w[i] = (p.w_nt & 1) ? __builtin_nontemporal_load(a) : *a; }
#pragma unroll
        for (int j = 0; j < J; ++j) { x[j] = z4; if (xok[j]) x[j] = *(const f4*)(xr[j] + kb * 16); }
    };
    float pq[J];                                  // NX: this lane quarter's chain of squares of row c16 of m-block j over the wave's slice
#pragma unroll
    for (int j = 0; j < J; ++j) pq[j] = 0.f;
    auto compute = [&](const f4 (&w)[I], const f4 (&x)[J]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < I; ++i)
#pragma unroll
                for (int j = 0; j < J; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][s], x[j][s], acc[i][j], 0, 0, 0);
        if (NX) {
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) pq[j] = fmaf(x[j][e], x[j][e], pq[j]);
        }
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
    float* sqs = red_all + F32_WAVES * I * J * 256;
    if (NX) {
#pragma unroll
        for (int j = 0; j < J; ++j) { const float s = quarter_sum(pq[j]); if (q4 == 0) sqs[(wave * J + j) * 16 + c16] = s; }
    }
    __syncthreads();
    auto fold = [&](int i, int j) -> f4 {      // the canonical tree over the 8 slices
        f4 a[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) a[w] = rv[((w * I + i) * J + j) * 64 + lane];
        return add4(add4(add4(a[0], a[1]), add4(a[2], a[3])), add4(add4(a[4], a[5]), add4(a[6], a[7])));
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
        if (NX) {
            float a[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) a[w] = sqs[(w * J + j) * 16 + c16];
            const float rstd = rstd_of(((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])), p.K, p.neps);
#pragma unroll
            for (int ii = 0; ii < IW; ++ii) v[ii] = scale4(v[ii], rstd);
        }
        f32_epilogue<EPI, IW>(p, v, rb0 + ip * IW, m, q4);
    }
}


// =============================================================================================== dec_gemm_f32t: LDS-shared operand tiles
// Workgroup = KG K-groups x (WN x WM) waves, tile (32·WN n) x (32·WM m), every wave a 2 x 2 block of 16 x 16 MFMA tiles (16 accumulator registers).
// K-group g owns the 8/KG consecutive slices [g·8/KG, (g+1)·8/KG) of the canonical arithmetic and streams their k-blocks through its own LDS ring:
// a stage = SK k-blocks = SK·(2·WN + 2·WM) one-KiB fragment chunks (W chunks are contiguous in the packed weight image; an X chunk is gathered from the
// row-major activations by the DMA's per-lane addresses — the destination is lane-linear, i.e. the chunk lands in MFMA fragment order), dealt to the
// group's waves as CPW global_load_lds instructions each.  Stage t + NST - 1 is issued while stage t is consumed: a counted `s_waitcnt vmcnt` + a raw
// s_barrier per stage (cdna_hip_programming.md §5: __syncthreads would drain the queue).  The slice partials live on a small register stack that is
// merged exactly as the canonical tree prescribes (slice loop unrolled: all stack indices are compile-time); K-groups exchange their subtree sums
// through the (dead) ring and the tree is finished by the wave that runs the epilogue unit.
// Host guarantees: K % 128 == 0 (8 equal slices of len = K/128 k-blocks), len % SK == 0, N % (32·WN) == 0.
typedef __attribute__((address_space(1))) const void gptr32_t;
typedef __attribute__((address_space(3))) void lptr32_t;

template <int N> __device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int WN, int WM, int KG, int SK, int NST, int EPI, int NX>
__global__ __launch_bounds__(WN * WM * KG * 64, (WN * WM * KG == 4 ? 3 : (KG == 1 ? 2 : 4))) void dec_gemm_f32t_kernel(GemmFP p) {
    car_kernarg_prefetch<(sizeof(GemmFP) + 63 + 48) / 64>();      // (round 6) every line of the argument block requested at once instead of one scalar-cache miss per first use
    extern __shared__ __attribute__((aligned(16))) float ring_all[];
    if (p.w_nt & 2) __builtin_amdgcn_s_setprio(3);
    constexpr int NW = WN * WM, RBW = 2 * WN, MBW = 2 * WM, CHK = RBW + MBW, CH = CHK * SK, SLG = 8 / KG;
    constexpr int PW = RBW * SK / NW, PX = MBW * SK / NW, CPW = PW + PX;          // W and X pieces per wave per stage
    static_assert((RBW * SK) % NW == 0 && (MBW * SK) % NW == 0 && (KG == 1 || KG == 2 || KG == 4 || KG == 8) && NST >= 2, "dec_gemm_f32t configuration");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c16 = lane & 15, q4 = lane >> 4;
    const int g = wave / NW, wl = wave - g * NW, wn = wl / WM, wm = wl - wn * WM;
    const int nkb = p.K >> 4, Mb = (p.M + 15) >> 4, MT = (Mb + MBW - 1) / MBW;
    int t = blockIdx.x; const int total = gridDim.x;
    if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);          // XCD-aware: each XCD takes a contiguous run of tiles (M tiles of one weight row-block share an L2)
    const int nt = t / MT, mt = t - nt * MT;
    const int rb0 = nt * RBW, mb0 = mt * MBW;
    const int len = nkb >> 3;                    // k-blocks per slice
    const int glo = g * SLG * len;               // this group's first k-block
    const int nst = SLG * len / SK;              // stages per group (the same for every group: the barriers are workgroup-wide)
    float* ring = ring_all + g * (NST * CH * 256);

    // ---- this wave's DMA pieces of a stage.  Stage layout: [kk][W row-blocks 0..RBW-1 | X m-blocks 0..MBW-1] chunks of 1 KiB.
    // W piece i: linear index u = i·NW + wl over (kk, row-block); X piece i: u = i·NW + wl over (kk, m-block).
    const float* srcw[PW]; const float* srcx[PX]; int dstw[PW], dstx[PX];
#pragma unroll
    for (int i = 0; i < PW; ++i) { const int u = i * NW + wl, kk = u / RBW, r = u - kk * RBW;
        srcw[i] = p.W + ((long)(rb0 + r) * nkb + glo + kk) * 256 + lane * 4; dstw[i] = (kk * CHK + r) * 256; }
#pragma unroll
    for (int i = 0; i < PX; ++i) { const int u = i * NW + wl, kk = u / MBW, r = u - kk * MBW;
        int m = (mb0 + r) * 16 + c16; if (m >= p.M) m = p.M - 1;          // rows past M: any values, their outputs are never stored
        srcx[i] = p.X + (long)m * p.ldx + (glo + kk) * 16 + q4 * 4; dstx[i] = (kk * CHK + RBW + r) * 256; }
    auto issue = [&](int buf) {
        float* sb = ring + buf * (CH * 256);
#pragma unroll
        for (int i = 0; i < PW; ++i) { __builtin_amdgcn_global_load_lds((gptr32_t*)srcw[i], (lptr32_t*)(sb + dstw[i]), 16, 0, 0); srcw[i] += SK * 256; }
#pragma unroll
        for (int i = 0; i < PX; ++i) { __builtin_amdgcn_global_load_lds((gptr32_t*)srcx[i], (lptr32_t*)(sb + dstx[i]), 16, 0, 0); srcx[i] += SK * 16; }
    };
#ifdef CAR_STAMP
    long long st_t0 = __builtin_amdgcn_s_memtime(), st_w0 = (long long)wall_clock64(), st_sync = 0, st_iss = 0, st_tmp = 0;
#define ST_BEGIN() do { if (p.stamp) st_tmp = __builtin_amdgcn_s_memtime(); } while (0)
#define ST_END(acc) do { if (p.stamp) acc += __builtin_amdgcn_s_memtime() - st_tmp; } while (0)
#else
#define ST_BEGIN() do {} while (0)
#define ST_END(acc) do {} while (0)
#endif
    const f4 z4 = (f4){0.f, 0.f, 0.f, 0.f};
    int buf = 0, ts = 0;                         // ring slot and index of the stage consumed next
    const int spl = len / SK;                    // stages per slice
    const float* sbase = ring + lane * 4;
#pragma unroll
    for (int d = 0; d < NST - 1; ++d) if (d < nst) issue(d);
    // A slice partial: the four 16 x 16 accumulators of the wave tile (+ NX: the slice's sum of squares of this lane's row in each of the two m-blocks).
    struct Part { f4 a[2][2]; float sq[2]; };
    // one slice = one MFMA chain per accumulator, started from zero, over the slice's k-blocks in order
    auto slice = [&](Part& r) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) r.a[i][j] = z4;
        float pq[2] = {0.f, 0.f};
        for (int it = 0; it < spl; ++it, ++ts) {
            ST_BEGIN();
            // this wave's pieces of stage ts have landed (NST-2 later stages may still be in flight) ...
            if (ts + NST - 2 < nst) wait_vmcnt<(NST - 2) * CPW>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();        // ... so have everyone's, and everyone is done reading the slot consumed at ts-1
            ST_END(st_sync); ST_BEGIN();
            if (ts + NST - 1 < nst) issue(buf == 0 ? NST - 1 : buf - 1);
            ST_END(st_iss);
#pragma unroll
            for (int kk = 0; kk < SK; ++kk) {
                const float* sb = sbase + (buf * CH + kk * CHK) * 256;
                f4 w[2], x[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) w[i] = *(const f4*)(sb + (wn * 2 + i) * 256);
#pragma unroll
                for (int j = 0; j < 2; ++j) x[j] = *(const f4*)(sb + (RBW + wm * 2 + j) * 256);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) r.a[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][s], x[j][s], r.a[i][j], 0, 0, 0);
                if (NX) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) pq[j] = fmaf(x[j][e], x[j][e], pq[j]);
                }
            }
            buf = buf + 1 == NST ? 0 : buf + 1;
        }
        if (NX) { r.sq[0] = quarter_sum(pq[0]); r.sq[1] = quarter_sum(pq[1]); }
    };
    auto addto = [&](Part& a, const Part& b) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) a.a[i][j] = add4(a.a[i][j], b.a[i][j]);
        if (NX) { a.sq[0] = a.sq[0] + b.sq[0]; a.sq[1] = a.sq[1] + b.sq[1]; }
    };
    // the group's subtree of the canonical fold, written out (SLG = 8, 4, 2 or 1 consecutive slices)
    Part r0;
    slice(r0);
    if (SLG >= 2) { Part r1; slice(r1); addto(r0, r1); }                                    // A0 + A1
    if (SLG >= 4) { Part r1, r2; slice(r1); slice(r2); addto(r1, r2); addto(r0, r1); }      // + (A2 + A3)
    if (SLG >= 8) {
        Part r1, r2;
        slice(r1); slice(r2); addto(r1, r2);                                                // A4 + A5
        { Part r3; slice(r2); slice(r3); addto(r2, r3); }                                   // A6 + A7
        addto(r1, r2); addto(r0, r1);
    }
#ifdef CAR_STAMP
    const long long st_t1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- epilogue units: (this wave tile's row-block pair, m-block j); with KG > 1 the groups' subtree sums meet through the ring
    if (KG == 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int mb = mb0 + wm * 2 + j, m = mb * 16 + c16;
            if (mb >= Mb || m >= p.M) continue;
            f4 v[2] = {r0.a[0][j], r0.a[1][j]};
            if (NX) { const float rstd = rstd_of(r0.sq[j], p.K, p.neps); v[0] = scale4(v[0], rstd); v[1] = scale4(v[1], rstd); }
            f32_epilogue<EPI, 2>(p, v, rb0 + wn * 2, m, q4);
        }
    } else {
        __syncthreads();                         // every DMA has been waited for (vmcnt(0) at the tail): the ring is dead
        f4* rv = (f4*)ring_all; float* rq = ring_all + KG * NW * 4 * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) rv[(((g * NW + wl) * 2 + i) * 2 + j) * 64 + lane] = r0.a[i][j];
        if (NX) { rq[((g * NW + wl) * 2 + 0) * 64 + lane] = r0.sq[0]; rq[((g * NW + wl) * 2 + 1) * 64 + lane] = r0.sq[1]; }
        __syncthreads();
        // the 2 units of a wave tile go to K-groups 0 and 1
        if (g < 2) {
            const int j = g;
            const int mb = mb0 + wm * 2 + j, m = mb * 16 + c16;
            if (mb < Mb) {
                f4 v[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f4 a[8];
#pragma unroll
                    for (int gg = 0; gg < KG; ++gg) a[gg] = rv[(((gg * NW + wl) * 2 + i) * 2 + j) * 64 + lane];
                    if (KG == 2) v[i] = add4(a[0], a[1]);
                    else if (KG == 4) v[i] = add4(add4(a[0], a[1]), add4(a[2], a[3]));
                    else v[i] = add4(add4(add4(a[0], a[1]), add4(a[2], a[3])), add4(add4(a[4], a[5]), add4(a[6], a[7])));
                }
                if (NX) {
                    float a[8];
#pragma unroll
                    for (int gg = 0; gg < KG; ++gg) a[gg] = rq[((gg * NW + wl) * 2 + j) * 64 + lane];
                    const float ssq = KG == 2 ? a[0] + a[1] : (KG == 4 ? (a[0] + a[1]) + (a[2] + a[3]) : ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])));
                    const float rstd = rstd_of(ssq, p.K, p.neps);
                    v[0] = scale4(v[0], rstd); v[1] = scale4(v[1], rstd);
                }
                if (m < p.M) f32_epilogue<EPI, 2>(p, v, rb0 + wn * 2, m, q4);
            }
        }
    }
#ifdef CAR_STAMP
    if (p.stamp && tid == 0) {
        long long* o = p.stamp + (long)blockIdx.x * 16;
        o[0] = st_w0; o[1] = (long long)wall_clock64(); o[2] = st_t0; o[3] = st_t1; o[4] = __builtin_amdgcn_s_memtime(); o[5] = st_sync; o[6] = st_iss;
        o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o[8] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_ID, XCC_ID
    }
#endif
}

template <int WN, int WM, int KG, int SK, int NST>
static int launch_f32t(const GemmFP& p, int epi, hipStream_t st) {
    constexpr int NW = WN * WM, CH = (2 * WN + 2 * WM) * SK;
    const int Mb = (p.M + 15) / 16, MT = (Mb + 2 * WM - 1) / (2 * WM), NT = p.N / (32 * WN);
    const dim3 g(NT * MT), b(NW * KG * 64);
    size_t sh = (size_t)KG * NST * CH * 1024;
    const size_t red = (size_t)KG * NW * (4 * 1024 + 512); if (KG > 1 && red > sh) sh = red;
    static size_t attr[8][16] = {};
    int dev = 0; (void)hipGetDevice(&dev); if (dev < 0 || dev >= 16) return -1;
#define LG(E, X)                                                                                                                 \
    do {                                                                                                                         \
        if (sh > 48 * 1024 && sh > attr[E * 2 + X][dev]) {                                                                       \
            if (hipFuncSetAttribute((const void*)dec_gemm_f32t_kernel<WN, WM, KG, SK, NST, E, X>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) return -1; \
            attr[E * 2 + X][dev] = sh; }                                                                                         \
        hipLaunchKernelGGL((dec_gemm_f32t_kernel<WN, WM, KG, SK, NST, E, X>), g, b, sh, st, p);                                  \
    } while (0)
    if (p.normx) { if (epi == FEPI_PLAIN) LG(FEPI_PLAIN, 1); else if (epi == FEPI_SWIGLU) LG(FEPI_SWIGLU, 1); else if (epi == FEPI_QKV) LG(FEPI_QKV, 1); else return -1; }
    else if (epi == FEPI_PLAIN) LG(FEPI_PLAIN, 0); else if (epi == FEPI_RESID) LG(FEPI_RESID, 0); else if (epi == FEPI_SWIGLU) LG(FEPI_SWIGLU, 0); else LG(FEPI_QKV, 0);
#undef LG
    return 0;
}

// tiled configurations: cfg = 1000 + 100·WN + 10·WM + KG.  The product picks among 1221 / 1212 / 1214 (car_pick_gemm_f32_cfg); the others are compiled
// for the sweep of experiments/f32_check only (-DF32T_ALL_CONFIGS).
static int launch_f32t_cfg(const GemmFP& p, int epi, int cfg, hipStream_t st) {
    const int WN = (cfg % 1000) / 100;
    if (p.K % 128 || p.N % (32 * WN) || (p.ldx & 3) || p.M < 1) return -1;
    const int len = p.K / 128;
    const bool sk2 = len % 2 == 0;
    switch (cfg) {
        case 1221: return sk2 ? launch_f32t<2, 2, 1, 2, 3>(p, epi, st) : launch_f32t<2, 2, 1, 1, 4>(p, epi, st);
        case 1212: return launch_f32t<2, 1, 2, 1, 4>(p, epi, st);
        case 1214: return launch_f32t<2, 1, 4, 1, 3>(p, epi, st);
#ifdef F32T_ALL_CONFIGS
        case 1114: return launch_f32t<1, 1, 4, 1, 4>(p, epi, st);
        case 1118: return launch_f32t<1, 1, 8, 1, 3>(p, epi, st);
        case 1241: return sk2 ? launch_f32t<2, 4, 1, 2, 3>(p, epi, st) : -1;
        case 1421: return sk2 ? launch_f32t<4, 2, 1, 2, 3>(p, epi, st) : -1;
        case 1222: return launch_f32t<2, 2, 2, 1, 4>(p, epi, st);
        case 1124: return launch_f32t<1, 2, 4, 1, 3>(p, epi, st);
        case 1122: return launch_f32t<1, 2, 2, 1, 4>(p, epi, st);
#endif
        default: return -1;
    }
}

template <int I, int J>
static int launch_f32_ij(const GemmFP& p, int epi, hipStream_t st) {
    const int Mb = (p.M + 15) / 16, MT = (Mb + J - 1) / J, NT = p.N / (16 * I);
    const dim3 g(NT * MT), b(F32_WAVES * 64);
    const size_t sh = (size_t)F32_WAVES * I * J * 64 * 16 + (p.normx ? (size_t)F32_WAVES * J * 16 * 4 : 0);
    static size_t attr[8][16] = {};
    int dev = 0; (void)hipGetDevice(&dev); if (dev < 0 || dev >= 16) return -1;
#define LG(E, X)                                                                                                                 \
    do {                                                                                                                         \
        if (sh > 48 * 1024 && sh > attr[E * 2 + X][dev]) {                                                                       \
            if (hipFuncSetAttribute((const void*)dec_gemm_f32_kernel<I, J, E, X>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) return -1; \
            attr[E * 2 + X][dev] = sh; }                                                                                         \
        hipLaunchKernelGGL((dec_gemm_f32_kernel<I, J, E, X>), g, b, sh, st, p);                                            \
    } while (0)
    if (p.normx) { if (epi == FEPI_PLAIN) LG(FEPI_PLAIN, 1); else if (epi == FEPI_SWIGLU) LG(FEPI_SWIGLU, 1); else if (epi == FEPI_QKV) LG(FEPI_QKV, 1); else return -1; }
    else if (epi == FEPI_PLAIN) LG(FEPI_PLAIN, 0); else if (epi == FEPI_RESID) LG(FEPI_RESID, 0); else if (epi == FEPI_SWIGLU) LG(FEPI_SWIGLU, 0); else LG(FEPI_QKV, 0);
#undef LG
    return 0;
}

// cfg = I*10 + J.  Any (I, J) gives the same bits per output element (the K slicing is fixed); the choice is throughput only.
extern "C" int car_launch_dec_gemm_f32_cfg(const GemmFP* p, int epi, int cfg, hipStream_t st) {
    if (epi == FEPI_QKV && (p->dim % 64 || p->N != 3 * p->dim)) return -1;
    if (p->normx && epi == FEPI_RESID) return -1;
    if (cfg >= 1000) return launch_f32t_cfg(*p, epi, cfg, st);
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

// Tile choice from the MI355X sweeps of experiments/f32_check.hip (round 4: profiles/r04_f32_check_v1_8waves.txt; round 5 with the LDS-tiled kernel:
// profiles/r05_f32_check_*.txt, r05_f32t_stamp_*.txt).  Both kernels are bound by the fp32 matrix pipe — which on random data is POWER-limited: the
// stamps show the shader clock falling from 2.17 GHz with one 4-wave workgroup per CU to 1.6-1.75 GHz with three (111-114 TFLOP/s of matrix peak at
// that clock, not 155), so 90-113 TFLOP/s is 80-95 % of what the chip sustains.  Register kernel: the 32 x 32 tile (two 8-wave workgroups per CU) is
// within 5 % of its best everywhere; one m-block: 32 x 16.  The tiled kernel wins where there are many tiles per CU: w1|w3 and the logits from 128
// rows (43.3 vs 49.2 us and 82.7 vs 98.4 at 192 rows), wqkv and — with 4 K-groups on 64 x 32 tiles — wo / w2 from 320 rows (43.4 / 17.2 / 43.2 vs
// 46.5 / 18.5 / 45.2 us at 384).  Every choice yields the same bits.
// `chains`: how many chains the step runs (engine_generate.hip).  With several, a chain's linears run BESIDE another chain's attention, which holds 24 of a CU's 32
// wave slots (12-wave workgroups): two 4-wave workgroups of the tiled kernel fit into the remaining 8 where one 8-wave workgroup of the register kernel does, and they
// keep the matrix pipe fed from their LDS rings while HBM is saturated.  384 sequences, mean position, ms per step (profiles/r05_exact_probe_v8/v9.txt):
//   3 chains of 128 rows: register wqkv / wo / w2 17.52 | tiled wqkv 17.17 | tiled wqkv + wo / w2 16.65 (1114: 16.74, 1118: 16.84)
//   2 chains of 192 rows: tiled wqkv 17.68, tiled wqkv + wo / w2 18.42 (80-workgroup grids at 192 rows: too few) — wo / w2 stay on the register kernel
// ALONE (one chain) the register kernel is the faster one below 320 rows (wo at 192 rows: 11.3 vs 18.6 us).
extern "C" int car_pick_gemm_f32_cfg2(int M, int N, int K, int epi, int chains) {
    const int Mb = (M + 15) / 16;
    const int I = N % 32 == 0 ? 2 : 1;
    const int reg = I * 10 + (Mb >= 2 ? 2 : 1);
    if (K % 128 || N % 64) return reg;
    if (epi == FEPI_RESID) {
        { const char* ev = CAR_KNOB("CAR_F32_RESID_CFG"); if (ev && M >= 64) return atoi(ev); }
        if (M >= 320) return 1214;
        return chains >= 3 && M >= 96 ? 1212 : reg;
    }
    if (N >= 16384) return M >= 64 ? 1212 : reg;
    if (N >= 4096) return M >= 128 || (chains >= 2 && M >= 96) ? 1212 : reg;
    { const char* ev = CAR_KNOB("CAR_F32_QKV_TILED_FROM"); if (ev) return M >= atoi(ev) ? 1212 : reg; }
    return M >= 320 || (chains >= 2 && M >= 96) ? 1212 : reg;
}
extern "C" int car_pick_gemm_f32_cfg(int M, int N, int K, int epi) { return car_pick_gemm_f32_cfg2(M, N, K, epi, 1); }

// =============================================================================================== attention
// One workgroup (4 waves) per (head, sequence, split).  Split s covers the cache positions [s*AF_SPLIT, (s+1)*AF_SPLIT) ∩ [0, pos]: the
// boundaries are absolute, so the per-split online-softmax states and their fold are the same arithmetic for any batch.  A 16-lane group
// reads one 256-byte row with 16 bytes per lane (a wave instruction moves 1 KiB); 4 K rows + 4 V rows per lane are requested before the
// first is used.  Wave w takes the rows j0 + 4·UNR·(4i + w) + 4u + grp of its split: AF_SPLIT / 4 rows per wave per split.
#ifndef AF_UNR
#define AF_UNR 2
#endif
__global__ __launch_bounds__(256) void dec_attn_f32_kernel(AttnFP p) {
    car_kernarg_prefetch<(sizeof(AttnFP) + 63 + 48) / 64>();      // (round 6) every line of the argument block requested at once instead of one scalar-cache miss per first use
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

// NI (head, sequence) items per workgroup, 4 waves each, walk ALL splits of their sequence and fold them themselves: the per-split arithmetic and the fold over
// splits are the statements of the two kernels above (the same fixed order, hence the same bits — experiments/f32_check compares them), without the round trip
// of the partials through HBM and without the second launch.  Used when (head, sequence) pairs alone fill the chip.
// NI = 3 (768-thread workgroups) is the form for steps with several chains: the 32-wave limit of a CU admits two such workgroups = 24 waves, so EIGHT WAVE SLOTS
// PER CU STAY FREE for the other chain's linears, which otherwise cannot place a workgroup until the attention grid drains (profiles/r05_exact_b384_timeline.txt:
// w2 of the other chain "ran" 169 us beside a 181-us attention and finished with its tail).  Same bytes in flight per CU as 4-wave workgroups at 6 per CU.
template <int NI>
__global__ __launch_bounds__(256 * NI) void dec_attn_f32_fused_kernel(AttnFP p, int n_items) {
    car_kernarg_prefetch<(sizeof(AttnFP) + 63 + 48) / 64>();      // (round 6) every line of the argument block requested at once instead of one scalar-cache miss per first use
    extern __shared__ float occupancy_pad[];  // never touched: the launcher sizes it to cap the workgroups per CU (form 4)
    __shared__ float red[NI][4][4][66];       // per item: per wave, per row group: m, l, o[64]
    __shared__ float spl[NI][8][66];          // per item, per split: m, l, o[64]
    const int tid = threadIdx.x, it = tid >> 8, t256 = tid & 255, lane = tid & 63, wave = t256 >> 6, grp = lane >> 4, sub = lane & 15;
    const int item = blockIdx.x * NI + it;
    const bool live = item < n_items;         // a dead item (grid tail) loads nothing but keeps the barrier count
    const int b = live ? item / p.H : 0, h = live ? item - b * p.H : 0;
    const int pos = *p.pos;
    const int ns = pos / AF_SPLIT + 1;        // <= nsplit_max <= 8 (checked by the launcher)
    const long sbase = ((long)b * p.H + h) * p.S_max * 64;
    const float* kc = p.kc + sbase + sub * 4;
    const float* vc = p.vc + sbase + sub * 4;
    const f4 q = *(const f4*)(p.q + ((long)b * p.H + h) * 64 + sub * 4);
    const unsigned char* mk = p.mask ? p.mask + (long)b * p.T : nullptr;
    constexpr int UNR = AF_UNR;
    for (int split = 0; split < ns; ++split) {
        const int j0 = split * AF_SPLIT, j1 = live ? min(pos + 1, j0 + AF_SPLIT) : j0;
        float m = -INFINITY, l = 0.f;
        f4 o = (f4){0.f, 0.f, 0.f, 0.f};
        for (int base = j0 + wave * (4 * UNR); base < j1; base += 16 * UNR) {
            f4 kv[UNR], vv[UNR]; bool ok[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int j = base + u * 4 + grp;
                ok[u] = j < j1 && !(mk && j < p.T && j != pos && !mk[j]);
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
        if (sub == 0) { red[it][wave][grp][0] = m; red[it][wave][grp][1] = l; }
        *(f4*)&red[it][wave][grp][2 + sub * 4] = o;
        __syncthreads();
        if (t256 < 64) {
            float M = -INFINITY;
            for (int w = 0; w < 4; ++w) for (int g = 0; g < 4; ++g) M = fmaxf(M, red[it][w][g][0]);
            float L = 0.f, O = 0.f;
            for (int w = 0; w < 4; ++w) for (int g = 0; g < 4; ++g) {
                const float mm = red[it][w][g][0];
                if (mm > -INFINITY) { const float a = expf(mm - M); L += red[it][w][g][1] * a; O += red[it][w][g][2 + t256] * a; }
            }
            if (t256 == 0) { spl[it][split][0] = M; spl[it][split][1] = L; }
            spl[it][split][2 + t256] = O;
        }
        __syncthreads();                      // `red` is rewritten by the next split
    }
    if (t256 < 64 && live) {
        float M = -INFINITY;
        for (int s = 0; s < ns; ++s) M = fmaxf(M, spl[it][s][0]);
        float L = 0.f, O = 0.f;
        for (int s = 0; s < ns; ++s) {
            const float mm = spl[it][s][0];
            if (mm > -INFINITY) { const float a = expf(mm - M); L += spl[it][s][1] * a; O += spl[it][s][2 + t256] * a; }
        }
        p.out[(long)b * p.dim + h * 64 + t256] = O / L;
    }
}

// `fused`: 0 = split kernel + combine; 1 = one launch, 4-wave workgroups; 3 = one launch, 12-wave workgroups of three (head, sequence) items (leaves a
// quarter of every CU's wave slots to concurrent kernels: steps with several chains); -1 = choose 1 when (head, sequence) pairs alone give every CU
// >= 8 workgroups (2048 on MI355X), else 0 (3 x the workgroups for a handful of sequences).  Every form yields the same bits.
extern "C" void car_launch_dec_attn_f32_ex(const AttnFP* p, int b, int fused, hipStream_t st) {
    const int items = p->H * b;
    if (fused < 0) fused = items >= 2048;
    if (fused && p->nsplit_max <= 8) {
        if (fused == 3) hipLaunchKernelGGL(dec_attn_f32_fused_kernel<3>, dim3((items + 2) / 3), dim3(768), 0, st, *p, items);
        else if (fused == 4) {      // 16-wave workgroups, ONE per CU (56 KiB of padding on top of 25 KiB): half of every CU's wave slots stay free
            static bool attr_set[16] = {}; int dev = 0; (void)hipGetDevice(&dev);
            if (dev >= 0 && dev < 16 && !attr_set[dev]) { (void)hipFuncSetAttribute((const void*)dec_attn_f32_fused_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 57344); attr_set[dev] = true; }
            hipLaunchKernelGGL(dec_attn_f32_fused_kernel<4>, dim3((items + 3) / 4), dim3(1024), 57344, st, *p, items);
        }
        else hipLaunchKernelGGL(dec_attn_f32_fused_kernel<1>, dim3(items), dim3(256), 0, st, *p, items);
        return;
    }
    hipLaunchKernelGGL(dec_attn_f32_kernel, dim3(p->H, b, p->nsplit_max), dim3(256), 0, st, *p);
    hipLaunchKernelGGL(dec_attn_f32_combine_kernel, dim3(p->H, b), dim3(64), 0, st, *p);
}
extern "C" void car_launch_dec_attn_f32(const AttnFP* p, int b, hipStream_t st) { car_launch_dec_attn_f32_ex(p, b, -1, st); }
