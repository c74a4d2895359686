// experiments/decode4_tiled_gemm.hip — LDS-tiled GEMM for the decode linears of a LARGE chain (384 .. 768 rows), fragment-packed operands,
// the epilogues of dec_gemm.  EXPERIMENT, NOT PRODUCT (not built into libcontrolar_hip.so); written after the round-2 GPU budget was
// spent — first run belongs to round 3 (experiments/t_check.hip checks it against dec_gemm and times it).
//
// Why: the decode step at 768 sequences is attention (12.0-12.4 ms) + linears (3.2 ms paired, 4.2 ms as one chain), and nothing can be
// hidden under the attention.  The linears are 1.15 TFLOP per step = 0.5 ms at the MFMA peak; dec_gemm reaches ~15 % of it
// at these sizes: its workgroup tile is what ONE wave's registers hold (<= 64 x 64) because the waves split K, so every operand byte is
// re-fetched from L2 by M/64 resp. N/64 workgroups and every workgroup ends in an LDS fold.
//
// Here a workgroup owns a (32·I) x (32·J) output tile; 4 waves sit in a 2 x 2 grid (wave tile 16·I x 16·J) and SHARE the operands
// through LDS: per iteration (2 k-blocks of 32) the 2I weight chunks and 2J X chunks (1 KiB each, already in MFMA fragment order in
// HBM) are fetched once per workgroup, parked in LDS (two stages, one barrier per iteration) and read back with ds_read_b128 —
// lane l reads bytes [16l, 16l+16) of a chunk: conflict-free without a swizzle.  KG = 2 adds a second group of 4 waves on the other half
// of every iteration's k-blocks (long K: w2), folded once through LDS in fixed order.  Operand traffic per output = K·2·(1/32I + 1/32J)
// bytes: 64 x 128 tiles move 0.6x, 128 x 128 tiles 0.5x of what dec_gemm's 64 x 64 tiles move, with no per-workgroup fold at KG = 1.
//
// Accumulation order: one fp32 chain per K-group over ascending k (KG = 2: group 0 + group 1) — differs from dec_gemm's WAVES partial
// sums, so results agree to fp32 round-off before the bf16 rounding points, not bit for bit (as experiments/decode3_ws_gemm.hip).
#include "../controlar_amd/csrc/car_common.h"

typedef __attribute__((ext_vector_type(4))) unsigned t_u32x4;
typedef __attribute__((ext_vector_type(2))) float t_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 t_bf16x2;

__device__ inline unsigned t_pack_bf16x2(float a, float b) {
    const t_f32x2 v = {a, b};
    const t_bf16x2 r = __builtin_convertvector(v, t_bf16x2);
    return *(const unsigned*)&r;
}

#ifndef CAR_GEMMDP_DEFINED
#error "include controlar_amd/csrc/decode2.hip first: GemmDP and the EPI_* constants come from there"
#endif

template <int I, int J, int KG, int EPI>
__global__ __launch_bounds__(256 * KG) void dec_gemm_t_kernel(GemmDP p) {
    constexpr int KS = 2;                         // k-blocks per iteration per K-group
    constexpr int NW = 2 * I, NX = 2 * J;         // weight row-blocks / m-blocks of the workgroup tile
    constexpr int ROWC = NW + NX;                 // chunks per k-block
    constexpr int CH = ROWC * KS;                 // chunks per K-group per stage
    constexpr int LPW = CH / 4;                   // chunks fetched per wave per iteration (4 waves per K-group)
    static_assert(CH % 4 == 0, "chunks must divide among the 4 waves of a K-group");
    extern __shared__ __attribute__((aligned(16))) t_u32x4 tl[];      // [2 stages][KG][CH][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = wave >> 2, ws = wave & 3, wn = ws & 1, wm = ws >> 1;
    if (p.w_nt & 2) __builtin_amdgcn_s_setprio(3);
    const int nkb = p.K >> 5, Mb = (p.M + 15) >> 4;
    const int nit = (nkb + KS * KG - 1) / (KS * KG);
    const int rb0 = blockIdx.x * NW, mb0 = blockIdx.y * NX;
    const t_u32x4 zw = (t_u32x4){0u, 0u, 0u, 0u};
    const t_u32x4* Wp = (const t_u32x4*)p.W + lane;
    const t_u32x4* Xp = (const t_u32x4*)p.X + lane;

    // this wave's chunks of a stage: c = ws, ws + 4, ...; chunk c = (ks, r): r < NW a weight row-block, else an X m-block
    long coff[LPW]; bool cval[LPW]; int cks[LPW];
#pragma unroll
    for (int u = 0; u < LPW; ++u) {
        const int c = ws + 4 * u, ks = c / ROWC, r = c - ks * ROWC;
        cks[u] = ks;
        if (r < NW) { coff[u] = (long)(rb0 + r) * nkb * 64; cval[u] = true; }
        else { const int mb = mb0 + (r - NW); cval[u] = mb < Mb; coff[u] = (long)mb * nkb * 64; }
    }
    auto kb_of = [&](int it, int ks) { return (it * KG + kg) * KS + ks; };        // iteration `it` covers KS·KG consecutive k-blocks
    auto fetch = [&](t_u32x4 (&r)[LPW], int it) {
#pragma unroll
        for (int u = 0; u < LPW; ++u) {
            const int c = ws + 4 * u, rr = c % ROWC;
            const int kb = kb_of(it, cks[u]);
            r[u] = zw;
            if (cval[u] && kb < nkb) r[u] = rr < NW ? __builtin_nontemporal_load(Wp + coff[u] + (long)kb * 64) : Xp[coff[u] + (long)kb * 64];
        }
    };
    auto park = [&](int stage, const t_u32x4 (&r)[LPW]) {
#pragma unroll
        for (int u = 0; u < LPW; ++u) tl[((stage * KG + kg) * CH + ws + 4 * u) * 64 + lane] = r[u];
    };
    f32x4 acc[I][J];
#pragma unroll
    for (int i = 0; i < I; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int stage) {
        const t_u32x4* base = tl + (size_t)((stage * KG + kg) * CH) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            t_u32x4 a[I], x[J];
#pragma unroll
            for (int i = 0; i < I; ++i) a[i] = base[(ks * ROWC + wn * I + i) * 64];
#pragma unroll
            for (int j = 0; j < J; ++j) x[j] = base[(ks * ROWC + NW + wm * J + j) * 64];
#pragma unroll
            for (int i = 0; i < I; ++i)
#pragma unroll
                for (int j = 0; j < J; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&a[i], *(const bf16x8*)&x[j], acc[i][j], 0, 0, 0);
        }
    };
    // LDS: two stages, one barrier per iteration (stage s is rewritten at it + 2, after the barrier of it + 1 which every wave passes only
    // once it has finished reading stage s at `it`).  Registers: THREE fetches in flight per wave (iterations it+1 .. it+3) — an iteration
    // is ~0.15 us of MFMAs against > 1 us of load latency, and a tile's K extent cannot sit in registers as dec_gemm's K slices do.
    t_u32x4 r0[LPW], r1[LPW], r2[LPW];
    fetch(r0, 0);
    if (1 < nit) fetch(r1, 1);
    if (2 < nit) fetch(r2, 2);
    for (int it = 0; it < nit; it += 3) {
        park(it & 1, r0);
        __syncthreads();
        if (it + 3 < nit) fetch(r0, it + 3);
        compute(it & 1);
        if (it + 1 >= nit) break;
        park((it + 1) & 1, r1);
        __syncthreads();
        if (it + 4 < nit) fetch(r1, it + 4);
        compute((it + 1) & 1);
        if (it + 2 >= nit) break;
        park((it + 2) & 1, r2);
        __syncthreads();
        if (it + 5 < nit) fetch(r2, it + 5);
        compute((it + 2) & 1);
    }
    if (KG == 2) {          // fold: group 1 parks its accumulators, group 0 adds them (fixed order) and owns the epilogue
        __syncthreads();    // all reads of the last stage are done: the staging area is free
        f32x4* fold = (f32x4*)tl;
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < I; ++i)
#pragma unroll
                for (int j = 0; j < J; ++j) fold[((ws * I + i) * J + j) * 64 + lane] = acc[i][j];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int i = 0; i < I; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) { const f32x4 v = fold[((ws * I + i) * J + j) * 64 + lane]; acc[i][j][0] += v[0]; acc[i][j][1] += v[1]; acc[i][j][2] += v[2]; acc[i][j][3] += v[3]; }
    }

    // ---- epilogue: the units of dec_gemm_kernel (pair of adjacent row-blocks x m-block); arithmetic and reference lines as decode2.hip
    constexpr int IP = I / 2;
    const int q4 = lane >> 4, c16 = lane & 15;
    const int rbw = rb0 + wn * I;                 // first row-block of this wave (even: the w1 | w3 pairs stay together)
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int mb = mb0 + wm * J + j;
        if (mb >= Mb) continue;
        const int m = mb * 16 + c16;
        if (m >= p.M) continue;
#pragma unroll
        for (int ip = 0; ip < IP; ++ip) {
            const f32x4 v0 = acc[ip * 2][j], v1 = acc[ip * 2 + 1][j];
            if (EPI == EPI_SWIGLU) {
                float s[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = bf2f(f2bf(v0[r])), g = bf2f(f2bf(v1[r]));
                    s[r] = bf2f(f2bf(silu_f(a))) * g;
                }
                const int hid = ((rbw >> 1) + ip) * 16 + q4 * 4;
                const int nkb2 = p.N >> 6;
                const long off = ((((long)(m >> 4) * nkb2 + (hid >> 5)) * 64 + ((hid & 31) >> 3) * 16 + (m & 15)) << 3) + (hid & 7);
                uint2 o; o.x = t_pack_bf16x2(s[0], s[1]); o.y = t_pack_bf16x2(s[2], s[3]);
                *(uint2*)(p.outp + off) = o;
            } else {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int n0 = (rbw + ip * 2 + ii) * 16 + q4 * 4;
                    const f32x4 a = ii ? v1 : v0;
                    if (EPI == EPI_LOGITS) {
                        float4 o; o.x = bf2f(f2bf(a[0])); o.y = bf2f(f2bf(a[1])); o.z = bf2f(f2bf(a[2])); o.w = bf2f(f2bf(a[3]));
                        *(float4*)(p.outf + (long)m * p.N + n0) = o;
                    } else if (EPI == EPI_RESID) {
                        bf16_t* hp = p.h + (long)m * p.N + n0;
                        const uint2 hv = *(const uint2*)hp;
                        const float h0 = __uint_as_float(hv.x << 16), h1 = __uint_as_float(hv.x & 0xffff0000u);
                        const float h2 = __uint_as_float(hv.y << 16), h3 = __uint_as_float(hv.y & 0xffff0000u);
                        uint2 o;
                        o.x = t_pack_bf16x2(h0 + bf2f(f2bf(a[0])), h1 + bf2f(f2bf(a[1])));
                        o.y = t_pack_bf16x2(h2 + bf2f(f2bf(a[2])), h3 + bf2f(f2bf(a[3])));
                        *(uint2*)hp = o;
                    } else {   // EPI_QKV
                        const int pos = *p.pos;
                        const int sec = n0 / p.dim, within = n0 - sec * p.dim, hh = within >> 6, d0 = within & 63;
                        const float x0 = bf2f(f2bf(a[0])), x1 = bf2f(f2bf(a[1])), x2 = bf2f(f2bf(a[2])), x3 = bf2f(f2bf(a[3]));
                        const long sb = ((long)m * p.H + hh) * p.SA * 64;
                        if (sec == 2) {
                            const int w = pos & 31, qv = w < 16 ? (w >> 2) : ((w - 16) >> 2), ev = w < 16 ? (w & 3) : (4 + ((w - 16) & 3));
                            bf16_t* vb = p.vc + sb + ((long)(pos >> 5) * 4 + (d0 >> 4)) * 512 + ((qv * 16 + (d0 & 15)) << 3) + ev;
                            vb[0] = f2bf(x0); vb[8] = f2bf(x1); vb[16] = f2bf(x2); vb[24] = f2bf(x3);
                        } else {
                            const float4 cs = *(const float4*)(p.rope + ((long)pos * 32 + (d0 >> 1)) * 2);
                            const float r0 = x0 * cs.x - x1 * cs.y, r1 = x1 * cs.x + x0 * cs.y;
                            const float r2 = x2 * cs.z - x3 * cs.w, r3 = x3 * cs.z + x2 * cs.w;
                            if (sec == 0) {
                                uint2 o;
                                o.x = t_pack_bf16x2(bf2f(f2bf(r0)) * 0.125f, bf2f(f2bf(r1)) * 0.125f);
                                o.y = t_pack_bf16x2(bf2f(f2bf(r2)) * 0.125f, bf2f(f2bf(r3)) * 0.125f);
                                *(uint2*)(p.qout + ((long)m * p.H + hh) * 64 + d0) = o;
                            } else {
                                uint2 o; o.x = t_pack_bf16x2(r0, r1); o.y = t_pack_bf16x2(r2, r3);
                                bf16_t* kb_ = p.kc + sb + ((long)(pos >> 4) * 2 + (d0 >> 5)) * 512 + ((((d0 & 31) >> 3) * 16 + (pos & 15)) << 3) + (d0 & 7);
                                *(uint2*)kb_ = o;
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int I, int J, int KG>
static int launch_t_ijk(const GemmDP& p, int epi, hipStream_t st) {
    const int Mb = (p.M + 15) / 16;
    const dim3 g(p.N / (32 * I), (Mb + 2 * J - 1) / (2 * J)), b(256 * KG);
    constexpr size_t stage = (size_t)2 * KG * (2 * I + 2 * J) * 2 * 1024;          // two stages
    constexpr size_t foldb = KG == 2 ? (size_t)4 * I * J * 1024 : 0;
    const size_t sh = stage > foldb ? stage : foldb;
#define LT(E)                                                                                                                      \
    do {                                                                                                                           \
        static bool attr = false;                                                                                                  \
        if (sh > 48 * 1024 && !attr) { (void)hipFuncSetAttribute((const void*)dec_gemm_t_kernel<I, J, KG, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; } \
        hipLaunchKernelGGL((dec_gemm_t_kernel<I, J, KG, E>), g, b, sh, st, p);                                                      \
    } while (0)
    if (epi == EPI_LOGITS) LT(EPI_LOGITS); else if (epi == EPI_RESID) LT(EPI_RESID); else if (epi == EPI_SWIGLU) LT(EPI_SWIGLU); else LT(EPI_QKV);
#undef LT
    return 0;
}

// cfg = I*100 + J*10 + KG: workgroup tile (32·I) weight rows x (32·J) batch rows, KG groups of 4 waves over K.  -1 = outside the domain
// (442 is left out: 256 VGPRs and 370 B of scratch per lane).
extern "C" int car_launch_dec_gemm_t(const GemmDP* p, int epi, int cfg, hipStream_t st) {
    const int I = cfg / 100, J = (cfg / 10) % 10, KG = cfg % 10;
    if (p->wscale || p->nw || p->K % 32 || p->N % (32 * I) || p->M < 1) return -1;          // bf16 weights, no fused norm
    switch (cfg) {
        case 221: return launch_t_ijk<2, 2, 1>(*p, epi, st);
        case 222: return launch_t_ijk<2, 2, 2>(*p, epi, st);
        case 241: return launch_t_ijk<2, 4, 1>(*p, epi, st);
        case 242: return launch_t_ijk<2, 4, 2>(*p, epi, st);
        case 421: return launch_t_ijk<4, 2, 1>(*p, epi, st);
        case 422: return launch_t_ijk<4, 2, 2>(*p, epi, st);
        case 441: return launch_t_ijk<4, 4, 1>(*p, epi, st);
        default: return -1;
    }
}
