// experiments/decode5_lds_dma_gemm.hip (EXPERIMENT, not product) — the decode linears of a LARGE chain (>= 256 rows: the 768-sequence bench step) as an LDS-tiled MFMA GEMM.
//
// dec_gemm (decode2.hip) gives every wave its own K slice and its own copy of the operands: right for <= 64 rows, where the step is
// bound by weight bytes, but at 384 / 768 rows every weight chunk is re-fetched from L2 by M/64 workgroups and every X chunk by N/32,
// each workgroup ends in an LDS fold, and the kernel sits at ~6 % of the MFMA peak (profiles/r02_bench_b768_*: 25.9 us for the 3.8 GFLOP
// of wqkv at 384 rows).  Here a workgroup owns a (32·NI) x (32·MJ) output tile over the WHOLE K; its 4 waves sit in a 2 x 2 grid and
// SHARE the operands through LDS:
//   * both operands are already stored in HBM in MFMA fragment order (1 KiB chunk = 16 rows x 32 k as 64 lanes x 16 B: engine.hip
//     pack_decode_bf16 / the XP layout of decode2.hip), so ONE global_load_lds instruction (LDS-DMA, 16 B per lane, lane-linear destination)
//     moves a whole chunk with no staging registers, and a fragment read is ds_read_b128 at lane*16: conflict-free, no swizzle;
//   * a stage = 2 k-blocks x (2NI weight + 2MJ X chunks); THREE stages in a ring: stage t+2 is issued while t is consumed and t+1 is
//     still in flight — the per-wave wait is a counted vmcnt followed by a raw s_barrier (__syncthreads() would drain the DMA queue);
//   * one fp32 accumulation chain per output over ascending k, no fold, the epilogues of dec_gemm (same rounding points).
// Operand traffic per workgroup and k-block: (2NI + 2MJ) KiB for 4·NI·MJ MFMAs — 128 x 128 tiles read each weight byte M/128 times from
// L2 instead of M/64 and each X byte N/128 times instead of N/32.
// Results differ from dec_gemm's by fp32 summation order only (one chain instead of WAVES partial sums): tolerance-graded like every
// bf16-mode kernel; rows of one call all take the same chain, so identical rows stay bit-identical wherever they sit in the batch.
#include "car_common.h"

#ifndef CAR_GEMMDP_DEFINED
#error "include controlar_amd/csrc/decode2.hip first (GemmDP, the EPI_* constants and pack_bf16x2 come from there)"
#endif

typedef __attribute__((address_space(1))) const void d3_gptr_t;
typedef __attribute__((address_space(3))) void d3_lptr_t;

template <int WN, int WM, int NI, int MJ, int NS, int EPI>
__global__ __launch_bounds__(64 * WN * WM) void dec_gemm_lds_kernel(GemmDP p) {
    constexpr int KS = 2;                          // k-blocks per stage
    constexpr int NWAVE = WN * WM;
    constexpr int NW = WN * NI, NX = WM * MJ;      // weight row-blocks / m-blocks of the workgroup tile
    constexpr int ROWC = NW + NX;                  // chunks per k-block
    constexpr int CH = ROWC * KS;                  // chunks per stage
    constexpr int LPW = CH / NWAVE;                // DMA pieces per wave per stage
    constexpr int AHEAD = NS - 1;                  // stages issued ahead of the one being consumed
    static_assert(CH % NWAVE == 0 && NI % 2 == 0 && NS >= 3, "tile shape");
    extern __shared__ __attribute__((aligned(16))) u32x4 d3_lds[];            // [NS][CH][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave % WN, wm = wave / WN;
    const int nkb = p.K >> 5, Mb = (p.M + 15) >> 4, nk = nkb / KS;
    const int MT = (Mb + NX - 1) / NX;
    // XCD-aware tile order: workgroup ids are dealt round-robin to the 8 XCDs; give each XCD a contiguous run of tiles so that the m-tiles
    // sharing a weight row-block run on the same L2 (bijective for any grid size)
    int t = blockIdx.x; const int total = gridDim.x;
    { const int x = t & 7, q = total >> 3, r = total & 7; t = x * q + (x < r ? x : r) + (t >> 3); }
    const int nt = t / MT, mt = t - nt * MT;
    const int rb0 = nt * NW, mb0 = mt * NX;

    // this wave's DMA pieces of a stage: chunk c = wave + NWAVE·u = (ks, r); r < NW: weight row-block rb0 + r, else X m-block (clamped: rows
    // beyond the last m-block re-read it and are dropped in the epilogue)
    const char* src[LPW];
#pragma unroll
    for (int u = 0; u < LPW; ++u) {
        const int c = wave + NWAVE * u, ks = c / ROWC, r = c - ks * ROWC;
        if (r < NW) src[u] = (const char*)p.W + ((long)(rb0 + r) * nkb + ks) * 1024 + lane * 16;
        else { int mb = mb0 + (r - NW); mb = mb < Mb ? mb : Mb - 1; src[u] = (const char*)p.X + ((long)mb * nkb + ks) * 1024 + lane * 16; }
    }
    int k_issue = 0;                               // k-stage of the next issue
    auto issue = [&](int buf) {
#pragma unroll
        for (int u = 0; u < LPW; ++u)
            __builtin_amdgcn_global_load_lds((d3_gptr_t*)(src[u] + (long)k_issue * (KS * 1024)), (d3_lptr_t*)(d3_lds + (size_t)(buf * CH + wave + NWAVE * u) * 64), 16, 0, 0);
        ++k_issue;
    };
    f32x4 acc[NI][MJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto frags = [&](u32x4 (&a)[NI], u32x4 (&x)[MJ], int buf, int ks) {
        const u32x4* base = d3_lds + ((size_t)buf * CH + (size_t)ks * ROWC) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i] = base[(wn * NI + i) * 64];
#pragma unroll
        for (int j = 0; j < MJ; ++j) x[j] = base[(NW + wm * MJ + j) * 64];
    };
    auto mma = [&](const u32x4 (&a)[NI], const u32x4 (&x)[MJ]) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&a[i], *(const bf16x8*)&x[j], acc[i][j], 0, 0, 0);
    };
    // wait until this wave's pieces of the OLDEST outstanding stage have landed, `younger` later stages may stay in flight
    auto wait_oldest = [&](int younger) {
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPW) : "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // Software pipeline.  Registers hold the fragments of ONE k-block ahead of the MFMAs; the DMA ring holds AHEAD stages ahead of the reads:
    //   iteration kt:  [ds_read (kt, k-block 1)] [MFMA (kt, k-block 0)] [vmcnt: stage kt+1 landed] [s_barrier] [DMA issue: stage kt+AHEAD -> the buffer of
    //                  stage kt-1, whose last fragments every wave consumed before this barrier] [ds_read (kt+1, k-block 0)] [MFMA (kt, k-block 1)]
    // A staged buffer is read only after the counted vmcnt of every issuing wave AND a barrier the reader has passed (cdna_hip_programming.md §5).
#pragma unroll
    for (int s_ = 0; s_ < AHEAD; ++s_) if (s_ < nk) issue(s_);
    { const int issued = nk < AHEAD ? nk : AHEAD; wait_oldest(issued - 1); }
    __builtin_amdgcn_s_barrier();
    u32x4 a0[NI], x0[MJ], a1[NI], x1[MJ];
    frags(a0, x0, 0, 0);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        frags(a1, x1, buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, x0);
        __builtin_amdgcn_sched_barrier(0);
        const int nbuf = buf + 1 == NS ? 0 : buf + 1;
        if (kt + 1 < nk) {
            // outstanding DMA stages now: kt+1 .. min(kt+AHEAD-1, nk-1)
            const int last = kt + AHEAD - 1 < nk - 1 ? kt + AHEAD - 1 : nk - 1;
            wait_oldest(last - (kt + 1));
            __builtin_amdgcn_s_barrier();
            if (k_issue < nk) issue(buf >= 1 ? buf - 1 : NS - 1);
            frags(a0, x0, nbuf, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, x1);
        __builtin_amdgcn_sched_barrier(0);
        buf = nbuf;
    }

    // ---- epilogue: the units of dec_gemm_kernel (pair of adjacent row-blocks x m-block), arithmetic and reference lines as decode2.hip
    constexpr int IP = NI / 2;
    const int q4 = lane >> 4, c16 = lane & 15;
    const int rbw = rb0 + wn * NI;                 // first row-block of this wave (even: the w1 | w3 pairs stay together)
    int pos = 0;
    if (EPI == EPI_QKV) pos = *p.pos;
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        const int mb = mb0 + wm * MJ + j;
        if (mb >= Mb) continue;
        const int m = mb * 16 + c16;
        if (m >= p.M) continue;
#pragma unroll
        for (int ip = 0; ip < IP; ++ip) {
            const f32x4 v0 = acc[ip * 2][j], v1 = acc[ip * 2 + 1][j];
            if (EPI == EPI_SWIGLU) {               // row-blocks alternate w1 | w3: v0 = a, v1 = c (gpt_t2i.py:217)
                float s[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = bf2f(f2bf(v0[r])), g = bf2f(f2bf(v1[r]));
                    s[r] = bf2f(f2bf(silu_f(a))) * g;
                }
                const int hid = ((rbw >> 1) + ip) * 16 + q4 * 4;
                const int nkb2 = p.N >> 6;
                const long off = ((((long)(m >> 4) * nkb2 + (hid >> 5)) * 64 + ((hid & 31) >> 3) * 16 + (m & 15)) << 3) + (hid & 7);
                uint2 o; o.x = pack_bf16x2(s[0], s[1]); o.y = pack_bf16x2(s[2], s[3]);
                *(uint2*)(p.outp + off) = o;
            } else {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int n0 = (rbw + ip * 2 + ii) * 16 + q4 * 4;
                    const f32x4 a = ii ? v1 : v0;
                    if (EPI == EPI_LOGITS) {       // bf16 round then widen (gpt_t2i.py:470)
                        float4 o; o.x = bf2f(f2bf(a[0])); o.y = bf2f(f2bf(a[1])); o.z = bf2f(f2bf(a[2])); o.w = bf2f(f2bf(a[3]));
                        *(float4*)(p.outf + (long)m * p.N + n0) = o;
                    } else if (EPI == EPI_RESID) { // h = rnd(h + rnd(acc)) (gpt_t2i.py:305-306)
                        bf16_t* hp = p.h + (long)m * p.N + n0;
                        const uint2 hv = *(const uint2*)hp;
                        const float h0 = __uint_as_float(hv.x << 16), h1 = __uint_as_float(hv.x & 0xffff0000u);
                        const float h2 = __uint_as_float(hv.y << 16), h3 = __uint_as_float(hv.y & 0xffff0000u);
                        uint2 o;
                        o.x = pack_bf16x2(h0 + bf2f(f2bf(a[0])), h1 + bf2f(f2bf(a[1])));
                        o.y = pack_bf16x2(h2 + bf2f(f2bf(a[2])), h3 + bf2f(f2bf(a[3])));
                        *(uint2*)hp = o;
                    } else {                       // EPI_QKV: bf16 round, 2-D RoPE, q -> scratch, K / V rows -> packed cache at *pos (gpt_t2i.py:264-277, :522-532, :227-235)
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
                                o.x = pack_bf16x2(bf2f(f2bf(r0)) * 0.125f, bf2f(f2bf(r1)) * 0.125f);
                                o.y = pack_bf16x2(bf2f(f2bf(r2)) * 0.125f, bf2f(f2bf(r3)) * 0.125f);
                                *(uint2*)(p.qout + ((long)m * p.H + hh) * 64 + d0) = o;
                            } else {
                                uint2 o; o.x = pack_bf16x2(r0, r1); o.y = pack_bf16x2(r2, r3);
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

template <int WN, int WM, int NI, int MJ, int NS>
static int launch_gemm_lds(const GemmDP& p, int epi, hipStream_t st) {
    const int Mb = (p.M + 15) / 16, MT = (Mb + WM * MJ - 1) / (WM * MJ), NT = p.N / (16 * WN * NI);
    const dim3 g(NT * MT), b(64 * WN * WM);
    constexpr size_t sh = (size_t)NS * 2 * (WN * NI + WM * MJ) * 1024;
#define LT(E)                                                                                                                     \
    do {                                                                                                                          \
        static bool attr = false;                                                                                                 \
        if (sh > 48 * 1024 && !attr) { (void)hipFuncSetAttribute((const void*)dec_gemm_lds_kernel<WN, WM, NI, MJ, NS, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; } \
        hipLaunchKernelGGL((dec_gemm_lds_kernel<WN, WM, NI, MJ, NS, E>), g, b, sh, st, p);                                         \
    } while (0)
    if (epi == EPI_LOGITS) LT(EPI_LOGITS); else if (epi == EPI_RESID) LT(EPI_RESID); else if (epi == EPI_SWIGLU) LT(EPI_SWIGLU); else LT(EPI_QKV);
#undef LT
    return 0;
}

// cfg: 1 = 128 x 128 tile, 8 waves (4 x 2, wave tile 32 x 64), 4 stages     2 = 128 x 128, 4 waves (2 x 2, wave tile 64 x 64), 4 stages
//      3 = 64 x 64, 4 waves (2 x 2, wave tile 32 x 32), 5 stages             4 = 128 (n) x 64 (m), 8 waves (4 x 2, wave tile 32 x 32), 5 stages
//      5 = 64 (n) x 128 (m), 8 waves (2 x 4, wave tile 32 x 32), 5 stages
// -1 = outside the domain (the caller falls back to dec_gemm).
extern "C" int car_launch_dec_gemm_lds(const GemmDP* p, int epi, int cfg, hipStream_t st) {
    const int tn = (cfg == 3 || cfg == 5) ? 64 : 128;
    if (p->wscale || p->nw || p->K % 64 || p->N % tn || p->M < 1) return -1;      // bf16 weights, whole stages, no fused norm
    switch (cfg) {
        case 1: return launch_gemm_lds<4, 2, 2, 4, 4>(*p, epi, st);
        case 2: return launch_gemm_lds<2, 2, 4, 4, 4>(*p, epi, st);
        case 3: return launch_gemm_lds<2, 2, 2, 2, 5>(*p, epi, st);
        case 4: return launch_gemm_lds<4, 2, 2, 2, 5>(*p, epi, st);
        case 5: return launch_gemm_lds<2, 4, 2, 2, 5>(*p, epi, st);
        default: return -1;
    }
}
