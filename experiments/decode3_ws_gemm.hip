// experiments/decode3_ws_gemm.hip — weight-stationary slab GEMM for the decode linears of a LARGE chain (M >= 96 rows).
// EXPERIMENT, NOT PRODUCT (not built into libcontrolar_hip.so).  Result on MI355X (profiles/r02_ws_check_half_period.txt): correct
// against dec_gemm on all four epilogues at the first run, but SLOWER in isolation (few, long workgroups: wqkv 24-36 vs 20 us, w2 42-61 vs
// 11 us at M = 384) and no better beside the attention — the measurement that came with it shows why no variant can be: on two streams
// the attention of one chain (167 us, 6.3 TB/s) and the six linears of the other (82 us) take 255 us together = their serial sum.
// A kernel that saturates HBM starves every latency-bound kernel next to it; the 2.5 ms the two-chain graph hides per step is
// linear-beside-linear overlap (the chains run in lockstep), not linear-under-attention.
//
// Why (profiles/r02_small_batch.txt, "decode overlap experiments"): with two chains in flight the step time does not react to any
// launch-schedule change — what the linears of one chain and the attention of the other share is the path that feeds the CUs.
// dec_gemm (decode2.hip) owns a (16·I) x (16·J) tile per workgroup with its waves splitting K: every weight row-block is re-read by
// each of the M/(16·J) workgroups that sit above it (6 times at M = 384) and every X block by each of the N/(16·I) workgroups beside it —
// ~490 MB of operand traffic per chain-layer at 64 x 64 tiles.  Here ONE workgroup owns 16·I weight rows against ALL rows of the chain:
//
//   grid      N/(16·I) workgroups (x ceil(Mb / 8J) row groups when the chain has more than 128·J = 384 rows), 8 waves each
//   W         the workgroup's I row-blocks cross the fabric exactly ONCE: every iteration each wave fetches one 1-KiB fragment chunk
//             (non-temporal) and parks it in LDS (2 x 8 KiB ring, one barrier per iteration); all 8 waves read it back with
//             ds_read_b128 in fragment order — conflict-free, no swizzle needed (lane l reads bytes [16l, 16l+16) of a chunk)
//   X         wave w owns m-blocks {w, w+8, ...} (J of them) over the WHOLE K: fragments straight from L2 into registers (the XP layout
//             is the MFMA operand image), next iteration's loads in flight during the current MFMAs
//   acc       I x J tiles of v_mfma_f32_16x16x32_bf16 per wave, one fp32 chain over K in ascending order — no K split, no fold, no LDS
//             round trip for partial sums; the epilogues are those of dec_gemm (same rounding points; reference lines cited there)
//
// Operand traffic per linear: W once (N·K·2 B) + X once per workgroup (N/(16·I) · M·K·2 B): wqkv at M = 384 with I = 4: 9.8 + 59 MB
// against 118 MB for dec_gemm's 64 x 64 tiles; the accumulation order differs from dec_gemm (single chain instead of WAVES partial
// sums), so results agree to fp32 round-off before the bf16 rounding points, not bit for bit.
#include "../controlar_amd/csrc/car_common.h"

typedef __attribute__((ext_vector_type(4))) unsigned ws_u32x4;
typedef __attribute__((ext_vector_type(2))) float ws_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 ws_bf16x2;

__device__ inline unsigned ws_pack_bf16x2(float a, float b) {          // v_cvt_pk_bf16_f32 (round-to-nearest-even)
    const ws_f32x2 v = {a, b};
    const ws_bf16x2 r = __builtin_convertvector(v, ws_bf16x2);
    return *(const unsigned*)&r;
}

#ifndef CAR_GEMMDP_DEFINED          // decode2.hip defines the same struct when both files are compiled into one translation unit (experiments/)
enum { EPI_LOGITS = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_QKV = 3 };
struct GemmDP {
    const bf16_t* W; const bf16_t* X; int M, N, K; int w_nt; int f8_mfma; const float* wscale;
    bf16_t* h; bf16_t* outp; float* outf;
    bf16_t* qout; bf16_t* kc; bf16_t* vc; const float* rope; const int* pos; int H, SA, dim;
    const bf16_t* nh_in; const bf16_t* nemb; const int* nidx; bf16_t* nh_out; const bf16_t* nw; const bf16_t* nctrl;
    int nadd, nT, n_tok; float ncs, neps;
};
#endif

// I row-blocks per workgroup (I in {2, 4}; 8 / I k-blocks per iteration), J m-blocks per wave.
template <int I, int J, int EPI>
__global__ __launch_bounds__(512) void dec_gemm_ws_kernel(GemmDP p) {
    constexpr int KS = 8 / I;                                            // k-blocks (32 wide) per iteration: I·KS = 8 chunks = one per wave
    __shared__ __attribute__((aligned(16))) ws_u32x4 wl[2][8][64];       // [ring slot][chunk = ks·I + i][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p.w_nt & 2) __builtin_amdgcn_s_setprio(3);
    const int nkb = p.K >> 5, Mb = (p.M + 15) >> 4;
    const int nit = (nkb + KS - 1) / KS;
    const int rb0 = blockIdx.x * I;
    const int mbase = blockIdx.y * (8 * J);
    int mb[J]; bool mv[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { mb[j] = mbase + j * 8 + wave; mv[j] = mb[j] < Mb; }          // wave-uniform
    // this wave's W chunk of an iteration: row-block i_w = wave % I, k-block ks_w = wave / I
    const int i_w = wave % I, ks_w = wave / I;
    const ws_u32x4* wp = (const ws_u32x4*)p.W + ((long)(rb0 + i_w) * nkb) * 64 + lane;
    const ws_u32x4* xp = (const ws_u32x4*)p.X + lane;
    const ws_u32x4 zw = (ws_u32x4){0u, 0u, 0u, 0u};

    f32x4 acc[I][J];
#pragma unroll
    for (int i = 0; i < I; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto load_w = [&](int it) -> ws_u32x4 {
        const int kb = it * KS + ks_w;
        return kb < nkb ? __builtin_nontemporal_load(wp + (long)kb * 64) : zw;                 // wave-uniform guard (ragged last iteration)
    };
    auto load_x = [&](ws_u32x4 (&x)[J][KS], int it) {
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int kb = it * KS + ks;
                x[j][ks] = zw;
                if (mv[j] && kb < nkb) x[j][ks] = xp[((long)mb[j] * nkb + kb) * 64];
            }
    };
    auto compute = [&](int buf, const ws_u32x4 (&x)[J][KS]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            ws_u32x4 a[I];
#pragma unroll
            for (int i = 0; i < I; ++i) a[i] = wl[buf][ks * I + i][lane];
#pragma unroll
            for (int i = 0; i < I; ++i)
#pragma unroll
                for (int j = 0; j < J; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&a[i], *(const bf16x8*)&x[j][ks], acc[i][j], 0, 0, 0);
        }
    };
    ws_u32x4 xa[J][KS], xb[J][KS];
    ws_u32x4 wr = load_w(0);
    load_x(xa, 0);
    // one barrier per iteration: slot `it & 1` is rewritten at it + 2, after the barrier of it + 1, which a wave only reaches once it
    // has finished reading the slot at `it`
    for (int it = 0; it < nit; it += 2) {
        wl[0][wave][lane] = wr;
        __syncthreads();
        if (it + 1 < nit) { wr = load_w(it + 1); load_x(xb, it + 1); }
        compute(0, xa);
        if (it + 1 >= nit) break;
        wl[1][wave][lane] = wr;
        __syncthreads();
        if (it + 2 < nit) { wr = load_w(it + 2); load_x(xa, it + 2); }
        compute(1, xb);
    }

    // ---- epilogue: the units of dec_gemm_kernel (pair of adjacent row-blocks x m-block), each owned by the wave that accumulated it
    constexpr int IP = I / 2, IW = 2;
    const int q4 = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        if (!mv[j]) continue;
        const int m = mb[j] * 16 + c16;
        if (m >= p.M) continue;
#pragma unroll
        for (int ip = 0; ip < IP; ++ip) {
            const f32x4 v0 = acc[ip * IW][j], v1 = acc[ip * IW + 1][j];
            if (EPI == EPI_SWIGLU) {
                // row-blocks alternate w1 | w3 (engine.hip car_load_tensor): v0 = a, v1 = c for hidden block (rb0/2 + ip); gpt_t2i.py:217
                float s[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = bf2f(f2bf(v0[r])), g = bf2f(f2bf(v1[r]));
                    s[r] = bf2f(f2bf(silu_f(a))) * g;
                }
                const int hid = ((rb0 >> 1) + ip) * 16 + q4 * 4;
                const int nkb2 = p.N >> 6;
                const long off = ((((long)(m >> 4) * nkb2 + (hid >> 5)) * 64 + ((hid & 31) >> 3) * 16 + (m & 15)) << 3) + (hid & 7);
                uint2 o; o.x = ws_pack_bf16x2(s[0], s[1]); o.y = ws_pack_bf16x2(s[2], s[3]);
                *(uint2*)(p.outp + off) = o;
            } else {
#pragma unroll
                for (int ii = 0; ii < IW; ++ii) {
                    const int n0 = (rb0 + ip * IW + ii) * 16 + q4 * 4;
                    const f32x4 a = ii ? v1 : v0;
                    if (EPI == EPI_LOGITS) {                                   // bf16 round then widen (gpt_t2i.py:470)
                        float4 o; o.x = bf2f(f2bf(a[0])); o.y = bf2f(f2bf(a[1])); o.z = bf2f(f2bf(a[2])); o.w = bf2f(f2bf(a[3]));
                        *(float4*)(p.outf + (long)m * p.N + n0) = o;
                    } else if (EPI == EPI_RESID) {                             // h = rnd(h + rnd(acc)) (gpt_t2i.py:305-306)
                        bf16_t* hp = p.h + (long)m * p.N + n0;
                        const uint2 hv = *(const uint2*)hp;
                        const float h0 = __uint_as_float(hv.x << 16), h1 = __uint_as_float(hv.x & 0xffff0000u);
                        const float h2 = __uint_as_float(hv.y << 16), h3 = __uint_as_float(hv.y & 0xffff0000u);
                        uint2 o;
                        o.x = ws_pack_bf16x2(h0 + bf2f(f2bf(a[0])), h1 + bf2f(f2bf(a[1])));
                        o.y = ws_pack_bf16x2(h2 + bf2f(f2bf(a[2])), h3 + bf2f(f2bf(a[3])));
                        *(uint2*)hp = o;
                    } else {                                                   // EPI_QKV (gpt_t2i.py:264-277, :522-532, :227-235)
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
                                o.x = ws_pack_bf16x2(bf2f(f2bf(r0)) * 0.125f, bf2f(f2bf(r1)) * 0.125f);
                                o.y = ws_pack_bf16x2(bf2f(f2bf(r2)) * 0.125f, bf2f(f2bf(r3)) * 0.125f);
                                *(uint2*)(p.qout + ((long)m * p.H + hh) * 64 + d0) = o;
                            } else {
                                uint2 o; o.x = ws_pack_bf16x2(r0, r1); o.y = ws_pack_bf16x2(r2, r3);
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

template <int I, int J>
static void launch_ws_ij(const GemmDP& p, int epi, hipStream_t st) {
    const int Mb = (p.M + 15) / 16;
    const dim3 g(p.N / (16 * I), (Mb + 8 * J - 1) / (8 * J)), b(512);
    if (epi == EPI_LOGITS) hipLaunchKernelGGL((dec_gemm_ws_kernel<I, J, EPI_LOGITS>), g, b, 0, st, p);
    else if (epi == EPI_RESID) hipLaunchKernelGGL((dec_gemm_ws_kernel<I, J, EPI_RESID>), g, b, 0, st, p);
    else if (epi == EPI_SWIGLU) hipLaunchKernelGGL((dec_gemm_ws_kernel<I, J, EPI_SWIGLU>), g, b, 0, st, p);
    else hipLaunchKernelGGL((dec_gemm_ws_kernel<I, J, EPI_QKV>), g, b, 0, st, p);
}

// I = weight row-blocks per workgroup (2 or 4).  Returns -1 when the shape is outside the kernel's domain (the caller falls back to dec_gemm).
extern "C" int car_launch_dec_gemm_ws(const GemmDP* p, int epi, int I, hipStream_t st) {
    if (p->wscale || p->nw || p->K % 32 || (I != 2 && I != 4) || p->N % (16 * I) || p->M < 1) return -1;     // bf16 weights, no fused norm
    const int Mb = (p->M + 15) / 16;
    int J = (Mb + 7) / 8;                               // m-blocks per wave when one workgroup spans the whole chain
    if (J > 3) J = 3;                                   // register budget (192 VGPRs at I = 4, J = 3; J = 6 spills): longer chains take several row groups
    if (I == 4) {
        switch (J) { case 1: launch_ws_ij<4, 1>(*p, epi, st); break; case 2: launch_ws_ij<4, 2>(*p, epi, st); break; default: launch_ws_ij<4, 3>(*p, epi, st); break; }
    } else {
        switch (J) { case 1: launch_ws_ij<2, 1>(*p, epi, st); break; case 2: launch_ws_ij<2, 2>(*p, epi, st); break; default: launch_ws_ij<2, 3>(*p, epi, st); break; }
    }
    return 0;
}
