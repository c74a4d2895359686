// attn.hip — fused attention for the 64-wide heads of the compute-bound stages (fast mode): DINOv2 / ViT encoder (1025 or 197
// tokens, HF eager_attention_forward), LlamaGen prefill (120 text rows, causal + pad mask, gpt_t2i.py:446-470 via
// F.scaled_dot_product_attention) and the T5 caption encoder (120 rows, relative bias + key mask).  The unfused form materialised
// fp32 scores and bf16 probabilities in HBM (3 GEMM launches + softmax + 2 round trips per layer); this kernel keeps them in
// registers: online softmax over 32-key blocks, S^T = K Q^T and O^T += V^T P^T on v_mfma_f32_16x16x32_bf16.
//
//   workgroup = 4 waves, 64*QT queries of one (batch, head); wave w owns QT tiles of 16 queries.
//   K block [32 keys][64 d] and V^T block [64 d][32 keys] are staged through LDS once per workgroup (register double buffer:
//   the next block's global loads are in flight while the current block is multiplied).
//   S^T tile: A = K fragment (row = key), B = Q fragment (col = query)  ->  D[row = key (lane>>4)*4+r][col = query lane&15]
//   so a lane holds, for ITS query, keys {g*4..g*4+3} of both 16-key tiles (g = lane>>4): exactly the 8 "k" slots of the B operand
//   of the second product when V^T's A fragment enumerates keys in the same order {g*4+r, 16+g*4+r} — no transpose, no LDS trip for P.
//   Softmax statistics are per query = per lane column: reductions are two xor-shuffles (16, 32) across the four lane groups.
//
// Rounding points (fast mode): scores stay fp32 (as the fused SDPA kernels the reference runs), probabilities are rounded to bf16
// before P·V, the output is rounded once.  T5 mode reproduces the eager op sequence rnd(rnd(q·k) + bias).
#include "car_common.h"
#include "kernel_params.h"

#define FA_KLD 72      // K tile row stride (elements): 144 B keeps the 16-byte fragment reads of 16 rows on distinct banks
#define FA_VLD 40      // V^T tile row stride: 80 B

template <int QT, int MODE>
__global__ __launch_bounds__(256) void flash64_kernel(FlashP p) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[32 * FA_KLD];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * FA_VLD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    const int h = blockIdx.y; const long b = blockIdx.z;
    const int qbase = blockIdx.x * (64 * QT) + w * (16 * QT);
    const bf16_t* qp = p.q + b * p.q_sb + h * 64;
    const bf16_t* kp = p.k + b * p.k_sb + h * 64;
    const bf16_t* vp = p.vt + b * p.vt_sb + (long)h * 64 * p.vt_ld;
    const unsigned char* mk = MODE ? p.mask + b * p.Tk : nullptr;

    bf16x8 qf[QT][2];
    int qi[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        qi[t] = qbase + t * 16 + c;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (qi[t] < p.Tq) v = *(const uint4*)(qp + (long)qi[t] * p.q_st + ks * 32 + g * 8);
            qf[t][ks] = *(bf16x8*)&v;
        }
    }
    float m[QT], l[QT];
    f32x4 acc[4][QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m[t] = -INFINITY; l[t] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[d][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    int nblk = (p.Tk + 31) >> 5;
    if (MODE == 1) {      // causal: no key beyond the workgroup's last query
        int last = blockIdx.x * (64 * QT) + 64 * QT - 1; if (last > p.Tq - 1) last = p.Tq - 1;
        const int nb2 = (last >> 5) + 1; if (nb2 < nblk) nblk = nb2;
    }
    // staging assignment: K: thread -> (key = tid>>3, 16-byte chunk tid&7); V^T: (d = tid>>2, chunk tid&3)
    const int kk = tid >> 3, kc = tid & 7, vd = tid >> 2, vc = tid & 3;
    uint4 kreg, vreg;
    auto gload = [&](int blk) {
        const int key = blk * 32 + kk;
        kreg = make_uint4(0, 0, 0, 0);
        if (key < p.Tk) kreg = *(const uint4*)(kp + (long)key * p.k_st + kc * 8);
        vreg = *(const uint4*)(vp + (long)vd * p.vt_ld + blk * 32 + vc * 8);
    };
    gload(0);
    for (int blk = 0; blk < nblk; ++blk) {
        *(uint4*)&Ks[kk * FA_KLD + kc * 8] = kreg;
        *(uint4*)&Vs[vd * FA_VLD + vc * 8] = vreg;
        __syncthreads();
        if (blk + 1 < nblk) gload(blk + 1);
        // ---- S^T for 32 keys x 16*QT queries
        f32x4 s[2][QT];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const bf16x8 a0 = *(const bf16x8*)&Ks[(kt * 16 + c) * FA_KLD + g * 8];
            const bf16x8 a1 = *(const bf16x8*)&Ks[(kt * 16 + c) * FA_KLD + 32 + g * 8];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf[t][0], z, 0, 0, 0);
                s[kt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf[t][1], z, 0, 0, 0);
            }
        }
        // ---- mask / bias, online softmax (per query = per lane column)
        unsigned mkb[2] = {0xffffffffu, 0xffffffffu};     // 4 mask bytes per key tile
        if (MODE) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int j0 = blk * 32 + kt * 16 + g * 4;
                unsigned v = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) v |= (unsigned)((j0 + r < p.Tk) ? (mk[j0 + r] != 0) : 0) << (8 * r);
                mkb[kt] = v;
            }
        }
        bf16x8 pb[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float v[2][4];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int j0 = blk * 32 + kt * 16 + g * 4;
                float bz[4] = {0.f, 0.f, 0.f, 0.f};
                if (MODE == 2 && qi[t] < p.Tq) {
                    const float* bp = p.bias + ((long)h * p.Tq + qi[t]) * p.Tk + j0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (j0 + r < p.Tk) bz[r] = bp[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = j0 + r;
                    float x = s[kt][t][r] * p.scale;
                    if (MODE == 2) x = bf2f(f2bf(bf2f(f2bf(x)) + bz[r]));
                    const int inr = j < p.Tk;
                    int ok;
                    if (MODE == 0) ok = inr;
                    else if (MODE == 1) ok = inr & (j <= qi[t]) & ((((mkb[kt] >> (8 * r)) & 1u) != 0) | (j == qi[t]));
                    else ok = inr & (((mkb[kt] >> (8 * r)) & 1u) != 0);
                    x = ok ? x : -INFINITY;
                    v[kt][r] = x; mx = fmaxf(mx, x);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m[t], mx);
            const float base = mn == -INFINITY ? 0.f : mn;           // every key so far masked: all p = 0, alpha = 1
            const float alpha = __expf(m[t] - base);                  // m = -inf -> 0 (l and acc are 0 anyway)
            float ps = 0.f;
            unsigned pk[4];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const float p0 = __expf(v[kt][r] - base), p1 = __expf(v[kt][r + 1] - base);
                    ps += p0 + p1;
                    pk[kt * 2 + (r >> 1)] = (unsigned)f2bf(p0) | ((unsigned)f2bf(p1) << 16);
                }
            ps += __shfl_xor(ps, 16, 64); ps += __shfl_xor(ps, 32, 64);
            l[t] = l[t] * alpha + ps; m[t] = mn;
#pragma unroll
            for (int d = 0; d < 4; ++d) { acc[d][t][0] *= alpha; acc[d][t][1] *= alpha; acc[d][t][2] *= alpha; acc[d][t][3] *= alpha; }
            const uint4 u = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            pb[t] = *(const bf16x8*)&u;
        }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint2 lo = *(const uint2*)&Vs[(d * 16 + c) * FA_VLD + g * 4];
            const uint2 hi = *(const uint2*)&Vs[(d * 16 + c) * FA_VLD + 16 + g * 4];
            const uint4 u = make_uint4(lo.x, lo.y, hi.x, hi.y);
            const bf16x8 a = *(const bf16x8*)&u;
#pragma unroll
            for (int t = 0; t < QT; ++t) acc[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[t], acc[d][t], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- normalise and store: lane holds d = dt*16 + g*4 + r of query qi[t]
    bf16_t* op = p.o + b * p.o_sb + h * 64;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        if (qi[t] >= p.Tq) continue;
        const float inv = l[t] > 0.f ? 1.0f / l[t] : 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const unsigned o0 = (unsigned)f2bf(acc[d][t][0] * inv) | ((unsigned)f2bf(acc[d][t][1] * inv) << 16);
            const unsigned o1 = (unsigned)f2bf(acc[d][t][2] * inv) | ((unsigned)f2bf(acc[d][t][3] * inv) << 16);
            *(uint2*)(op + (long)qi[t] * p.o_st + d * 16 + g * 4) = make_uint2(o0, o1);
        }
    }
}

// 0 on success, -1 when the shape does not fit the kernel's alignment rules (caller falls back to the unfused path)
extern "C" int car_launch_flash64(const FlashP* pp, int B, hipStream_t st) {
    const FlashP& p = *pp;
    if ((p.q_st & 7) || (p.k_st & 7) || (p.q_sb & 7) || (p.k_sb & 7) || (p.vt_ld & 31) || (p.vt_sb & 7) || (p.o_st & 3) || (p.o_sb & 3) ||
        ((uintptr_t)p.q & 15) || ((uintptr_t)p.k & 15) || ((uintptr_t)p.vt & 15) || ((uintptr_t)p.o & 7) || p.vt_ld < ((p.Tk + 31) & ~31)) return -1;
    if (p.mode == 1 && p.Tq != p.Tk) return -1;
    const int QT = p.Tq > 192 ? 2 : 1;
    const dim3 grid((p.Tq + 64 * QT - 1) / (64 * QT), p.H, B);
#define FA_GO(qt, md) hipLaunchKernelGGL((flash64_kernel<qt, md>), grid, dim3(256), 0, st, p)
    if (QT == 2) { if (p.mode == 0) FA_GO(2, 0); else if (p.mode == 1) FA_GO(2, 1); else FA_GO(2, 2); }
    else { if (p.mode == 0) FA_GO(1, 0); else if (p.mode == 1) FA_GO(1, 1); else FA_GO(1, 2); }
#undef FA_GO
    return 0;
}
