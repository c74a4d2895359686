// experiments/decode2_lat.hip — dec_gemm (controlar_amd/csrc/decode2.hip) with the epilogue's dependent loads hoisted into the prologue.
// EXPERIMENT, NOT PRODUCT.  Generated from decode2.hip's dec_gemm_kernel by textual substitution (kernel renamed dec_gemm_lat_kernel,
// launcher car_launch_dec_gemm_lat_cfg); two changes only:
//   * RESID epilogue: the residual h[m][n0..n0+3] of every (unit, row-block) this wave will finish is loaded before the K loop;
//   * QKV epilogue: *pos and this lane's RoPE (cos, sin) pairs are loaded before the K loop.
// In the small-batch regime (2..16 rows) a decode linear is ~4-6 us of latency chains; these loads sat at the END of the chain (after
// the fold), one more ~1 us round trip each.  Arithmetic is untouched: experiments/small_chain checks bit-equality against dec_gemm and
// times the layer loop with either kernel.  Include decode2.hip first.
#ifndef CAR_GEMMDP_DEFINED
#error "include controlar_amd/csrc/decode2.hip first"
#endif

template <int I, int J, int WAVES, int EPI, int F8, int NORM>
__global__ __launch_bounds__(WAVES * 64) void dec_gemm_lat_kernel(GemmDP p) {
    extern __shared__ __attribute__((aligned(16))) float red_all[];   // [NORM: 16 x (K+8) bf16] then [WAVES][I*J][64] f32x4
    static_assert(!NORM || J == 1, "the fused-norm variant serves one m-block");
    const int xs_ld = p.K + 8;                                         // bf16 elements per LDS row: 16-byte reads of 16 rows hit 16 distinct bank groups
    bf16_t* xs = (bf16_t*)red_all;
    float* red = NORM ? red_all + (16 * xs_ld) / 2 : red_all;
    constexpr int XPU = F8 ? 2 : 1;                                    // X chunks (k-blocks) per weight load unit
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // raised wave priority: when this kernel shares a CU with the other decode chain's attention waves (HBM-bound, thousands of them),
    // the instruction arbiter serves these few latency-bound waves first
    if (p.w_nt & 2) __builtin_amdgcn_s_setprio(3);
    const int nkb = p.K >> 5, nku = nkb / XPU, Mb = (p.M + 15) >> 4;
    const int MT = (Mb + J - 1) / J;
    // XCD-aware tile order: workgroup ids are dealt round-robin to the 8 XCDs, so give each XCD a contiguous run of tiles —
    // the M tiles that share a weight row-block then hit the same L2
    int t = blockIdx.x; const int total = gridDim.x;
    if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);
    const int nt = t / MT, mt = t - nt * MT;
    const int rb0 = nt * I, mb0 = mt * J;
    const int jn = (Mb - mb0) < J ? (Mb - mb0) : J;                    // m-blocks that exist in this tile (wave-uniform)
    const int ku_lo = (int)((long)nku * wave / WAVES), ku_hi = (int)((long)nku * (wave + 1) / WAVES);
    const u32x4* wp = (const u32x4*)p.W + (long)rb0 * nku * 64 + lane;
    const u32x4* xp = (const u32x4*)p.X + (long)mb0 * nkb * 64 + lane;

    f32x4 acc[I][J];
#pragma unroll
    for (int i = 0; i < I; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const u32x4 zw = (u32x4){0u, 0u, 0u, 0u};
    // DEPTH load stages stay in flight per wave (a tile's K slice is short: the kernel is bound by how many bytes a CU
    // keeps outstanding, not by MFMA issue): ~28 KiB-chunks of operands per wave, within the register budget
    constexpr int DEPTH = (28 / (I + J * XPU)) < 2 ? 2 : ((28 / (I + J * XPU)) > 6 ? 6 : (28 / (I + J * XPU)));
    u32x4 wr[DEPTH][I], xr[DEPTH][J * XPU];
    auto load = [&](u32x4 (&w)[I], u32x4 (&x)[J * XPU], int ku) {
#pragma unroll
        for (int i = 0; i < I; ++i) {
            const u32x4* a = wp + ((long)i * nku + ku) * 64;
            w[i] = (p.w_nt & 1) ? __builtin_nontemporal_load(a) : *a;
        }
        if (!NORM) {
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int u = 0; u < XPU; ++u) { x[j * XPU + u] = zw; if (j < jn) x[j * XPU + u] = xp[((long)j * nkb + ku * XPU + u) * 64]; }
        }
    };
    const bf16_t* xl = xs + (lane & 15) * xs_ld + (lane >> 4) * 8;     // NORM: this lane's row / k offset inside a k-block
    auto compute = [&](const u32x4 (&w)[I], u32x4 (&x)[J * XPU], int ku) {
        if (NORM) {
#pragma unroll
            for (int u = 0; u < XPU; ++u) x[u] = *(const u32x4*)(xl + (ku * XPU + u) * 32);
        }
        long x8[J][2];
        if (F8 == 2) {
#pragma unroll
            for (int j = 0; j < J; ++j) { x8[j][0] = bf16x8_to_fp8x8_(x[j * XPU]); x8[j][1] = bf16x8_to_fp8x8_(x[j * XPU + XPU - 1]); }
        }
#pragma unroll
        for (int i = 0; i < I; ++i) {
            if (F8 == 2) {
                const long a0 = (long)(((unsigned long long)w[i][1] << 32) | w[i][0]), a1 = (long)(((unsigned long long)w[i][3] << 32) | w[i][2]);
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, x8[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, x8[j][1], acc[i][j], 0, 0, 0);
                }
            } else if (F8) {
                const bf16x8 a0 = fp8x8_to_bf16x8_(w[i][0], w[i][1]), a1 = fp8x8_to_bf16x8_(w[i][2], w[i][3]);
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, *(const bf16x8*)&x[j * XPU], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, *(const bf16x8*)&x[j * XPU + XPU - 1], acc[i][j], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < J; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w[i], *(const bf16x8*)&x[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    const int nkw = ku_hi - ku_lo;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) if (d < nkw) load(wr[d], xr[d], ku_lo + d);
    if (NORM) {
        // ---- prologue: one wave per row (the code of rmsnorm2_kernel), rows >= M are never stored downstream
        const int D = p.K, ng = D >> 2;
        for (int m = wave; m < p.M; m += WAVES) {
            const bf16_t* src = p.nidx ? p.nemb + (long)p.nidx[m] * D : p.nh_in + (long)m * D;
            const bf16_t* add = p.nadd ? p.nctrl + ((long)m * p.n_tok + (*p.pos - p.nT + 1)) * D : nullptr;
            float val[8][4];
            float ss = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int gi = lane + q * 64;
                if (gi < ng) {
                    const uint2 u = *(const uint2*)(src + gi * 4);
                    float v[4] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
                    if (add) {
                        const uint2 a = *(const uint2*)(add + gi * 4);
                        const float c[4] = {__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xffff0000u), __uint_as_float(a.y << 16), __uint_as_float(a.y & 0xffff0000u)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = bf2f(f2bf(v[e] + bf2f(f2bf(p.ncs * c[e]))));
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { val[q][e] = v[e]; ss += v[e] * v[e]; }
                }
            }
            const float rstd = rsqrtf(wave_sum(ss) / D + p.neps);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int gi = lane + q * 64;
                if (gi < ng) {
                    const int k = gi * 4;
                    if (p.nh_out && blockIdx.x == 0) { uint2 u; u.x = pack_bf16x2(val[q][0], val[q][1]); u.y = pack_bf16x2(val[q][2], val[q][3]); *(uint2*)(p.nh_out + (long)m * D + k) = u; }
                    const uint2 wu = *(const uint2*)(p.nw + k);
                    const float w[4] = {__uint_as_float(wu.x << 16), __uint_as_float(wu.x & 0xffff0000u), __uint_as_float(wu.y << 16), __uint_as_float(wu.y & 0xffff0000u)};
                    uint2 u;
                    u.x = pack_bf16x2(bf2f(f2bf(val[q][0] * rstd)) * w[0], bf2f(f2bf(val[q][1] * rstd)) * w[1]);
                    u.y = pack_bf16x2(bf2f(f2bf(val[q][2] * rstd)) * w[2], bf2f(f2bf(val[q][3] * rstd)) * w[3]);
                    *(uint2*)(xs + m * xs_ld + k) = u;
                }
            }
        }
        __syncthreads();
    }
    // ---- LAT: what the epilogue depends on is fetched HERE, under the weight stream, instead of after the fold: the residual (RESID), the
    // position and this lane's RoPE row (QKV).  Same values, same arithmetic — only issued ~a load latency earlier.
    constexpr int IP_ = I >= 2 ? I / 2 : 1, IW_ = I >= 2 ? 2 : 1, UPW = (IP_ * J + WAVES - 1) / WAVES;      // epilogue units per wave
    int pos_pf = 0; float4 cs_pf[UPW][IW_]; uint2 hv_pf[UPW][IW_];
    if (EPI == EPI_QKV) pos_pf = *p.pos;
#pragma unroll
    for (int t_ = 0; t_ < UPW; ++t_) {
        const int u = wave + t_ * WAVES;
#pragma unroll
        for (int ii = 0; ii < IW_; ++ii) { cs_pf[t_][ii] = make_float4(0.f, 0.f, 0.f, 0.f); hv_pf[t_][ii] = make_uint2(0u, 0u); }
        if (u < IP_ * J) {
            const int ip = u / J, j = u - ip * J;
            const int m = (mb0 + j) * 16 + (lane & 15);
            if (j < jn && m < p.M) {
#pragma unroll
                for (int ii = 0; ii < IW_; ++ii) {
                    const int n0 = (rb0 + ip * IW_ + ii) * 16 + (lane >> 4) * 4;
                    if (EPI == EPI_RESID) hv_pf[t_][ii] = *(const uint2*)(p.h + (long)m * p.N + n0);
                    if (EPI == EPI_QKV) {
                        const int sec = n0 / p.dim, within = n0 - sec * p.dim, d0 = within & 63;
                        if (sec != 2) cs_pf[t_][ii] = *(const float4*)(p.rope + ((long)pos_pf * 32 + (d0 >> 1)) * 2);
                    }
                }
            }
        }
    }
    for (int base = 0; base < nkw; base += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (base + d < nkw) {                                     // wave-uniform
                compute(wr[d], xr[d], ku_lo + base + d);
                if (base + d + DEPTH < nkw) load(wr[d], xr[d], ku_lo + base + d + DEPTH);
            }
        }
    }
    // ---- fold the WAVES K-slices in fixed order through LDS
    f32x4* rv = (f32x4*)red;
#pragma unroll
    for (int i = 0; i < I; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) rv[((wave * I + i) * J + j) * 64 + lane] = acc[i][j];
    __syncthreads();
    auto fold = [&](int i, int j) -> f32x4 {
        f32x4 s = rv[((0 * I + i) * J + j) * 64 + lane];
        for (int w = 1; w < WAVES; ++w) { const f32x4 v = rv[((w * I + i) * J + j) * 64 + lane]; s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
        return s;
    };
    // epilogue units: (pair of adjacent row-blocks, m-block) — the SwiGLU (a, c) pair must meet in one lane
    constexpr int IP = I >= 2 ? I / 2 : 1, IW = I >= 2 ? 2 : 1;
    const int q4 = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int t_ = 0; t_ < UPW; ++t_) {
        const int u = wave + t_ * WAVES;
        if (u >= IP * J) continue;
        const int ip = u / J, j = u - ip * J;
        if (j >= jn) continue;
        const int m = (mb0 + j) * 16 + c16;
        f32x4 v[IW];
#pragma unroll
        for (int ii = 0; ii < IW; ++ii) {
            v[ii] = fold(ip * IW + ii, j);
            if (F8) {
                const float4 sc = *(const float4*)(p.wscale + (rb0 + ip * IW + ii) * 16 + q4 * 4);
                v[ii][0] *= sc.x; v[ii][1] *= sc.y; v[ii][2] *= sc.z; v[ii][3] *= sc.w;
            }
        }
        if (m >= p.M) continue;
        if (EPI == EPI_SWIGLU) {
            // row-blocks alternate w1 | w3 (engine.hip car_load_tensor): v[0] = a, v[1] = c for hidden block (rb0/2 + ip)
            float s[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = bf2f(f2bf(v[0][r])), g = bf2f(f2bf(v[IW - 1][r]));
                s[r] = bf2f(f2bf(silu_f(a))) * g;
            }
            const int hid = ((rb0 >> 1) + ip) * 16 + q4 * 4;            // 4 consecutive hidden units
            const int nkb2 = p.N >> 6;                                   // (N/2)/32
            const long off = ((((long)(m >> 4) * nkb2 + (hid >> 5)) * 64 + ((hid & 31) >> 3) * 16 + (m & 15)) << 3) + (hid & 7);
            uint2 o; o.x = pack_bf16x2(s[0], s[1]); o.y = pack_bf16x2(s[2], s[3]);
            *(uint2*)(p.outp + off) = o;
        } else {
#pragma unroll
            for (int ii = 0; ii < IW; ++ii) {
                const int n0 = (rb0 + ip * IW + ii) * 16 + q4 * 4;
                const f32x4 a = v[ii];
                if (EPI == EPI_LOGITS) {
                    float4 o; o.x = bf2f(f2bf(a[0])); o.y = bf2f(f2bf(a[1])); o.z = bf2f(f2bf(a[2])); o.w = bf2f(f2bf(a[3]));
                    *(float4*)(p.outf + (long)m * p.N + n0) = o;
                } else if (EPI == EPI_RESID) {
                    bf16_t* hp = p.h + (long)m * p.N + n0;
                    const uint2 hv = hv_pf[t_][ii];
                    const float h0 = __uint_as_float(hv.x << 16), h1 = __uint_as_float(hv.x & 0xffff0000u);
                    const float h2 = __uint_as_float(hv.y << 16), h3 = __uint_as_float(hv.y & 0xffff0000u);
                    uint2 o;
                    o.x = pack_bf16x2(h0 + bf2f(f2bf(a[0])), h1 + bf2f(f2bf(a[1])));
                    o.y = pack_bf16x2(h2 + bf2f(f2bf(a[2])), h3 + bf2f(f2bf(a[3])));
                    *(uint2*)hp = o;
                } else {   // EPI_QKV
                    const int pos = pos_pf;
                    const int sec = n0 / p.dim, within = n0 - sec * p.dim, hh = within >> 6, d0 = within & 63;
                    const float x0 = bf2f(f2bf(a[0])), x1 = bf2f(f2bf(a[1])), x2 = bf2f(f2bf(a[2])), x3 = bf2f(f2bf(a[3]));   // Linear output -> bf16
                    const long sb = ((long)m * p.H + hh) * p.SA * 64;
                    if (sec == 2) {
                        const int w = pos & 31, qv = w < 16 ? (w >> 2) : ((w - 16) >> 2), ev = w < 16 ? (w & 3) : (4 + ((w - 16) & 3));
                        bf16_t* vb = p.vc + sb + ((long)(pos >> 5) * 4 + (d0 >> 4)) * 512 + ((qv * 16 + (d0 & 15)) << 3) + ev;
                        vb[0] = f2bf(x0); vb[8] = f2bf(x1); vb[16] = f2bf(x2); vb[24] = f2bf(x3);
                    } else {
                        const float4 cs = cs_pf[t_][ii];
                        const float r0 = x0 * cs.x - x1 * cs.y, r1 = x1 * cs.x + x0 * cs.y;
                        const float r2 = x2 * cs.z - x3 * cs.w, r3 = x3 * cs.z + x2 * cs.w;
                        if (sec == 0) {
                            // rotated q is rounded to bf16, then scaled by head_dim^-0.5 = 1/8 (exact)
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

template <int I, int J, int WAVES, int F8, int NORM>
static void launch_gemm_lat_ij(const GemmDP& p, int epi, hipStream_t st) {
    const int Mb = (p.M + 15) / 16, MT = (Mb + J - 1) / J, NT = p.N / (16 * I);
    const dim3 g(NT * MT), b(WAVES * 64);
    const size_t sh = (size_t)WAVES * I * J * 64 * 16 + (NORM ? (size_t)16 * (p.K + 8) * 2 : 0);
    static size_t attr[4] = {0, 0, 0, 0};
#define LG(E)                                                                                                                   \
    do {                                                                                                                        \
        if (sh > 48 * 1024 && sh > attr[E]) { (void)hipFuncSetAttribute((const void*)dec_gemm_lat_kernel<I, J, WAVES, E, F8, NORM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr[E] = sh; } \
        hipLaunchKernelGGL((dec_gemm_lat_kernel<I, J, WAVES, E, F8, NORM>), g, b, sh, st, p);                                        \
    } while (0)
    if (NORM) {                      // the residual-add epilogue never follows a norm (gpt_t2i.py:305-306)
        if (epi == EPI_LOGITS) LG(EPI_LOGITS); else if (epi == EPI_SWIGLU) LG(EPI_SWIGLU); else LG(EPI_QKV);
    } else {
        if (epi == EPI_LOGITS) LG(EPI_LOGITS); else if (epi == EPI_RESID) LG(EPI_RESID); else if (epi == EPI_SWIGLU) LG(EPI_SWIGLU); else LG(EPI_QKV);
    }
#undef LG
}

// the small-batch configurations only (one m-block): cfg = I*100 + 10 + (WAVES == 8), bf16 weights; fused-norm variant when p->nw is set
extern "C" int car_launch_dec_gemm_lat_cfg(const GemmDP* p, int epi, int cfg, hipStream_t st) {
    if (p->wscale || p->M > 16 || (cfg / 10) % 10 != 1 || p->N % (16 * (cfg / 100)) || p->K % 32) return -1;
    if (epi == EPI_SWIGLU && cfg < 200) return -1;
    if (p->nw) {
        if (epi == EPI_RESID || p->K > 2048) return -1;
        switch (cfg) {
            case 110: launch_gemm_lat_ij<1, 1, 4, 0, 1>(*p, epi, st); break; case 111: launch_gemm_lat_ij<1, 1, 8, 0, 1>(*p, epi, st); break;
            case 210: launch_gemm_lat_ij<2, 1, 4, 0, 1>(*p, epi, st); break; case 211: launch_gemm_lat_ij<2, 1, 8, 0, 1>(*p, epi, st); break;
            case 410: launch_gemm_lat_ij<4, 1, 4, 0, 1>(*p, epi, st); break; case 411: launch_gemm_lat_ij<4, 1, 8, 0, 1>(*p, epi, st); break;
            default: return -1;
        }
        return 0;
    }
    switch (cfg) {
        case 110: launch_gemm_lat_ij<1, 1, 4, 0, 0>(*p, epi, st); break; case 111: launch_gemm_lat_ij<1, 1, 8, 0, 0>(*p, epi, st); break;
        case 210: launch_gemm_lat_ij<2, 1, 4, 0, 0>(*p, epi, st); break; case 211: launch_gemm_lat_ij<2, 1, 8, 0, 0>(*p, epi, st); break;
        case 410: launch_gemm_lat_ij<4, 1, 4, 0, 0>(*p, epi, st); break; case 411: launch_gemm_lat_ij<4, 1, 8, 0, 0>(*p, epi, st); break;
        default: return -1;
    }
    return 0;
}
