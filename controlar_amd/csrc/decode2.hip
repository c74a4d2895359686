// decode2.hip — round-2 decode-step kernels (bf16 fast mode): every operand that an MFMA consumes is stored in HBM
// in the instruction's own fragment order, so a wave moves 1 KiB of contiguous bytes per load instruction straight into
// the registers the matrix core reads — no LDS staging, no shuffles, no layout fix-ups on the hot path.
//
//   dec_gemm   out = epi(X · W^T) for all b sequences of a chain in ONE pass over the weights (reference: the nn.Linear
//              calls of gpt_t2i.py:264 wqkv, :289 wo, :217 w1/w3/w2, :470 output).  W is the fragment-packed image built at
//              load time (engine_weights.hip pack_decode_bf16); X is fragment-packed by its producer (rmsnorm / attention /
//              SwiGLU epilogue).  A workgroup owns an (16·I n) x (16·J m) output tile over the WHOLE K; its waves split K
//              and fold their accumulators through LDS in a fixed order (deterministic, no fp32 partials in HBM).
//              Epilogues: QKV (bf16 round, 2-D RoPE, q -> scratch, K/V -> packed cache rows at *pos; gpt_t2i.py:264-277,
//              :522-532, :227-235), RESID (h = rnd(h + rnd(acc)); gpt_t2i.py:305-306), SWIGLU (rnd(rnd(silu(rnd a)) *
//              rnd c); gpt_t2i.py:217), LOGITS (bf16 round then widen; gpt_t2i.py:470).
//   dec_attn2  single-query attention over the valid prefix of the packed KV cache on the matrix cores:
//              S = K·q (A = K rows in fragment order, B = q broadcast), online softmax on 8 scores per lane,
//              O += P·V (A = P, B = V in fragment order).  reference: gpt_t2i.py:282-286 + the mask row of
//              generate.py:184-193; masked slots contribute exactly 0, never-written slots are never touched
//              (SURVEY.md Appendix E.4).
//
// Fragment-packed activation layout ("XP"), X[M][K] bf16:  element (m, k) lives at
//     (((m/16) * (K/32) + k/32) * 64 + ((k%32)/8) * 16 + m%16) * 8 + k%8
// i.e. chunk (mb, kb) is the 64-lane x 16-byte operand image of v_mfma_f32_16x16x32_bf16 (lane l: row l&15, k (l>>4)*8..+8).
// Packed KV cache per (sequence, head) stream of SA = roundup(S_max, 32) positions x 64 dims:
//     K (p, d): ((p/16)*2 + d/32) * 512 + (((d%32)/8)*16 + p%16) * 8 + d%8           (A operand of S = K·q)
//     V (p, d): ((p/32)*4 + d/16) * 512 + ((qv*16 + d%16) * 8 + ev),  w = p%32,       (B operand of O = P·V)
//               (qv, ev) = w < 16 ? (w/4, w%4) : ((w-16)/4, 4 + (w-16)%4)
// (the V position order inside a 32-block is the order in which the two 16x16 score tiles leave the accumulator, so P
// needs no cross-lane movement between the two MFMAs).
// e4m3 KV cache (car_config.kv_cache_fp8, opt-in; the reference has no such mode): one BYTE per element, unit scale, the same fragment geometry with two
// MFMA operands per 16-byte lane slot (as the e4m3 weight image), streams of SA x 64 bytes per (sequence, head):
//     K8 (p, d): (p/16)*1024 + (((d%32)/8)*16 + p%16)*16 + (d/32)*8 + d%8          (bytes 0-7: dims 0-31 operand, bytes 8-15: dims 32-63 operand)
//     V8 (p, d): (p/32)*2048 + (d/32)*1024 + (qv*16 + d%16)*16 + ((d/16)%2)*8 + ev
// dec_attn2 widens the bytes to bf16 in registers (exact: e4m3 is a subset of bf16) in front of the same bf16 MFMAs.
#include "car_common.h"

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ inline unsigned pack_bf16x2(float a, float b) {          // v_cvt_pk_bf16_f32 (round-to-nearest-even)
    const f32x2_t v = {a, b};
    const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return *(const unsigned*)&r;
}

#include "decode2_params.h"

#ifdef CAR_STAMP
#define STAMP(p, i) do { if (threadIdx.x == 0 && (p).stamp) { const int wg_ = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)); \
    if (wg_ < 2048) (p).stamp[((long)(p).stamp_slot * 2048 + wg_) * 8 + (i)] = (long long)wall_clock64(); } } while (0)
#else
#define STAMP(p, i) do { } while (0)
#endif

// L2 run-ahead helper (decode2_params.h CAR_PF_FIELDS): workgroup `hidx` of `pf_wgs` helpers touches one dword per 128-byte line of its XCD's eighth of up to two
// tensors, U lines per lane in flight.  Plain loads (default cache policy: the consumers' non-temporal loads hit the lines); nothing is stored.
template <typename P>
__device__ inline void car_pf_helper_kv(const P& p, int hidx, unsigned& acc) {
    if (!p.pf_kc) return;
    const int x = (int)(blockIdx.x & 7), R = p.pf_wgs >> 3, r = hidx >> 3, nth = (int)blockDim.x, tid = (int)threadIdx.x;
    if (r >= R) return;
    const int pos = *p.pf_pos;
    const unsigned L = (unsigned)((((pos >> 5) + 1) * 32 * 64 * p.pf_kvb) >> 7);      // lines of one stream's prefix (whole 32-position blocks), K and V alike
    const unsigned nix = (unsigned)((p.pf_items - x + 7) >> 3);                        // items of this XCD: x, x + 8, ...
    const unsigned tot = nix * 2u * L, step = (unsigned)(R * nth);
    const size_t stream = (size_t)p.pf_SA * 64 * p.pf_kvb;
    for (unsigned i = (unsigned)(r * nth + tid); i < tot; i += step) {
        const unsigned it = i / (2u * L), rem = i - it * 2u * L;
        const char* base = (const char*)(rem < L ? p.pf_kc : p.pf_vc) + (size_t)(x + 8 * (int)it) * stream;
        acc ^= *(const unsigned*)(base + (size_t)(rem < L ? rem : rem - L) * 128);
    }
}
__device__ inline void car_pf_helper(const void* p0, unsigned b0, const void* p1, unsigned b1, int hidx, int pf_wgs) {
    const int x = (int)(blockIdx.x & 7), R = pf_wgs >> 3, r = hidx >> 3, nth = (int)blockDim.x, tid = (int)threadIdx.x;
    if (r >= R) return;
    unsigned acc = 0;
#pragma unroll
    for (int job = 0; job < 2; ++job) {
        const unsigned* q = (const unsigned*)(job ? p1 : p0);
        const unsigned lines = (job ? b1 : b0) >> 7;
        if (!q || !lines) continue;
        const unsigned per = (lines + 7) >> 3, lo = (unsigned)x * per, hi = lo + per < lines ? lo + per : lines;
        const unsigned step = (unsigned)(R * nth);
        for (unsigned i = lo + (unsigned)(r * nth + tid); i < hi; i += step * 8) {
            unsigned v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const unsigned j = i + (unsigned)u * step; v[u] = 0; if (j < hi) v[u] = q[(size_t)j * 32]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
    }
    asm volatile("" ::"v"(acc));          // keeps the loads alive
}

// ---------------------------------------------------------------------------------------------- early launch (decode2_params.h CAR_HS_FIELDS)
// Compiled in only with -DCAR_EARLY_LAUNCH (experiments/lat_probe): in the product build HS_FRESH / HS_WT are the constant false and the loads / stores below are plain.
#ifdef CAR_EARLY_LAUNCH
#define HS_FRESH(p) ((p).dep != nullptr)
#define HS_WT(p) ((p).done != nullptr)
#define HS_WAIT(p) car_hs_wait((p).dep, (p).dep_n, (p).hs_err)
#define HS_ARRIVE(p) car_hs_arrive((p).done)
#else
#define HS_FRESH(p) false
#define HS_WT(p) false
#define HS_WAIT(p) do { } while (0)
#define HS_ARRIVE(p) do { } while (0)
#endif
typedef unsigned long long u64_t;
typedef __attribute__((address_space(1))) const void car_gptr_t;
typedef __attribute__((address_space(3))) void car_lptr_t;
__device__ __forceinline__ u64_t car_ld8_agent(const void* q) { return __hip_atomic_load((const u64_t*)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16 bytes the predecessor kernel wrote (fresh != 0: two 8-byte agent-scope loads) or anything older (a plain 16-byte load)
__device__ __forceinline__ u32x4 car_ld16(const void* q, bool fresh) {
    if (!fresh) return *(const u32x4*)q;
    const u64_t a = car_ld8_agent(q), b = car_ld8_agent((const char*)q + 8);
    u32x4 r; r[0] = (unsigned)a; r[1] = (unsigned)(a >> 32); r[2] = (unsigned)b; r[3] = (unsigned)(b >> 32);
    return r;
}
__device__ __forceinline__ uint2 car_ld8(const void* q, bool fresh) {
    if (!fresh) return *(const uint2*)q;
    const u64_t a = car_ld8_agent(q); uint2 r; r.x = (unsigned)a; r.y = (unsigned)(a >> 32); return r;
}
// stores of what the SUCCESSOR kernel reads: write-through agent-scope atomics when it may already be running (wt), plain stores otherwise
__device__ __forceinline__ void car_st8(void* q, uint2 v, bool wt) {
    if (wt) __hip_atomic_store((u64_t*)q, ((u64_t)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *(uint2*)q = v;
}
__device__ __forceinline__ void car_st4(void* q, unsigned v, bool wt) {
    if (wt) __hip_atomic_store((unsigned*)q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *(unsigned*)q = v;
}
__device__ __forceinline__ void car_st2(void* q, bf16_t v, bool wt) {
    if (wt) __hip_atomic_store((unsigned short*)q, (unsigned short)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *(bf16_t*)q = v;
}
__device__ __forceinline__ void car_st1(void* q, unsigned char v, bool wt) {
    if (wt) __hip_atomic_store((unsigned char*)q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *(unsigned char*)q = v;
}
// Whole workgroup: returns once every workgroup of the predecessor has arrived (or the schedule is declared dead).  One wave polls: lanes 0-7 the eight shard
// counters, lane 8 the sticky error word.  Bounded: ~2^17 polls (~0.1 s) then the error word is set and every later wait in the step returns at once.
__device__ __forceinline__ void car_hs_wait(const unsigned* dep, int dep_n, unsigned* err) {
    if (!dep) return;
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        const unsigned need = lane < 8 ? (unsigned)((dep_n + 7 - lane) >> 3) : 0u;
        for (unsigned spins = 0;; ++spins) {
            unsigned v = 0;
            if (lane < 8) v = __hip_atomic_load(dep + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (lane == 8) v = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool ok = lane < 8 ? v >= need : true, dead = lane == 8 && v != 0u;
            if (__all(ok) || __any(dead)) break;
            if (spins > (1u << 17)) { if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
}
// Whole workgroup, after its last store: every wave drains, the workgroup meets, one lane arrives on its shard.
__device__ __forceinline__ void car_hs_arrive(unsigned* done) {
    if (!done) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(done + (blockIdx.x & 7) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// OCP e4m3fn has no infinity: values beyond +-448 must saturate BEFORE the conversion (the oracle's kv_fp8 model and include/controlar_hip.h say
// clamp(-448, 448); an unclamped outlier would be stored as NaN and poison every later attention step of the sequence, since P * NaN = NaN even at P = 0)
__device__ inline float sat448(float v) { return __builtin_amdgcn_fmed3f(v, -448.0f, 448.0f); }

// OCP e4m3fn bytes -> bf16 (exact: e4m3 is a subset of bf16).  lo/hi: 4 bytes each = 8 consecutive k of one row.
// v_cvt_scalef32_pk_bf16_fp8 (gfx950): two bytes -> a packed bf16 pair in ONE instruction (scale 1.0); four per fragment where the
// cvt_pk_f32_fp8 + shift/mask form needed sixteen — it matters in dec_attn2's e4m3-KV form, which widens every byte it streams.
__device__ inline bf16x8 fp8x8_to_bf16x8_(unsigned lo, unsigned hi) {
    const bf16x2_t a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
    const bf16x2_t c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
    u32x4 r;
    r[0] = *(const unsigned*)&a; r[1] = *(const unsigned*)&b; r[2] = *(const unsigned*)&c; r[3] = *(const unsigned*)&d;
    return *(bf16x8*)&r;
}

// 8 bf16 (one X fragment) -> 8 OCP e4m3fn bytes, round-to-nearest-even, clamped to +-448 (unit activation scale)
__device__ inline long bf16x8_to_fp8x8_(const u32x4 x) {
    int lo = 0, hi = 0;
#define CL(v) __builtin_amdgcn_fmed3f((v), -448.0f, 448.0f)
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(CL(__uint_as_float(x[0] << 16)), CL(__uint_as_float(x[0] & 0xffff0000u)), lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(CL(__uint_as_float(x[1] << 16)), CL(__uint_as_float(x[1] & 0xffff0000u)), lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(CL(__uint_as_float(x[2] << 16)), CL(__uint_as_float(x[2] & 0xffff0000u)), hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(CL(__uint_as_float(x[3] << 16)), CL(__uint_as_float(x[3] & 0xffff0000u)), hi, true);
#undef CL
    return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}

// F8 = 1: weight-only e4m3 (BASELINE config 5; the reference has no fp8 path): one 16-byte weight load carries the fragments
// of TWO k-blocks, widened to bf16 in registers; the per-row scale multiplies the folded fp32 sum in the epilogue.
// F8 = 2: W8A8 — the same weight image, but the X fragments are quantised to e4m3 in registers (once per fragment, reused by the I
// weight row-blocks) and the product runs on the fp8 MFMA (v_mfma_f32_16x16x32_fp8_fp8): no widening of W at all.
// NORM = 1 (J = 1, M <= 16: the latency-bound small-batch regime): the RMSNorm that produces X runs in the prologue of every
// workgroup (16 rows x 2.5 KB from L2) while the first weight stages are already in flight, and X fragments come from LDS —
// one dependent kernel per linear fewer than "norm kernel -> GEMM".
template <int I, int J, int WAVES, int EPI, int F8, int NORM>
__global__ __launch_bounds__(WAVES * 64) void dec_gemm_kernel(GemmDP p) {
    extern __shared__ __attribute__((aligned(16))) float red_all[];   // [NORM: 16 x (K+8) bf16] then [WAVES][I*J][64] f32x4
    static_assert((NORM != 1 && NORM != 3) || J == 1, "the prologue-norm and staged-norm variants serve one m-block");
    const int xs_ld = p.K + 8;                                         // bf16 elements per LDS row: 16-byte reads of 16 rows hit 16 distinct bank groups
    bf16_t* xs = (bf16_t*)red_all;
    float* red = NORM == 1 ? red_all + (16 * xs_ld) / 2 : (NORM == 3 ? red_all + ((size_t)WAVES * p.stg) / 4 : red_all);
    constexpr int XPU = F8 ? 2 : 1;                                    // X chunks (k-blocks) per weight load unit
    // (the latency-bound tiles only — one m-block, 8 or 16 waves: on the 4-wave / multi-m-block tiles of 24-128-row chains the up-front wait cost 2-3 % of a layer)
    if (J == 1 && WAVES >= 8) car_kernarg_prefetch<(sizeof(GemmDP) + 63 + 48) / 64>();           // the block + the hidden grid-size arguments behind it
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p.pf_wgs > 0 && (int)blockIdx.x >= (int)gridDim.x - p.pf_wgs) {      // L2 run-ahead helper (whole workgroup): no tile, no barrier
        STAMP(p, 0);
        car_pf_helper(p.pf_p0, p.pf_b0, p.pf_p1, p.pf_b1, (int)blockIdx.x - ((int)gridDim.x - p.pf_wgs), p.pf_wgs);
        { unsigned acc = 0; car_pf_helper_kv(p, (int)blockIdx.x - ((int)gridDim.x - p.pf_wgs), acc); asm volatile("" ::"v"(acc)); }
        STAMP(p, 5);
        return;
    }
    // raised wave priority: when this kernel shares a CU with the other decode chain's attention waves (HBM-bound, thousands of them),
    // the instruction arbiter serves these few latency-bound waves first
    if (p.w_nt & 2) __builtin_amdgcn_s_setprio(3);
    STAMP(p, 0);
    const int nkb = p.K >> 5, nku = nkb / XPU, Mb = (p.M + 15) >> 4;
    const int MT = (Mb + J - 1) / J;
    // XCD-aware tile order: workgroup ids are dealt round-robin to the 8 XCDs, so give each XCD a contiguous run of tiles —
    // the M tiles that share a weight row-block then hit the same L2
    int t = blockIdx.x; const int total = (int)gridDim.x - p.pf_wgs;
    if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);
    const int nt = (NORM == 3) ? t : t / MT, mt = t - nt * MT;         // (NORM == 3: one m-block, MT == 1 — no division in the latency-bound prologue)
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
    // (round 4 tried 14 / 9 stages for the narrow tiles — w2 at K = 3584 leaves 14 k-blocks to each of 8 waves: the main loop's 2.3 us moved into the issue
    //  phase and the kernel stayed at 5.6 us: 80 workgroups x 115 KB is bound by what one CU pulls, ~60 GB/s; profiles/r04_lat_probe_v2_rows2.txt)
    // 16-wave tiles (round 6: the 80-workgroup linears wo / w2 of a small chain — twice the waves pulling per CU): 8 stages, so that w2's 7 k-blocks per wave are all in flight at once
    constexpr int DEPTH = WAVES == 16 ? 8 : ((28 / (I + J * XPU)) < 2 ? 2 : ((28 / (I + J * XPU)) > 6 ? 6 : (28 / (I + J * XPU))));
    u32x4 wr[DEPTH][I], xr[DEPTH][J * XPU], nr[DEPTH][NORM == 2 ? XPU : 1];
    // NORM == 2: X fragments are bf16 h rows read in place (row-major, ld = K: lane (c16, q4) takes 16 bytes of row m at k = 32 kb + 8 q4), normalised in registers
    const bf16_t* hrow[J]; bool hok[J]; float rstd[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { const int m = (mb0 + j) * 16 + (lane & 15); hok[j] = NORM == 2 && m < p.M; hrow[j] = NORM == 2 ? p.nh_in + (long)(hok[j] ? m : 0) * p.K + (lane >> 4) * 8 : nullptr; rstd[j] = 0.f; }
    const bool hs_fresh = HS_FRESH(p);               // early launch: X / the residual rows / their statistics come from a kernel that may still be running
    const bool hs_wt = HS_WT(p);                     // ... and this kernel's outputs are read by one that may already be
    // a stage's operands in two halves: what does not depend on the predecessor (weights, norm weight) and what does (X)
    auto loadW = [&](u32x4 (&w)[I], u32x4 (&nwf)[NORM == 2 ? XPU : 1], int ku) {
#pragma unroll
        for (int i = 0; i < I; ++i) {
            const u32x4* a = wp + ((long)i * nku + ku) * 64;
            w[i] = (p.w_nt & 1) ? __builtin_nontemporal_load(a) : *a;
        }
        if (NORM == 2) {
#pragma unroll
            for (int u = 0; u < XPU; ++u) nwf[u] = *(const u32x4*)(p.nw + (ku * XPU + u) * 32 + (lane >> 4) * 8);
        }
    };
    auto loadX = [&](u32x4 (&x)[J * XPU], int ku) {
        if (NORM == 0) {
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int u = 0; u < XPU; ++u) { x[j * XPU + u] = zw; if (j < jn) x[j * XPU + u] = car_ld16(xp + ((long)j * nkb + ku * XPU + u) * 64, hs_fresh); }      // (round 6 tried masking the lanes of rows >= M: the exec juggling cost more issue time than the narrower loads saved — wo 0.52 -> 0.86 us, w2 1.16 -> 1.42)
        } else if (NORM == 2) {
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int u = 0; u < XPU; ++u) { x[j * XPU + u] = zw; if (hok[j]) x[j * XPU + u] = car_ld16(hrow[j] + (ku * XPU + u) * 32, hs_fresh); }
        }
    };
    auto load = [&](u32x4 (&w)[I], u32x4 (&x)[J * XPU], u32x4 (&nwf)[NORM == 2 ? XPU : 1], int ku) { loadW(w, nwf, ku); loadX(x, ku); };
    const bf16_t* xl = xs + (lane & 15) * xs_ld + (lane >> 4) * 8;     // NORM == 1: this lane's row / k offset inside a k-block
    // NORM == 3: the wave-private staging region [X rounds][norm weight: 1 KiB][partials rounds], 1 KiB per DMA instruction
    const int s_nkbw = NORM == 3 ? (ku_hi - ku_lo) * XPU : 0;           // k-blocks of this wave's slice
    const int s_xr = NORM == 3 ? (p.M * ((nku + WAVES - 1) / WAVES) * XPU * 4 + 63) >> 6 : 0;
    const int s_row = (lane & 15) < p.M ? (lane & 15) : p.M - 1;
    const char* stg_x = (const char*)red_all + (size_t)wave * (NORM == 3 ? p.stg : 0);
    const char* stg_w = stg_x + (size_t)s_xr * 1024;
    const char* stg_s = stg_w + 1024;
    auto compute = [&](const u32x4 (&w)[I], u32x4 (&x)[J * XPU], const u32x4 (&nwf)[NORM == 2 ? XPU : 1], int ku) {
        if (NORM == 1) {
#pragma unroll
            for (int u = 0; u < XPU; ++u) x[u] = *(const u32x4*)(xl + (ku * XPU + u) * 32);
        } else if (NORM == 3) {
            // staged operands: this lane's 16 bytes of row min(c16, M-1) and of the norm weight for k-block (ku * XPU + u), then the arithmetic of NORM == 2
#pragma unroll
            for (int u = 0; u < XPU; ++u) {
                const int kq = ((ku - ku_lo) * XPU + u) * 4 + (lane >> 4);
                x[u] = *(const u32x4*)(stg_x + ((size_t)s_row * (s_nkbw * 4) + kq) * 16);
                const u32x4 wq = *(const u32x4*)(stg_w + (size_t)kq * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned hv = x[u][e], wv = wq[e];
                    const unsigned t = pack_bf16x2(__uint_as_float(hv << 16) * rstd[0], __uint_as_float(hv & 0xffff0000u) * rstd[0]);
                    x[u][e] = pack_bf16x2(__uint_as_float(t << 16) * __uint_as_float(wv << 16), __uint_as_float(t & 0xffff0000u) * __uint_as_float(wv & 0xffff0000u));
                }
            }
        } else if (NORM == 2) {
            // x = rnd(rnd(h * rstd) * w) — the rounding points of rmsnorm2_kernel (gpt_t2i.py:193-198), two bf16 per dword
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int u = 0; u < XPU; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned hv = x[j * XPU + u][e], wv = nwf[u][e];
                        const unsigned t = pack_bf16x2(__uint_as_float(hv << 16) * rstd[j], __uint_as_float(hv & 0xffff0000u) * rstd[j]);
                        x[j * XPU + u][e] = pack_bf16x2(__uint_as_float(t << 16) * __uint_as_float(wv << 16), __uint_as_float(t & 0xffff0000u) * __uint_as_float(wv & 0xffff0000u));
                    }
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
    // NORM prologue, part 1: the loads of this wave's FIRST row (residual row, control token, norm weight) go out BEFORE the weight stream.  A wave's loads
    // return in order: queued behind DEPTH stages of weights the norm waited for the whole burst, and fetched the norm weight only after the reduction —
    // the prologue measured 5.2-6.6 us of an 8 us kernel (experiments/lat_probe, profiles/r04_lat_probe_before_rows2.txt); issued first, it costs one L2 round trip.
    uint2 nh[8], na[8], nwv[8];
    bool n_add = false;
    if (NORM == 1 && !HS_FRESH(p)) {          // (early launch: these go out behind the wait, below)
        const int D = p.K, ng = D >> 2;
        if (wave < p.M) {
            const int m = wave;
            const bf16_t* src = p.nidx ? p.nemb + (long)p.nidx[m] * D : p.nh_in + (long)m * D;
            const bf16_t* add = p.nadd ? p.nctrl + ((long)m * p.n_tok + (*p.pos - p.nT + 1)) * D : nullptr;
            n_add = add != nullptr;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int gi = lane + q * 64;
                if (gi < ng) { nh[q] = *(const uint2*)(src + gi * 4); nwv[q] = *(const uint2*)(p.nw + gi * 4); if (add) na[q] = *(const uint2*)(add + gi * 4); }
            }
        }
    }
    // NORM == 2: rstd of this lane's rows from the producer's per-tile sums of squares, folded in a fixed order.  The four lanes (c16, q4 = 0..3) of a row share
    // the work: lane q4 takes the 16-byte chunks q4, q4 + 4, ... of the row's partials (at most 8 loads, all requested at once — a scalar loop over the 40-80
    // partials was 80 dependent L2 round trips: 7.7 us at 2 rows, 38-75 us at 64), sums them in chunk order, and the four lane sums fold through two shuffles:
    // (s0 + s1) + (s2 + s3), the same bits in every lane.  A wave's loads return in order, so these go out BEFORE the operand stages (behind them they waited
    // for the whole weight burst: 2.5-3.5 us); narrow tiles (J <= 2) keep them in registers across the stage issue, wide tiles finish them first.
    float4 sv[(NORM == 2 && J <= 2) ? J : 1][8];
    auto ssq_issue = [&](float4 (&v)[8], int j) {
        const int np4 = p.ssq_np >> 2, q4n = lane >> 4;
        // (round 6 tried unconditional loads at clamped addresses + selects here: fewer branches, but ~45 % more prologue instructions — at 24-64 rows, where 15 waves
        //  per CU run this prologue at once, the layer lost 6-32 us (profiles/r06_lat_probe_mid_rows_regression.txt).  One m-block chains use NORM == 3 instead.)
        const float4* sp = (const float4*)(p.ssq_in + (long)((mb0 + j) * 16 + (lane & 15)) * p.ssq_np);
#pragma unroll
        for (int t = 0; t < 8; ++t) { v[t] = make_float4(0.f, 0.f, 0.f, 0.f); if (hok[j] && q4n + 4 * t < np4) { const u32x4 r = car_ld16(sp + q4n + 4 * t, hs_fresh); v[t] = make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3])); } }
    };
    auto ssq_finish = [&](const float4 (&v)[8], int j) {
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) { sum += v[t].x; sum += v[t].y; sum += v[t].z; sum += v[t].w; }
        sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
        rstd[j] = hok[j] ? rsqrtf(sum / p.K + p.neps) : 0.f;
    };
    // Epilogue operands of this wave's FIRST epilogue unit (round 6): the residual values go out ahead of the weight stream, the RoPE row right behind it (its
    // address needs *pos: a scalar load requested here, waited for only after the stages are in flight) — in round 5 the epilogue started two dependent round
    // trips (pos -> rope row, or the residual load) after the fold.  Unconditional loads at clamped addresses; later units (u > wave) load in the loop as before.
    // (Early launch: the residual stream was last written two kernels back — complete before this kernel was dispatched.)
    constexpr int IPc = I >= 2 ? I / 2 : 1, IWc = I >= 2 ? 2 : 1;
    const bool epi_first = wave < IPc * J && p.M <= 16;                  // wave-uniform; one m-block chains only (the latency-bound regime: elsewhere the extra prologue instructions cost more than the round trip)
    uint2 hv_h[IWc]; float4 cs_h[IWc];
    int pos_h = 0;
    // *pos is read up front only in one-m-block chains (a single-branch graph).  With two chains as parallel graph branches every build in which the QKV epilogue
    // took the position from a register loaded before the main loop made the chain on the SECOND branch drift (tools/twin_probe.py; swapping the row groups between
    // the branches moves the fault with the branch).  It is NOT a stale value: instrumented builds compared the entry read (scalar and agent-scope) with the epilogue
    // read in every epilogue unit — 0 mismatches in 1.4e8, also in runs that drifted.  The faults are lane / wave level (a few rows of one 16-row block per event; q or
    // K units, never NaN), survive a sleep, an s_waitcnt vmcnt(0) or write-through stores in front of the epilogue stores, an opaque copy of the register and
    // poisoned VGPRs at entry; the build with round 5's fresh load in the epilogue never shows one (0 of 192 / 384 rows over 8 x 96 steps with the same
    // instrumentation).  Mechanism unknown (profiles/r06_posdbg_*.txt): the large-batch kernels keep the late read, and
    // tests/test_parity_gpu.py + bench.py's twin rows guard it.
    if (EPI == EPI_QKV && p.M <= 16) pos_h = *(const __attribute__((address_space(4))) int*)(unsigned long long)p.pos;      // constant for the kernel's lifetime: through the scalar cache
    if (EPI == EPI_RESID && epi_first) {
        const int ip0 = wave / J, j0 = wave - ip0 * J;
        int m0 = (mb0 + j0) * 16 + (lane & 15); m0 = m0 < p.M ? m0 : p.M - 1;
#pragma unroll
        for (int ii = 0; ii < IWc; ++ii) hv_h[ii] = *(const uint2*)(p.h + (long)m0 * p.N + (rb0 + ip0 * IWc + ii) * 16 + (lane >> 4) * 4);
    }
    auto rope_first = [&]() {
        if (EPI == EPI_QKV && epi_first) {
            const int ip0 = wave / J;
#pragma unroll
            for (int ii = 0; ii < IWc; ++ii) {
                const int n0 = (rb0 + ip0 * IWc + ii) * 16 + (lane >> 4) * 4, d0 = (n0 % p.dim) & 63;
                cs_h[ii] = *(const float4*)(p.rope + ((long)pos_h * 32 + (d0 >> 1)) * 2);
            }
        }
    };
    if (hs_fresh) {
        // ---- early launch: everything that does not depend on the predecessor first (weights of all stages, norm weight, RoPE row), then the wait, then X
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) if (d < nkw) loadW(wr[d], nr[d], ku_lo + d);
        rope_first();
        HS_WAIT(p);
        if (NORM == 1) {
            const int D = p.K, ng = D >> 2;
            if (wave < p.M) {
                const int m = wave;
                const bf16_t* src = p.nidx ? p.nemb + (long)p.nidx[m] * D : p.nh_in + (long)m * D;
                const bf16_t* add = p.nadd ? p.nctrl + ((long)m * p.n_tok + (*p.pos - p.nT + 1)) * D : nullptr;
                n_add = add != nullptr;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int gi = lane + q * 64;
                    if (gi < ng) { nh[q] = car_ld8(src + gi * 4, p.nidx == nullptr); nwv[q] = *(const uint2*)(p.nw + gi * 4); if (add) na[q] = *(const uint2*)(add + gi * 4); }
                }
            }
        }
        if (NORM == 2) {
#pragma unroll
            for (int j = 0; j < J; ++j) { if (J <= 2) ssq_issue(sv[j], j); else { ssq_issue(sv[0], j); ssq_finish(sv[0], j); } }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) if (d < nkw) loadX(xr[d], ku_lo + d);
    } else {
        if (NORM == 3 && s_nkbw > 0) {      // lane-dense DMA of this wave's X slice, norm-weight slice and the rows' partials (clamped chunk indices: every lane loads something valid)
            const int c4 = s_nkbw * 4, xtot = p.M * c4, kb0 = ku_lo * XPU;
            for (int r = 0; r < s_xr; ++r) {                            // wave-uniform trip count
                int ci = lane + 64 * r; ci = ci < xtot ? ci : xtot - 1;
                const int m = ci / c4, k8 = ci - m * c4;
                __builtin_amdgcn_global_load_lds((car_gptr_t*)(p.nh_in + (long)m * p.K + kb0 * 32 + k8 * 8), (car_lptr_t*)(stg_x + (size_t)r * 1024), 16, 0, 0);
            }
            { const int ci = lane < c4 ? lane : c4 - 1;
              __builtin_amdgcn_global_load_lds((car_gptr_t*)(p.nw + kb0 * 32 + ci * 8), (car_lptr_t*)stg_w, 16, 0, 0); }
            const int stot = p.M * (p.ssq_np >> 2), sr = (stot + 63) >> 6;
            for (int r = 0; r < sr; ++r) {
                int ci = lane + 64 * r; ci = ci < stot ? ci : stot - 1;
                __builtin_amdgcn_global_load_lds((car_gptr_t*)(p.ssq_in + (long)ci * 4), (car_lptr_t*)(stg_s + (size_t)r * 1024), 16, 0, 0);
            }
        }
        if (NORM == 2) {
#pragma unroll
            for (int j = 0; j < J; ++j) { if (J <= 2) ssq_issue(sv[j], j); else { ssq_issue(sv[0], j); ssq_finish(sv[0], j); } }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) if (d < nkw) load(wr[d], xr[d], nr[d], ku_lo + d);
        rope_first();
    }
    STAMP(p, 6);                                                       // every load of the prologue has been issued
    if (NORM == 3) {
        // the DMA is invisible to the compiler's wait-count pass: the covering vmcnt is ours (MI355X_MICROARCH.md: nothing orders a ds_read behind a pending LDS-DMA
        // but the issuing wave's vmcnt).  rstd from the staged partials in the order of NORM == 2: lane q4 adds chunks q4, q4 + 4, ..., then (s0 + s1) + (s2 + s3).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int np4 = p.ssq_np >> 2, q4n = lane >> 4;
        float sum = 0.f;
        for (int t = 0; 4 * t < np4; ++t) {
            const int ci = q4n + 4 * t;
            const float4 v = *(const float4*)(stg_s + ((size_t)s_row * np4 + (ci < np4 ? ci : np4 - 1)) * 16);
            const bool ok = ci < np4;
            sum += ok ? v.x : 0.f; sum += ok ? v.y : 0.f; sum += ok ? v.z : 0.f; sum += ok ? v.w : 0.f;
        }
        sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
        rstd[0] = (lane & 15) < p.M ? rsqrtf(sum / p.K + p.neps) : 0.f;
    }
    if (NORM == 2 && J <= 2) {
#pragma unroll
        for (int j = 0; j < J; ++j) ssq_finish(sv[J <= 2 ? j : 0], j);
    }
    if (NORM == 1) {
        // ---- prologue, part 2: one wave per row (the arithmetic of rmsnorm2_kernel), rows >= M are never stored downstream
        const int D = p.K, ng = D >> 2;
        for (int m = wave; m < p.M; m += WAVES) {
            if (m != wave) {      // further rows of this wave (M > WAVES): loaded here, behind the weight stages already in flight
                const bf16_t* src = p.nidx ? p.nemb + (long)p.nidx[m] * D : p.nh_in + (long)m * D;
                const bf16_t* add = p.nadd ? p.nctrl + ((long)m * p.n_tok + (*p.pos - p.nT + 1)) * D : nullptr;
                n_add = add != nullptr;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int gi = lane + q * 64;
                    if (gi < ng) { nh[q] = car_ld8(src + gi * 4, hs_fresh && p.nidx == nullptr); if (add) na[q] = *(const uint2*)(add + gi * 4); }
                }
            }
            float val[8][4];
            float ss = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int gi = lane + q * 64;
                if (gi < ng) {
                    const uint2 u = nh[q];
                    float v[4] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
                    if (n_add) {
                        const uint2 a = na[q];
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
                    const uint2 wu = nwv[q];
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
    STAMP(p, 1);
    for (int base = 0; base < nkw; base += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (base + d < nkw) {                                     // wave-uniform
                compute(wr[d], xr[d], nr[d], ku_lo + base + d);
#ifdef CAR_STAMP
                if (base + d == 0) { asm volatile("s_nop 0" ::"v"(acc[0][0][0])); STAMP(p, 2); }
#endif
                if (base + d + DEPTH < nkw) load(wr[d], xr[d], nr[d], ku_lo + base + d + DEPTH);
            }
        }
    }
    // ---- fold the WAVES K-slices in fixed order through LDS
    f32x4* rv = (f32x4*)red;
#ifdef CAR_STAMP
    asm volatile("s_nop 0" ::"v"(acc[0][0][0])); STAMP(p, 3);
#endif
#pragma unroll
    for (int i = 0; i < I; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) rv[((wave * I + i) * J + j) * 64 + lane] = acc[i][j];
    __syncthreads();
    STAMP(p, 4);
    auto fold = [&](int i, int j) -> f32x4 {
        f32x4 s = rv[((0 * I + i) * J + j) * 64 + lane];
        for (int w = 1; w < WAVES; ++w) { const f32x4 v = rv[((w * I + i) * J + j) * 64 + lane]; s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
        return s;
    };
    // epilogue units: (pair of adjacent row-blocks, m-block) — the SwiGLU (a, c) pair must meet in one lane
    constexpr int IP = I >= 2 ? I / 2 : 1, IW = I >= 2 ? 2 : 1;
    const int q4 = lane >> 4, c16 = lane & 15;
    for (int u = wave; u < IP * J; u += WAVES) {
        const bool first_u = u == wave && epi_first;                     // this unit's residual / RoPE operands were requested in the prologue
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
            // row-blocks alternate w1 | w3 (engine_weights.hip car_load_tensor): v[0] = a, v[1] = c for hidden block (rb0/2 + ip)
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
            car_st8(p.outp + off, o, hs_wt);
        } else {
            float ssq_acc = 0.f;
#pragma unroll
            for (int ii = 0; ii < IW; ++ii) {
                const int n0 = (rb0 + ip * IW + ii) * 16 + q4 * 4;
                const f32x4 a = v[ii];
                if (EPI == EPI_LOGITS) {
                    float4 o; o.x = bf2f(f2bf(a[0])); o.y = bf2f(f2bf(a[1])); o.z = bf2f(f2bf(a[2])); o.w = bf2f(f2bf(a[3]));
                    *(float4*)(p.outf + (long)m * p.N + n0) = o;
                } else if (EPI == EPI_RESID) {
                    bf16_t* hp = p.h + (long)m * p.N + n0;
                    const uint2 hv = first_u ? hv_h[ii] : *(const uint2*)hp;
                    const float h0 = __uint_as_float(hv.x << 16), h1 = __uint_as_float(hv.x & 0xffff0000u);
                    const float h2 = __uint_as_float(hv.y << 16), h3 = __uint_as_float(hv.y & 0xffff0000u);
                    uint2 o;
                    o.x = pack_bf16x2(h0 + bf2f(f2bf(a[0])), h1 + bf2f(f2bf(a[1])));
                    o.y = pack_bf16x2(h2 + bf2f(f2bf(a[2])), h3 + bf2f(f2bf(a[3])));
                    car_st8(hp, o, hs_wt);
                    if (p.ssq_out) {       // the squares of the STORED residual values: the next RMSNorm's row sum, one partial per (row, pair of row-blocks)
                        const float s0 = __uint_as_float(o.x << 16), s1 = __uint_as_float(o.x & 0xffff0000u), s2 = __uint_as_float(o.y << 16), s3 = __uint_as_float(o.y & 0xffff0000u);
                        ssq_acc += s0 * s0; ssq_acc += s1 * s1; ssq_acc += s2 * s2; ssq_acc += s3 * s3;
                    }
                } else {   // EPI_QKV
                    const int pos = p.M <= 16 ? pos_h : *p.pos;
                    const int sec = n0 / p.dim, within = n0 - sec * p.dim, hh = within >> 6, d0 = within & 63;
                    const float x0 = bf2f(f2bf(a[0])), x1 = bf2f(f2bf(a[1])), x2 = bf2f(f2bf(a[2])), x3 = bf2f(f2bf(a[3]));   // Linear output -> bf16
                    const long sb = ((long)m * p.H + hh) * p.SA * 64;
                    if (sec == 2) {
                        const int w = pos & 31, qv = w < 16 ? (w >> 2) : ((w - 16) >> 2), ev = w < 16 ? (w & 3) : (4 + ((w - 16) & 3));
                        if (p.kv8) {
                            unsigned char* vb = (unsigned char*)p.vc + sb + (long)(pos >> 5) * 2048 + (d0 >> 5) * 1024 + ((qv * 16 + (d0 & 15)) << 4) + ((d0 >> 4) & 1) * 8 + ev;
                            const int e01 = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(x0), sat448(x1), 0, false), e23 = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(x2), sat448(x3), 0, false);
                            car_st1(vb, (unsigned char)(e01 & 0xff), hs_wt); car_st1(vb + 16, (unsigned char)((e01 >> 8) & 0xff), hs_wt); car_st1(vb + 32, (unsigned char)(e23 & 0xff), hs_wt); car_st1(vb + 48, (unsigned char)((e23 >> 8) & 0xff), hs_wt);
                        } else {
                            bf16_t* vb = p.vc + sb + ((long)(pos >> 5) * 4 + (d0 >> 4)) * 512 + ((qv * 16 + (d0 & 15)) << 3) + ev;
                            car_st2(vb, f2bf(x0), hs_wt); car_st2(vb + 8, f2bf(x1), hs_wt); car_st2(vb + 16, f2bf(x2), hs_wt); car_st2(vb + 24, f2bf(x3), hs_wt);
                        }
                    } else {
                        const float4 cs = first_u ? cs_h[ii] : *(const float4*)(p.rope + ((long)pos * 32 + (d0 >> 1)) * 2);   // (cos, sin) of pairs d0/2, d0/2+1
                        const float r0 = x0 * cs.x - x1 * cs.y, r1 = x1 * cs.x + x0 * cs.y;
                        const float r2 = x2 * cs.z - x3 * cs.w, r3 = x3 * cs.z + x2 * cs.w;
                        if (sec == 0) {
                            // rotated q is rounded to bf16, then scaled by head_dim^-0.5 = 1/8 (exact)
                            uint2 o;
                            o.x = pack_bf16x2(bf2f(f2bf(r0)) * 0.125f, bf2f(f2bf(r1)) * 0.125f);
                            o.y = pack_bf16x2(bf2f(f2bf(r2)) * 0.125f, bf2f(f2bf(r3)) * 0.125f);
                            car_st8(p.qout + ((long)m * p.H + hh) * 64 + d0, o, hs_wt);
                        } else if (p.kv8) {       // rotated k: bf16 round (the Linear -> RoPE rounding points), then e4m3
                            unsigned char* kb_ = (unsigned char*)p.kc + sb + (long)(pos >> 4) * 1024 + ((((d0 & 31) >> 3) * 16 + (pos & 15)) << 4) + (d0 >> 5) * 8 + (d0 & 7);
                            int e = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(bf2f(f2bf(r0))), sat448(bf2f(f2bf(r1))), 0, false);
                            e = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(bf2f(f2bf(r2))), sat448(bf2f(f2bf(r3))), e, true);
                            car_st4(kb_, (unsigned)e, hs_wt);
                        } else {
                            uint2 o; o.x = pack_bf16x2(r0, r1); o.y = pack_bf16x2(r2, r3);
                            bf16_t* kb_ = p.kc + sb + ((long)(pos >> 4) * 2 + (d0 >> 5)) * 512 + ((((d0 & 31) >> 3) * 16 + (pos & 15)) << 3) + (d0 & 7);
                            car_st8(kb_, o, hs_wt);
                        }
                    }
                }
            }
            if (EPI == EPI_RESID && p.ssq_out) {       // lanes (c16, q4 = 0..3) hold the 4-column pieces of row m: fold them in a fixed order, one store per row
                ssq_acc += __shfl_xor(ssq_acc, 16, 64); ssq_acc += __shfl_xor(ssq_acc, 32, 64);
                if (q4 == 0) car_st4(p.ssq_out + (long)m * p.ssq_ld + (rb0 / IW + ip), __float_as_uint(ssq_acc), hs_wt);
            }
        }
    }
    HS_ARRIVE(p);
#ifdef CAR_STAMP
    __builtin_amdgcn_s_waitcnt(0); STAMP(p, 5);
#endif
}

template <int I, int J, int WAVES, int F8, int NORM>
static void launch_gemm_ij(const GemmDP& p_, int epi, hipStream_t st) {
    GemmDP p = p_;
    const int Mb = (p.M + 15) / 16, MT = (Mb + J - 1) / J, NT = p.N / (16 * I);
    const dim3 g(NT * MT + (p.pf_wgs > 0 ? p.pf_wgs : 0)), b(WAVES * 64);
    if (NORM == 3) {      // per-wave staging: X rounds + 1 (norm weight) + partials rounds, 1 KiB each
        const int nku = p.K / (F8 ? 64 : 32), xr = (p.M * ((nku + WAVES - 1) / WAVES) * (F8 ? 2 : 1) * 4 + 63) / 64, sr = (p.M * (p.ssq_np / 4) + 63) / 64;
        p.stg = (xr + 1 + sr) * 1024;
    }
    const size_t sh = (size_t)WAVES * I * J * 64 * 16 + (NORM == 1 ? (size_t)16 * (p.K + 8) * 2 : 0) + (NORM == 3 ? (size_t)WAVES * p.stg : 0);
    static size_t attr[4] = {0, 0, 0, 0};
#define LG(E)                                                                                                                   \
    do {                                                                                                                        \
        if (sh > 48 * 1024 && sh > attr[E]) { (void)hipFuncSetAttribute((const void*)dec_gemm_kernel<I, J, WAVES, E, F8, NORM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr[E] = sh; } \
        hipLaunchKernelGGL((dec_gemm_kernel<I, J, WAVES, E, F8, NORM>), g, b, sh, st, p);                                        \
    } while (0)
    if (NORM) {                      // the residual-add epilogue never follows a norm (gpt_t2i.py:305-306)
        if (epi == EPI_LOGITS) LG(EPI_LOGITS); else if (epi == EPI_SWIGLU) LG(EPI_SWIGLU); else LG(EPI_QKV);
    } else {
        if (epi == EPI_LOGITS) LG(EPI_LOGITS); else if (epi == EPI_RESID) LG(EPI_RESID); else if (epi == EPI_SWIGLU) LG(EPI_SWIGLU); else LG(EPI_QKV);
    }
#undef LG
}

// tile shape: I row-blocks x J m-blocks per workgroup, WAVES waves splitting K.  cfg = I*100 + J*10 + (WAVES == 8)
extern "C" int car_launch_dec_gemm_cfg(const GemmDP* p, int epi, int cfg, hipStream_t st) {
    if (epi == EPI_SWIGLU && cfg < 200) return -1;       // the (a, c) pair needs two adjacent row-blocks in one tile
    if (p->N % (16 * (cfg / 100)) || p->K % 32) return -1;
    if (p->wscale && p->K % 64) return -1;
    const int f8 = p->wscale ? (p->f8_mfma ? 2 : 1) : 0;
    if (p->ssq_in) {                                     // normalise-on-the-fly variant (NORM == 2): any tile; X = the bf16 residual rows themselves
        if (!p->nw || !p->nh_in || epi == EPI_RESID || p->ssq_np <= 0 || (p->ssq_np & 3) || p->ssq_np > 128) return -1;
        // one m-block, 8-wave tiles: the staged form (NORM == 3) — same arithmetic, lane-dense operand loads.  Limits: a slice of at most 16 k-blocks per wave
        // (one DMA instruction for the norm weight) and at least one k-unit for every wave.
        {
            const int nku = p->K / (f8 ? 64 : 32), nkw_max = (nku + 7) / 8;
            if (p->M <= 16 && (cfg / 10) % 10 == 1 && cfg % 10 == 1 && nku >= 8 && nkw_max * (f8 ? 2 : 1) <= 16 && !CAR_KNOB("CAR_NO_STAGED_NORMX")) {
                switch (cfg / 100) {
#define CASE3(I) case I: if (f8 == 2) launch_gemm_ij<I, 1, 8, 2, 3>(*p, epi, st); else if (f8) launch_gemm_ij<I, 1, 8, 1, 3>(*p, epi, st); else launch_gemm_ij<I, 1, 8, 0, 3>(*p, epi, st); return 0;
                    CASE3(1) CASE3(2) CASE3(4)
#undef CASE3
                    default: break;
                }
            }
        }
        switch (cfg) {
#define CASE(I, J) case I * 100 + J * 10: if (f8 == 2) launch_gemm_ij<I, J, 4, 2, 2>(*p, epi, st); else if (f8) launch_gemm_ij<I, J, 4, 1, 2>(*p, epi, st); else launch_gemm_ij<I, J, 4, 0, 2>(*p, epi, st); break; \
                   case I * 100 + J * 10 + 1: if (f8 == 2) launch_gemm_ij<I, J, 8, 2, 2>(*p, epi, st); else if (f8) launch_gemm_ij<I, J, 8, 1, 2>(*p, epi, st); else launch_gemm_ij<I, J, 8, 0, 2>(*p, epi, st); break;
            CASE(1, 1) CASE(1, 2) CASE(1, 4) CASE(2, 1) CASE(2, 2) CASE(2, 4) CASE(4, 1) CASE(4, 2) CASE(4, 4)
#undef CASE
            default: return -1;
        }
        return 0;
    }
    if (p->nw) {                                         // fused-norm variant: one m-block
        if (p->M > 16 || (cfg / 10) % 10 != 1 || epi == EPI_RESID || p->K > 2048) return -1;
        switch (cfg) {
#define CASE(I) case I * 100 + 10: if (f8 == 2) launch_gemm_ij<I, 1, 4, 2, 1>(*p, epi, st); else if (f8) launch_gemm_ij<I, 1, 4, 1, 1>(*p, epi, st); else launch_gemm_ij<I, 1, 4, 0, 1>(*p, epi, st); break; \
                case I * 100 + 11: if (f8 == 2) launch_gemm_ij<I, 1, 8, 2, 1>(*p, epi, st); else if (f8) launch_gemm_ij<I, 1, 8, 1, 1>(*p, epi, st); else launch_gemm_ij<I, 1, 8, 0, 1>(*p, epi, st); break;
            CASE(1) CASE(2) CASE(4)
#undef CASE
            default: return -1;
        }
        return 0;
    }
    switch (cfg) {
#define CASE(I, J) case I * 100 + J * 10: if (f8 == 2) launch_gemm_ij<I, J, 4, 2, 0>(*p, epi, st); else if (f8) launch_gemm_ij<I, J, 4, 1, 0>(*p, epi, st); else launch_gemm_ij<I, J, 4, 0, 0>(*p, epi, st); break; \
                   case I * 100 + J * 10 + 1: if (f8 == 2) launch_gemm_ij<I, J, 8, 2, 0>(*p, epi, st); else if (f8) launch_gemm_ij<I, J, 8, 1, 0>(*p, epi, st); else launch_gemm_ij<I, J, 8, 0, 0>(*p, epi, st); break;
        CASE(1, 1) CASE(1, 2) CASE(1, 4) CASE(2, 1) CASE(2, 2) CASE(2, 4) CASE(4, 1) CASE(4, 2) CASE(4, 4)
#undef CASE
        // 16 waves per 16-row tile (round 6): the narrow linears of a small chain (wo, w2: N / 16 = 80 workgroups) are bound by what one CU pulls; 16 waves keep
        // twice the bytes in flight per CU
        case 112: if (f8 == 2) launch_gemm_ij<1, 1, 16, 2, 0>(*p, epi, st); else if (f8) launch_gemm_ij<1, 1, 16, 1, 0>(*p, epi, st); else launch_gemm_ij<1, 1, 16, 0, 0>(*p, epi, st); break;
        default: return -1;
    }
    return 0;
}

// tile choice from the MI355X sweep of experiments/kbench.hip (profiles/r02_kbench.txt): XL shapes at M = 256/128/64/16/2
extern "C" int car_pick_gemm_cfg(int M, int N, int K, int epi) {
    const int Mb = (M + 15) / 16;
    const bool wideN = N >= 6144, hugeN = N >= 16384, longK = K >= 2048, smallNK = N <= 1536 && K <= 1536;
    int cfg;
    if (Mb >= 12) cfg = wideN ? 440 : 241;
    else if (Mb >= 6) cfg = hugeN ? 440 : (wideN ? 441 : (smallNK ? 110 : 221));
    else if (Mb >= 3) cfg = hugeN ? 441 : (wideN ? 241 : (longK ? 211 : 110));
    else if (Mb == 2) cfg = hugeN ? 421 : (wideN ? 221 : (longK ? 121 : 120));
    else cfg = hugeN ? 411 : (wideN ? 211 : 111);     // one m-block: 8 waves per tile everywhere (more bytes in flight per CU; one prologue-norm row per wave up to 8 rows)
    if (Mb == 1 && epi == EPI_RESID && cfg == 111 && N / 16 <= 128) cfg = 112;      // <= 128 workgroups: 16 waves each (wo, w2)
    int I = cfg / 100;
    if (epi == EPI_SWIGLU && I < 2) I = 2;
    while (I > 1 && N % (16 * I)) I >>= 1;
    return I * 100 + cfg % 100;
}

extern "C" void car_launch_dec_gemm(const GemmDP* p, int epi, hipStream_t st) {
    (void)car_launch_dec_gemm_cfg(p, epi, car_pick_gemm_cfg(p->M, p->N, p->K, epi), st);
}

// =============================================================================================== attention
// NWAVE waves share one (sequence, head): wave w of split s takes 32-position blocks blk0 + s*NWAVE + w, stride nsplit*NWAVE.
// PF = 1 keeps the next block's 8 KiB in flight in a second register set while the current one is consumed.
template <int NWAVE, int PF, int KV8>
__global__ __launch_bounds__(NWAVE * 64) void dec_attn2_kernel(Attn2P p) {
    __shared__ float red[NWAVE][66];
    const int split = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q4 = lane >> 4, c16 = lane & 15;
    STAMP(p, 0);
    const int pos = *p.pos;
    // Persistent form (n_seq > 0): every workgroup of the grid is resident from the start and walks items blockIdx.x, +gridDim.x, ...
    // A grid of thousands of (sequence, head) workgroups keeps the dispatcher busy for the whole kernel, and the OTHER decode chain's
    // short linears (a second branch of the captured graph) only get workgroup slots in its tail; a resident grid leaves them the
    // wave slots and registers it does not use.  Same arithmetic per item: results are bit-identical to the one-item-per-workgroup form.
    const bool persist = p.n_seq > 0;
    const int n_items = persist ? p.n_seq * p.H : 1;
    for (int it = persist ? (int)blockIdx.x : 0; it < n_items; it += gridDim.x) {
    const int h = persist ? it % p.H : (int)blockIdx.x, b = persist ? it / p.H : (int)blockIdx.y;
    const long sbase = ((long)b * p.H + h) * p.SA * 64;
    const u32x4* Kp = KV8 ? (const u32x4*)((const unsigned char*)p.kc + sbase) + lane : (const u32x4*)(p.kc + sbase) + lane;
    const u32x4* Vp = KV8 ? (const u32x4*)((const unsigned char*)p.vc + sbase) + lane : (const u32x4*)(p.vc + sbase) + lane;
    const bf16_t* qp = p.q + ((long)b * p.H + h) * 64;
    const bf16x8 qf0 = *(const bf16x8*)(qp + q4 * 8), qf1 = *(const bf16x8*)(qp + 32 + q4 * 8);
    const unsigned char* mk = p.mask ? p.mask + (long)b * p.T : nullptr;
    int jmin = 0;
    if (mk) {       // first attendable text position (left-padded prompts: everything before it is masked)
        if (p.jmin) jmin = p.jmin[b];
        else {
            int jm = p.T;
            for (int j = lane; j < p.T; j += 64) { const int v = mk[j]; if (v != 0 && j < jm) jm = j; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) jm = min(jm, __shfl_xor(jm, o, 64));
            jmin = jm;
        }
    }
    const int nblk = (pos >> 5) + 1, NW = p.nsplit * NWAVE;
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};

    // KV8: a 32-position block is 2 + 2 KiB instead of 4 + 4: two 16-byte loads per lane for K and two for V (kr[0..1], vr[0..1]), each carrying the
    // bytes of two MFMA operands
    // the text-pad mask bytes of a block inside the prefix are requested together with its K / V rows (they used to be fetched inside compute(), a second
    // dependent round trip for exactly the waves that own the text blocks — the tail of the one-launch small-batch form)
    auto load = [&](u32x4 (&kr)[4], u32x4 (&vr)[4], unsigned (&mb)[8], int blk) {
        if (mk != nullptr && blk * 32 < p.T) {
            const int jb = blk * 32 + q4 * 4;
#pragma unroll
            for (int e = 0; e < 8; ++e) mb[e] = mk[min(jb + (e < 4 ? e : 12 + e), p.T - 1)];
        }
#pragma unroll
        for (int i = 0; i < (KV8 ? 2 : 4); ++i) kr[i] = __builtin_nontemporal_load(Kp + ((long)blk * (KV8 ? 2 : 4) + i) * 64);
#pragma unroll
        for (int i = 0; i < (KV8 ? 2 : 4); ++i) vr[i] = __builtin_nontemporal_load(Vp + ((long)blk * (KV8 ? 2 : 4) + i) * 64);
    };
    auto compute = [&](const u32x4 (&kr_)[4], const u32x4 (&vr_)[4], const unsigned (&mb)[8], int blk) {
        u32x4 kr[4], vr[4];
        if (KV8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8 a = fp8x8_to_bf16x8_(kr_[i][0], kr_[i][1]), c2 = fp8x8_to_bf16x8_(kr_[i][2], kr_[i][3]);
                const bf16x8 v0 = fp8x8_to_bf16x8_(vr_[i][0], vr_[i][1]), v1 = fp8x8_to_bf16x8_(vr_[i][2], vr_[i][3]);
                kr[2 * i] = *(const u32x4*)&a; kr[2 * i + 1] = *(const u32x4*)&c2; vr[2 * i] = *(const u32x4*)&v0; vr[2 * i + 1] = *(const u32x4*)&v1;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { kr[i] = kr_[i]; vr[i] = vr_[i]; }
        }
        f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&kr[0], qf0, z4, 0, 0, 0);
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&kr[1], qf1, s0, 0, 0, 0);
        f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&kr[2], qf0, z4, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&kr[3], qf1, s1, 0, 0, 0);
        float sc[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        const int jb = blk * 32 + q4 * 4;
        const bool use_mk = mk != nullptr && blk * 32 < p.T;           // wave-uniform
        if (use_mk || blk * 32 + 31 > pos) {                           // wave-uniform: a block that needs per-position masking
            // branch-free selects (a short-circuit form of this test was mis-compiled by hipcc 7.2: the mask byte was loaded and dropped)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = jb + (e < 4 ? e : 12 + e);
                unsigned mv = 1u;
                if (use_mk) mv = mb[e];
                const bool ok = (j <= pos) & ((j >= p.T) | (mv != 0u));
                sc[e] = ok ? sc[e] : -INFINITY;
            }
        }
        float mx = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (mx > -INFINITY) {                                           // wave-uniform: something attendable in this block
            const float mn = fmaxf(m_run, mx), alpha = __expf(m_run - mn);
            float pe[8], ps = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { pe[e] = __expf(sc[e] - mn); ps += pe[e]; }
            l_run = l_run * alpha + ps; m_run = mn;
            u32x4 pu; pu[0] = pack_bf16x2(pe[0], pe[1]); pu[1] = pack_bf16x2(pe[2], pe[3]); pu[2] = pack_bf16x2(pe[4], pe[5]); pu[3] = pack_bf16x2(pe[6], pe[7]);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                o[d][0] *= alpha; o[d][1] *= alpha; o[d][2] *= alpha; o[d][3] *= alpha;
                o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&pu, *(const bf16x8*)&vr[d], o[d], 0, 0, 0);
            }
        }
    };
    if (PF) {
        u32x4 ka[4], va[4], kb2[4], vb2[4]; unsigned ma[8], mb2[8];
        int blk = (jmin >> 5) + split * NWAVE + wave;
        if (blk < nblk) load(ka, va, ma, blk);
        while (blk < nblk) {
            int nb = blk + NW;
            if (nb < nblk) load(kb2, vb2, mb2, nb);
            compute(ka, va, ma, blk);
            blk = nb;
            if (blk >= nblk) break;
            nb = blk + NW;
            if (nb < nblk) load(ka, va, ma, nb);
            compute(kb2, vb2, mb2, blk);
            blk = nb;
        }
    } else {
        u32x4 ka[4], va[4]; unsigned ma[8];
        for (int blk = (jmin >> 5) + split * NWAVE + wave; blk < nblk; blk += NW) { load(ka, va, ma, blk); compute(ka, va, ma, blk); }
    }
#ifdef CAR_STAMP
    asm volatile("s_nop 0" ::"v"(l_run)); STAMP(p, 3);
#endif
    // ---- merge: every lane of a q-group holds the same l partial; o[d][*] rows are identical (P rows are identical)
    float lt = l_run + __shfl_xor(l_run, 16, 64); lt += __shfl_xor(lt, 32, 64);
    if (q4 == 0) {
#pragma unroll
        for (int d = 0; d < 4; ++d) red[wave][2 + d * 16 + c16] = o[d][0];
    }
    if (lane == 0) { red[wave][0] = m_run; red[wave][1] = lt; }
    __syncthreads();
    STAMP(p, 4);
    if (tid < 64) {
        float M = red[0][0];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) M = fmaxf(M, red[w][0]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) {
            const float mm = red[w][0];
            const float a = mm > -INFINITY ? __expf(mm - M) : 0.f;
            L += red[w][1] * a; O += red[w][2 + tid] * a;
        }
        if (p.nsplit == 1) {
            const int k = h * 64 + tid;
            long off;
            if (p.out_packed) off = ((((long)(b >> 4) * (p.dim >> 5) + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (b & 15)) << 3) + (k & 7);
            else off = (long)b * p.dim + k;
            p.out[off] = f2bf(O / L);
        } else {
            float* pt = p.part + (((long)b * p.H + h) * p.nsplit + split) * 66;
            if (tid == 0) { pt[0] = M; pt[1] = L; }
            pt[2 + tid] = O;
        }
    }
    if (persist) __syncthreads();       // `red` is reused by the next item
    }
    if (persist && p.pf_wgs > 0 && (int)blockIdx.x >= n_items) car_pf_helper(p.pf_p0, p.pf_b0, p.pf_p1, p.pf_b1, (int)blockIdx.x - n_items, p.pf_wgs);
#ifdef CAR_STAMP
    __builtin_amdgcn_s_waitcnt(0); STAMP(p, 5);
#endif
}

// Small-batch attention (round 6; up to ~16 sequences: BASELINE configs 2, 4, 5): ONE 16-wave workgroup per (sequence, head) on a 1-D grid — items first, the L2
// run-ahead helpers (decode2_params.h CAR_PF_FIELDS) behind them.  What the 16-wave form of dec_attn2_kernel spent its 4.2 us on at 2 sequences (round-5 ISA and
// experiments/lat_probe): kernel arguments, then *pos, then q and jmin, then one K/V block per wave per round — four to five dependent round trips for a 160 KB stream.
// Here every wave requests TWO 32-position blocks at once (blocks jb + w and jb + w + 16: a 1024-position cache is one round trip; longer caches reuse the two
// register sets alternately) right behind the scalar loads of *pos and jmin[b]; q travels with the first block.  The arithmetic per block, the block -> wave
// assignment and the merge order are those of dec_attn2_kernel<16, 0, KV8> with nsplit = 1: bit-identical results (experiments/kbench checks it).
// Text-pad mask bytes only ever belong to a wave's FIRST block (launcher: T <= 512).
template <int KV8>
__global__ __launch_bounds__(1024) void dec_attn2s_kernel(Attn2P p) {
    __shared__ float red[16][66];
    car_kernarg_prefetch<(sizeof(Attn2P) + 63 + 48) / 64>();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q4 = lane >> 4, c16 = lane & 15;
    STAMP(p, 0);
    const int n_items = p.n_seq * p.H;
    if ((int)blockIdx.x >= n_items) {
        car_pf_helper(p.pf_p0, p.pf_b0, p.pf_p1, p.pf_b1, (int)blockIdx.x - n_items, p.pf_wgs);
#ifdef CAR_STAMP
        __builtin_amdgcn_s_waitcnt(0); STAMP(p, 5);
#endif
        return;
    }
    const int h = (int)blockIdx.x % p.H, b = (int)blockIdx.x / p.H;
    const int pos = *p.pos;
    const unsigned char* mk = p.mask ? p.mask + (long)b * p.T : nullptr;
    const int jmin = mk ? p.jmin[b] : 0;
    const long sbase = ((long)b * p.H + h) * p.SA * 64;
    const u32x4* Kp = KV8 ? (const u32x4*)((const unsigned char*)p.kc + sbase) + lane : (const u32x4*)(p.kc + sbase) + lane;
    const u32x4* Vp = KV8 ? (const u32x4*)((const unsigned char*)p.vc + sbase) + lane : (const u32x4*)(p.vc + sbase) + lane;
    const bf16_t* qp = p.q + ((long)b * p.H + h) * 64;
    bf16x8 qf0, qf1;                                                    // loaded behind the K / V requests (and behind the early-launch wait: the predecessor writes q)
    const int nblk = (pos >> 5) + 1;
    const bool hs_fresh = HS_FRESH(p), hs_wt = HS_WT(p);
    const int lastb = nblk - 1;                                         // the block that holds the new token's K / V row: written by the predecessor (wqkv)
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NL = KV8 ? 2 : 4;
    auto loadkv = [&](u32x4 (&kr)[4], u32x4 (&vr)[4], int blk) {
        if (hs_fresh && blk == lastb) {                                 // wave-uniform
#pragma unroll
            for (int i = 0; i < NL; ++i) kr[i] = car_ld16(Kp + ((long)blk * NL + i) * 64, true);
#pragma unroll
            for (int i = 0; i < NL; ++i) vr[i] = car_ld16(Vp + ((long)blk * NL + i) * 64, true);
        } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) kr[i] = __builtin_nontemporal_load(Kp + ((long)blk * NL + i) * 64);
#pragma unroll
            for (int i = 0; i < NL; ++i) vr[i] = __builtin_nontemporal_load(Vp + ((long)blk * NL + i) * 64);
        }
    };
    unsigned ma[8];
    auto compute = [&](const u32x4 (&kr_)[4], const u32x4 (&vr_)[4], int blk, bool use_mk) {
        u32x4 kr[4], vr[4];
        if (KV8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8 a = fp8x8_to_bf16x8_(kr_[i][0], kr_[i][1]), c2 = fp8x8_to_bf16x8_(kr_[i][2], kr_[i][3]);
                const bf16x8 v0 = fp8x8_to_bf16x8_(vr_[i][0], vr_[i][1]), v1 = fp8x8_to_bf16x8_(vr_[i][2], vr_[i][3]);
                kr[2 * i] = *(const u32x4*)&a; kr[2 * i + 1] = *(const u32x4*)&c2; vr[2 * i] = *(const u32x4*)&v0; vr[2 * i + 1] = *(const u32x4*)&v1;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { kr[i] = kr_[i]; vr[i] = vr_[i]; }
        }
        f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&kr[0], qf0, z4, 0, 0, 0);
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&kr[1], qf1, s0, 0, 0, 0);
        f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&kr[2], qf0, z4, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&kr[3], qf1, s1, 0, 0, 0);
        float sc[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        const int jb = blk * 32 + q4 * 4;
        if (use_mk || blk * 32 + 31 > pos) {                           // wave-uniform: a block that needs per-position masking
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = jb + (e < 4 ? e : 12 + e);
                unsigned mv = 1u;
                if (use_mk) mv = ma[e];
                const bool ok = (j <= pos) & ((j >= p.T) | (mv != 0u));
                sc[e] = ok ? sc[e] : -INFINITY;
            }
        }
        float mx = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (mx > -INFINITY) {                                           // wave-uniform: something attendable in this block
            const float mn = fmaxf(m_run, mx), alpha = __expf(m_run - mn);
            float pe[8], ps = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { pe[e] = __expf(sc[e] - mn); ps += pe[e]; }
            l_run = l_run * alpha + ps; m_run = mn;
            u32x4 pu; pu[0] = pack_bf16x2(pe[0], pe[1]); pu[1] = pack_bf16x2(pe[2], pe[3]); pu[2] = pack_bf16x2(pe[4], pe[5]); pu[3] = pack_bf16x2(pe[6], pe[7]);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                o[d][0] *= alpha; o[d][1] *= alpha; o[d][2] *= alpha; o[d][3] *= alpha;
                o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&pu, *(const bf16x8*)&vr[d], o[d], 0, 0, 0);
            }
        }
    };
    {
        u32x4 ka[4], va[4], kb2[4], vb2[4];
        int cur = (jmin >> 5) + wave;                                   // block in register set A; set B holds cur + 16
        const bool mk_a = mk != nullptr && cur * 32 < p.T && cur < nblk;   // wave-uniform
        if (mk_a) {
            const int jb = cur * 32 + q4 * 4;
#pragma unroll
            for (int e = 0; e < 8; ++e) ma[e] = mk[min(jb + (e < 4 ? e : 12 + e), p.T - 1)];
        }
        // early launch: the history blocks are requested BEFORE the wait (they were written by earlier steps), the block of the new row and q behind it
        const bool a_pre = cur < nblk && !(hs_fresh && cur == lastb), b_pre = cur + 16 < nblk && !(hs_fresh && cur + 16 == lastb);
        if (a_pre) loadkv(ka, va, cur);
        if (b_pre) loadkv(kb2, vb2, cur + 16);
        HS_WAIT(p);
        { const u32x4 q0 = car_ld16(qp + q4 * 8, hs_fresh), q1 = car_ld16(qp + 32 + q4 * 8, hs_fresh); qf0 = *(const bf16x8*)&q0; qf1 = *(const bf16x8*)&q1; }
        if (cur < nblk && !a_pre) loadkv(ka, va, cur);
        if (cur + 16 < nblk && !b_pre) loadkv(kb2, vb2, cur + 16);
        bool first = true;
        while (cur < nblk) {
            compute(ka, va, cur, first && mk_a);
            first = false;
            if (cur + 32 < nblk) loadkv(ka, va, cur + 32);
            if (cur + 16 < nblk) { compute(kb2, vb2, cur + 16, false); if (cur + 48 < nblk) loadkv(kb2, vb2, cur + 48); }
            cur += 32;
        }
    }
#ifdef CAR_STAMP
    asm volatile("s_nop 0" ::"v"(l_run)); STAMP(p, 3);
#endif
    // ---- merge (the statements of dec_attn2_kernel): every lane of a q-group holds the same l partial; o[d][*] rows are identical
    float lt = l_run + __shfl_xor(l_run, 16, 64); lt += __shfl_xor(lt, 32, 64);
    if (q4 == 0) {
#pragma unroll
        for (int d = 0; d < 4; ++d) red[wave][2 + d * 16 + c16] = o[d][0];
    }
    if (lane == 0) { red[wave][0] = m_run; red[wave][1] = lt; }
    __syncthreads();
    STAMP(p, 4);
    if (tid < 64) {
        float M = red[0][0];
#pragma unroll
        for (int w = 1; w < 16; ++w) M = fmaxf(M, red[w][0]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const float mm = red[w][0];
            const float a = mm > -INFINITY ? __expf(mm - M) : 0.f;
            L += red[w][1] * a; O += red[w][2 + tid] * a;
        }
        const int k = h * 64 + tid;
        long off;
        if (p.out_packed) off = ((((long)(b >> 4) * (p.dim >> 5) + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (b & 15)) << 3) + (k & 7);
        else off = (long)b * p.dim + k;
        car_st2(p.out + off, f2bf(O / L), hs_wt);
    }
    HS_ARRIVE(p);
#ifdef CAR_STAMP
    __builtin_amdgcn_s_waitcnt(0); STAMP(p, 5);
#endif
}

// text-pad mask rows for the decode batch: out[r][t] = emb_mask[row_img[r]][t] != 0 (all ones without a mask); generate.py:184-193
__global__ void build_mask_kernel(const int64_t* emb_mask, const int* row_img, unsigned char* out, int b, int T) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)b * T) return;
    const int r = (int)(i / T), t = (int)(i - (long)r * T);
    out[i] = emb_mask ? (unsigned char)(emb_mask[(long)row_img[r] * T + t] != 0) : (unsigned char)1;
}
extern "C" void car_launch_build_mask(const int64_t* emb_mask, const int* row_img, unsigned char* out, int b, int T, hipStream_t st) {
    const long n = (long)b * T;
    hipLaunchKernelGGL(build_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, emb_mask, row_img, out, b, T);
}

// jmin[b] = first position of the text prefix that may be attended (T if none); once per generate call
__global__ void mask_first_valid_kernel(const unsigned char* mask, int* jmin, int T) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int jm = T;
    for (int j = lane; j < T; j += 64) { const int v = mask[(long)b * T + j]; if (v != 0 && j < jm) jm = j; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) jm = min(jm, __shfl_xor(jm, o, 64));
    if (lane == 0) jmin[b] = jm;
}
extern "C" void car_launch_mask_first_valid(const unsigned char* mask, int* jmin, int b, int T, hipStream_t st) {
    hipLaunchKernelGGL(mask_first_valid_kernel, dim3(b), dim3(64), 0, st, mask, jmin, T);
}
// out[0] = min over v[0..n): the earliest attendable text position of the whole batch (the prefill window of engine_generate.hip starts there)
// `need` >= 0 and `flag`: the caller sized the prefill window from a HINT (car_sampling.first_valid_hint) instead of reading the minimum back; a minimum below the
// window start means valid prompt rows were dropped: raise the sticky host-mapped flag (engine_internal.h check_sticky)
__global__ __launch_bounds__(256) void min_int_kernel(const int* v, int n, int* out, int need, int* flag) {
    __shared__ int sm[4];
    int m = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 256) m = min(m, v[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) { const int r = min(min(sm[0], sm[1]), min(sm[2], sm[3])); out[0] = r; if (flag && need >= 0 && r < need) *flag = 1; }
}
extern "C" void car_launch_min_int(const int* v, int n, int* out, int need, int* flag, hipStream_t st) { hipLaunchKernelGGL(min_int_kernel, dim3(1), dim3(256), 0, st, v, n, out, need, flag); }

// split-KV combine -> bf16 attention output (XP-packed or row-major)
__global__ __launch_bounds__(64) void dec_attn2_combine_kernel(const float* part, bf16_t* out, int H, int nsplit, int dim, int out_packed) {
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* pt = part + ((long)b * H + h) * nsplit * 66;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, pt[s * 66]);
    float L = 0.f, O = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float mm = pt[s * 66];
        if (mm > -INFINITY) { const float a = __expf(mm - M); L += pt[s * 66 + 1] * a; O += pt[s * 66 + 2 + d] * a; }
    }
    const int k = h * 64 + d;
    long off;
    if (out_packed) off = ((((long)(b >> 4) * (dim >> 5) + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (b & 15)) << 3) + (k & 7);
    else off = (long)b * dim + k;
    out[off] = f2bf(O / L);
}

// variant = NWAVE*10 + PF  (42 = default: 4 waves, prefetch... see DESIGN.md for the measured choice)
// lds_pad: bytes of (unused) dynamic LDS requested per workgroup — an occupancy cap: with two decode chains in flight the
// attention of one chain must leave registers and wave slots on every CU for the other chain's GEMM workgroups
extern "C" void car_launch_dec_attn2_var(const Attn2P* p, int b, int variant, int lds_pad, hipStream_t st) {
    if (variant == 162 && p->nsplit == 1 && p->T <= 512 && (!p->mask || p->jmin)) {      // small-batch form: 1-D grid of (sequence, head) items + L2 run-ahead helpers
        Attn2P q = *p; q.n_seq = b; q.pgrid = 0;
        if (q.pf_wgs < 0 || ((b * q.H) & 7)) q.pf_wgs = 0;
        const dim3 g(b * q.H + q.pf_wgs);
        if (q.kv8) hipLaunchKernelGGL((dec_attn2s_kernel<1>), g, dim3(1024), 0, st, q); else hipLaunchKernelGGL((dec_attn2s_kernel<0>), g, dim3(1024), 0, st, q);
        return;
    }
    if (variant == 162) variant = 160;
    const bool persist = p->n_seq > 0 && p->pgrid > 0 && p->nsplit == 1;
    Attn2P q = *p; if (!persist) { q.n_seq = 0; q.pgrid = 0; q.pf_wgs = 0; }
    p = &q;
    const dim3 g = persist ? dim3(q.pgrid, 1, 1) : dim3(p->H, b, p->nsplit);
    const size_t sh = (size_t)(lds_pad > 0 ? lds_pad : 0);
#define LA(NW, PF_) do { if (p->kv8) hipLaunchKernelGGL((dec_attn2_kernel<NW, PF_, 1>), g, dim3(NW * 64), sh, st, *p); else hipLaunchKernelGGL((dec_attn2_kernel<NW, PF_, 0>), g, dim3(NW * 64), sh, st, *p); } while (0)
    switch (variant) {
        case 20: LA(2, 0); break;
        case 21: LA(2, 1); break;
        case 40: LA(4, 0); break;
        // 8 / 16 waves per (sequence, head): the small-batch form — with a handful of sequences ONE launch without split-KV partials and
        // without the combine kernel beats nsplit x 4 waves + combine (one dependent kernel less per layer, experiments/small_chain)
        case 80: LA(8, 0); break;
        case 81: LA(8, 1); break;
        case 160: LA(16, 0); break;
        case 161: LA(16, 1); break;
        default: LA(4, 1); break;
    }
#undef LA
    if (p->nsplit > 1 && p->out)
        hipLaunchKernelGGL(dec_attn2_combine_kernel, dim3(p->H, b), dim3(64), 0, st, p->part, p->out, p->H, p->nsplit, p->dim, p->out_packed);
}
extern "C" void car_launch_dec_attn2(const Attn2P* p, int b, hipStream_t st) { car_launch_dec_attn2_var(p, b, 40, 0, st); }
// =============================================================================================== prefill -> packed cache
// k/v of the T prefix rows -> packed cache, RoPE on q,k in place (reference: gpt_t2i.py:266-277).  Same arithmetic as
// decode.hip prefill_rope_kv_kernel; only the cache addressing differs.
__global__ void prefill_rope_kv2_kernel(bf16_t* qkv, bf16_t* kcache, bf16_t* vcache, const float* rope, int b, int Tn, int H, int dim, int SA, int kv8, int t0) {
    const long total = (long)b * Tn * H * 32;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int pr = (int)(i % 32); const int h = (int)((i / 32) % H); const int tw = (int)((i / (32L * H)) % Tn); const long bb = i / (32L * H * Tn);
        bf16_t* row = qkv + (bb * Tn + tw) * 3 * dim;
        const int t = tw + t0;                               // row tw of the prefill window is prefix position t0 + tw (cache row, rope row)
        const float cs = rope[((long)t * 32 + pr) * 2], sn = rope[((long)t * 32 + pr) * 2 + 1];
        const float q0 = bf2f(row[h * 64 + 2 * pr]), q1 = bf2f(row[h * 64 + 2 * pr + 1]);
        const float k0 = bf2f(row[dim + h * 64 + 2 * pr]), k1 = bf2f(row[dim + h * 64 + 2 * pr + 1]);
        row[h * 64 + 2 * pr] = f2bf(q0 * cs - q1 * sn); row[h * 64 + 2 * pr + 1] = f2bf(q1 * cs + q0 * sn);
        const bf16_t kr0 = f2bf(k0 * cs - k1 * sn), kr1 = f2bf(k1 * cs + k0 * sn);
        row[dim + h * 64 + 2 * pr] = kr0; row[dim + h * 64 + 2 * pr + 1] = kr1;
        const long sb = (bb * H + h) * (long)SA * 64;
        const int d = 2 * pr;
        const int w = t & 31, qv = w < 16 ? (w >> 2) : ((w - 16) >> 2), ev = w < 16 ? (w & 3) : (4 + ((w - 16) & 3));
        if (kv8) {      // e4m3 bytes (K8 / V8 layouts); the prefill's own attention reads the bf16 rows of `qkv`, not the cache
            const int ek = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(bf2f(kr0)), sat448(bf2f(kr1)), 0, false);
            unsigned char* kp = (unsigned char*)kcache + sb + (long)(t >> 4) * 1024 + ((((d & 31) >> 3) * 16 + (t & 15)) << 4) + (d >> 5) * 8 + (d & 7);
            kp[0] = (unsigned char)(ek & 0xff); kp[1] = (unsigned char)((ek >> 8) & 0xff);
            const int ev2 = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(bf2f(row[2 * dim + h * 64 + d])), sat448(bf2f(row[2 * dim + h * 64 + d + 1])), 0, false);
            unsigned char* vp = (unsigned char*)vcache + sb + (long)(t >> 5) * 2048 + (d >> 5) * 1024 + ((qv * 16 + (d & 15)) << 4) + ((d >> 4) & 1) * 8 + ev;
            vp[0] = (unsigned char)(ev2 & 0xff); vp[16] = (unsigned char)((ev2 >> 8) & 0xff);
        } else {
            bf16_t* kp = kcache + sb + ((long)(t >> 4) * 2 + (d >> 5)) * 512 + ((((d & 31) >> 3) * 16 + (t & 15)) << 3) + (d & 7);
            kp[0] = kr0; kp[1] = kr1;
            bf16_t* vp = vcache + sb + ((long)(t >> 5) * 4 + (d >> 4)) * 512 + ((qv * 16 + (d & 15)) << 3) + ev;
            vp[0] = row[2 * dim + h * 64 + d]; vp[8] = row[2 * dim + h * 64 + d + 1];
        }
    }
}
extern "C" void car_launch_prefill_rope_kv2(void* qkv, void* kc, void* vc, const float* rope, int b, int Tn, int H, int dim, int SA, int kv8, int t0, hipStream_t st) {
    long total = (long)b * Tn * H * 32; int g = (int)((total + 255) / 256); if (g > 4096) g = 4096;
    hipLaunchKernelGGL(prefill_rope_kv2_kernel, dim3(g), dim3(256), 0, st, (bf16_t*)qkv, (bf16_t*)kc, (bf16_t*)vc, rope, b, Tn, H, dim, SA, kv8, t0);
}

// =============================================================================================== RMSNorm -> packed xn
// Same arithmetic as ops.hip rmsnorm_kernel<bf16_t> (gpt_t2i.py:193-198, :445, :463/:466) without the split-K residual
// branch; xn is written in the XP layout that dec_gemm consumes.
// one WAVE per row (4 rows per workgroup): a decode-step row is 2.5 KB, so the kernel is pure latency — no LDS, no barrier,
// the sum of squares folds through wave shuffles in a fixed order.  Lane l owns column groups l, l+64, ... (4 columns each).
template <int NQ>
__global__ __launch_bounds__(256) void rmsnorm2_kernel(Norm2P p, int rows) {
    car_kernarg_prefetch<(sizeof(Norm2P) + 63 + 48) / 64>();
    if (p.pf_wgs > 0 && (int)blockIdx.x >= (int)gridDim.x - p.pf_wgs) {      // L2 run-ahead helper (whole workgroup; this kernel has no barrier)
        STAMP(p, 0);
        car_pf_helper(p.pf_p0, p.pf_b0, p.pf_p1, p.pf_b1, (int)blockIdx.x - ((int)gridDim.x - p.pf_wgs), p.pf_wgs);
        STAMP(p, 5);
        return;
    }
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    STAMP(p, 0);
    const bool hs_fresh = HS_FRESH(p), hs_wt = HS_WT(p);
    HS_WAIT(p);                                                            // early launch: the residual stream comes from the predecessor
    if (r < rows) {
    const int lane = threadIdx.x & 63;
    const int D = p.D, ng = D >> 2;
    const bf16_t* src = p.idx ? p.emb + (long)p.idx[r] * D : p.h_in + r * D;
    if (p.add & 2) __builtin_amdgcn_s_setprio(3);        // see dec_gemm_kernel
    const bf16_t* add = (p.add & 1) ? p.ctrl + (r * p.n_tok + (*p.pos - p.T + 1)) * D : nullptr;
    float val[NQ][4];
    float ss = 0.f;
    uint2 wreg[NQ];        // the norm weight does not depend on the reduction: requested up front, one round trip instead of two
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const int gi = lane + q * 64; if (gi < ng) wreg[q] = *(const uint2*)(p.w + gi * 4); }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int gi = lane + q * 64;
        if (gi < ng) {
            const uint2 u = car_ld8(src + gi * 4, hs_fresh && p.idx == nullptr);
            float v[4] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
            if (add) {
                const uint2 a = *(const uint2*)(add + gi * 4);
                const float c[4] = {__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xffff0000u), __uint_as_float(a.y << 16), __uint_as_float(a.y & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = bf2f(f2bf(v[e] + bf2f(f2bf(p.cs * c[e]))));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { val[q][e] = v[e]; ss += v[e] * v[e]; }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / D + p.eps);
#ifdef CAR_STAMP
    asm volatile("s_nop 0" ::"v"(rstd)); STAMP(p, 3);
#endif
    const int nkb = D >> 5;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int gi = lane + q * 64;
        if (gi < ng) {
            const int k = gi * 4;
            if (p.h_out) { uint2 u; u.x = pack_bf16x2(val[q][0], val[q][1]); u.y = pack_bf16x2(val[q][2], val[q][3]); car_st8(p.h_out + r * D + k, u, hs_wt); }
            const uint2 wu = wreg[q];
            const float w[4] = {__uint_as_float(wu.x << 16), __uint_as_float(wu.x & 0xffff0000u), __uint_as_float(wu.y << 16), __uint_as_float(wu.y & 0xffff0000u)};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = bf2f(f2bf(val[q][e] * rstd)) * w[e];
            uint2 u; u.x = pack_bf16x2(o[0], o[1]); u.y = pack_bf16x2(o[2], o[3]);
            const long off = ((((r >> 4) * nkb + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (r & 15)) << 3) + (k & 7);
            car_st8(p.xn + off, u, hs_wt);
        }
    }
    }
    HS_ARRIVE(p);
#ifdef CAR_STAMP
    __builtin_amdgcn_s_waitcnt(0); STAMP(p, 5);
#endif
}
extern "C" void car_launch_rmsnorm2(const Norm2P* p, long rows, hipStream_t st) {
    const int nq = (p->D / 4 + 63) / 64;                      // column groups per lane (D <= 4096: every LlamaGen size)
    Norm2P q = *p;
    const unsigned main_wgs = (unsigned)((rows + 3) / 4);
    if (q.pf_wgs < 0 || (main_wgs & 7)) q.pf_wgs = 0;         // the helper's XCD arithmetic assumes the main grid is a multiple of 8
    const dim3 g(main_wgs + (unsigned)q.pf_wgs), b(256);
    if (nq <= 4) hipLaunchKernelGGL((rmsnorm2_kernel<4>), g, b, 0, st, q, (int)rows);
    else if (nq <= 8) hipLaunchKernelGGL((rmsnorm2_kernel<8>), g, b, 0, st, q, (int)rows);
    else hipLaunchKernelGGL((rmsnorm2_kernel<16>), g, b, 0, st, q, (int)rows);
}
