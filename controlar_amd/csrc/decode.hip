// decode.hip — the HBM-bound kernels of the one-token decode step.
//
// dec_attn: RoPE(q,k) + KV-cache write + single-query attention over the valid prefix of the
// cache, for one (sequence, head, kv-split) per workgroup.
//   reference: gpt_t2i.py:264-286 (wqkv split, apply_rotary_emb :522-532, KVCache.update :227-235,
//   scaled_dot_product_attention over all S_max slots with the bool mask row built at
//   generate.py:184-193).  Masked slots contribute exactly 0 after softmax, so attending over the
//   valid length p+1 is exact.
// Layout: K/V cache [b, head, S_max, 64] of T — one (b, head) stream is contiguous, so a wave
// instruction reads 1 KiB of consecutive rows with 16 B per lane (fully coalesced).
#include "car_common.h"
#include "kernel_params.h"

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <typename T> struct VecT;
template <> struct VecT<bf16_t> { static constexpr int EPL = 8; };   // elements per 16-byte lane load
template <> struct VecT<float>  { static constexpr int EPL = 4; };

template <typename T>
__device__ inline void load16(const T* p, float (&v)[VecT<T>::EPL]);
template <>
__device__ inline void load16<bf16_t>(const bf16_t* p, float (&v)[8]) {
    // KV rows are read once per step and the cache (tens of GB) never fits on chip: non-temporal, keep L2/MALL for weights
    const u32x4 u = __builtin_nontemporal_load((const u32x4*)p);
    const unsigned w[4] = {u[0], u[1], u[2], u[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
template <>
__device__ inline void load16<float>(const float* p, float (&v)[4]) {
    const float4 u = *(const float4*)p; v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
}

template <typename T>
__global__ __launch_bounds__(256) void dec_attn_kernel(AttnP p) {
    constexpr int EPL = VecT<T>::EPL;        // 8 (bf16) / 4 (fp32)
    constexpr int LPR = 64 / EPL;            // lanes per cache row: 8 / 16
    constexpr int RPI = 64 / LPR;            // rows per wave instruction: 8 / 4
    __shared__ float sq[64], sk[64], sv[64];
    __shared__ float red[4][RPI > 8 ? RPI : 8][66];   // per wave, per row-group: m, l, o[64]  (reused for the wave merge)
    const int h = blockIdx.x, b = blockIdx.y, split = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pos = *p.pos;
    const T* qkv = (const T*)p.qkv + (long)b * 3 * p.dim;
    T* kc = (T*)p.kcache + ((long)b * p.H + h) * p.S_max * 64;
    T* vc = (T*)p.vcache + ((long)b * p.H + h) * p.S_max * 64;

    // ---- RoPE on q and k (fp32 rotate, round to T), v passthrough; new k/v go to the cache
    auto ldqkv = [&](int col) -> float {
        if (!p.qkv_parts) return ET<T>::ld(qkv + col);
        float a = 0.f;
        for (int s = 0; s < p.qkv_ks; ++s) a += p.qkv_parts[s * p.qkv_stride + (long)b * 3 * p.dim + col];
        return ET<T>::rnd(a);         // the wqkv Linear output is rounded to T
    };
    if (tid < 32) {
        const float cs = p.rope[((long)pos * 32 + tid) * 2], sn = p.rope[((long)pos * 32 + tid) * 2 + 1];
        const float q0 = ldqkv(h * 64 + 2 * tid), q1 = ldqkv(h * 64 + 2 * tid + 1);
        const float k0 = ldqkv(p.dim + h * 64 + 2 * tid), k1 = ldqkv(p.dim + h * 64 + 2 * tid + 1);
        sq[2 * tid] = ET<T>::rnd(q0 * cs - q1 * sn); sq[2 * tid + 1] = ET<T>::rnd(q1 * cs + q0 * sn);
        sk[2 * tid] = ET<T>::rnd(k0 * cs - k1 * sn); sk[2 * tid + 1] = ET<T>::rnd(k1 * cs + k0 * sn);
    } else if (tid >= 64 && tid < 128) {
        sv[tid - 64] = ldqkv(2 * p.dim + h * 64 + (tid - 64));
    }
    __syncthreads();
    if (split == 0 && tid < 64) {
        ET<T>::st(kc + (long)pos * 64 + tid, sk[tid]);
        ET<T>::st(vc + (long)pos * 64 + tid, sv[tid]);
    }

    // ---- cached positions j in [j0, j1) of this split, strided over waves and row groups
    const int per = (pos + p.nsplit - 1) / p.nsplit;            // positions 0..pos-1 are in the cache
    const int j0 = split * per, j1 = min(pos, j0 + per);
    const int grp = lane / LPR, sub = lane % LPR;               // row group within the instruction, 16-byte chunk within the row
    float q[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) q[e] = sq[sub * EPL + e] * 0.125f;   // head_dim^-0.5 = 1/8 exactly
    const unsigned char* mk = p.emb_mask ? p.emb_mask + (long)b * p.T : nullptr;

    float m = -INFINITY, l = 0.f, o[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = 0.f;

    constexpr int UNR = 4;
    for (int base = j0 + wave * RPI * UNR; base < j1; base += 4 * RPI * UNR) {
        float kv[UNR][EPL], vv[UNR][EPL]; bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int j = base + u * RPI + grp;
            ok[u] = j < j1 && !(mk && j < p.T && !mk[j]);
            if (ok[u]) { load16<T>(kc + (long)j * 64 + sub * EPL, kv[u]); load16<T>(vc + (long)j * 64 + sub * EPL, vv[u]); }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float s = 0.f;
            if (ok[u]) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) s = fmaf(q[e], kv[u][e], s);
            }
#pragma unroll
            for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off, 64);
            if (ok[u]) {
                const float mn = fmaxf(m, s), a = expf(m - mn), w = expf(s - mn);
                l = l * a + w;
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[e] = fmaf(o[e], a, w * vv[u][e]);
                m = mn;
            }
        }
    }
    // ---- the new token itself (always allowed: diagonal forced on), handled by split 0 / wave 0 / group 0
    if (split == 0 && wave == 0 && grp == 0) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) s = fmaf(q[e], sk[sub * EPL + e], s);
#pragma unroll
        for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off, LPR);
        const float mn = fmaxf(m, s), a = expf(m - mn), w = expf(s - mn);
        l = l * a + w;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = fmaf(o[e], a, w * sv[sub * EPL + e]);
        m = mn;
    }
    // ---- merge the (m, l, o) states: row groups -> LDS, then one wave folds all 4*RPI states in fixed order
    if (sub == 0) { red[wave][grp][0] = m; red[wave][grp][1] = l; }
#pragma unroll
    for (int e = 0; e < EPL; ++e) red[wave][grp][2 + sub * EPL + e] = o[e];
    __syncthreads();
    if (tid < 64) {
        float M = -INFINITY;
        for (int w = 0; w < 4; ++w) for (int g = 0; g < RPI; ++g) M = fmaxf(M, red[w][g][0]);
        float L = 0.f, O = 0.f;
        for (int w = 0; w < 4; ++w) for (int g = 0; g < RPI; ++g) {
            const float mm = red[w][g][0];
            if (mm > -INFINITY) { const float a = expf(mm - M); L += red[w][g][1] * a; O += red[w][g][2 + tid] * a; }
        }
        if (p.nsplit == 1) {
            ET<T>::st((T*)p.out + (long)b * p.dim + h * 64 + tid, O / L);
        } else {
            float* pt = p.part + (((long)b * p.H + h) * p.nsplit + split) * 66;
            if (tid == 0) { pt[0] = M; pt[1] = L; }
            pt[2 + tid] = O;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(64) void dec_attn_combine_kernel(const float* part, void* out, int H, int nsplit, int dim) {
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* pt = part + ((long)b * H + h) * nsplit * 66;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, pt[s * 66]);
    float L = 0.f, O = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float mm = pt[s * 66];
        if (mm > -INFINITY) { const float a = expf(mm - M); L += pt[s * 66 + 1] * a; O += pt[s * 66 + 2 + d] * a; }
    }
    ET<T>::st((T*)out + (long)b * dim + h * 64 + d, O / L);
}

extern "C" void car_launch_dec_attn(int mode, const AttnP* p, int b, hipStream_t st) {
    dim3 g(p->H, b, p->nsplit);
    if (mode == 1) hipLaunchKernelGGL(dec_attn_kernel<bf16_t>, g, dim3(256), 0, st, *p);
    else hipLaunchKernelGGL(dec_attn_kernel<float>, g, dim3(256), 0, st, *p);
    if (p->nsplit > 1 && p->out) {        // out == nullptr: the consumer (dec_linear xmode 2) combines the splits itself
        if (mode == 1) hipLaunchKernelGGL(dec_attn_combine_kernel<bf16_t>, dim3(p->H, b), dim3(64), 0, st, p->part, p->out, p->H, p->nsplit, p->dim);
        else hipLaunchKernelGGL(dec_attn_combine_kernel<float>, dim3(p->H, b), dim3(64), 0, st, p->part, p->out, p->H, p->nsplit, p->dim);
    }
}

// ---------------------------------------------------------------------------------------------
// Prefill KV scatter: k/v of the T prefix rows -> cache, with RoPE on q,k in place.
// qkv [b*T, 3*dim] T;  writes q (rotated) back in place, K/V into cache [b,H,S_max,64] at rows 0..T-1.
// reference: gpt_t2i.py:266-277.  Prefix RoPE rows are zero, so rotated q,k of the prefix are 0 —
// reproduced by arithmetic, not special-cased.
template <typename T>
__global__ void prefill_rope_kv_kernel(void* qkv_, void* kcache, void* vcache, const float* rope, int b, int Tn, int H, int dim, int S_max) {
    const long total = (long)b * Tn * H * 32;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int pr = (int)(i % 32); const int h = (int)((i / 32) % H); const int t = (int)((i / (32L * H)) % Tn); const long bb = i / (32L * H * Tn);
        T* row = (T*)qkv_ + (bb * Tn + t) * 3 * dim;
        const float cs = rope[((long)t * 32 + pr) * 2], sn = rope[((long)t * 32 + pr) * 2 + 1];
        const float q0 = ET<T>::ld(row + h * 64 + 2 * pr), q1 = ET<T>::ld(row + h * 64 + 2 * pr + 1);
        const float k0 = ET<T>::ld(row + dim + h * 64 + 2 * pr), k1 = ET<T>::ld(row + dim + h * 64 + 2 * pr + 1);
        ET<T>::st(row + h * 64 + 2 * pr, q0 * cs - q1 * sn); ET<T>::st(row + h * 64 + 2 * pr + 1, q1 * cs + q0 * sn);
        const float kr0 = ET<T>::rnd(k0 * cs - k1 * sn), kr1 = ET<T>::rnd(k1 * cs + k0 * sn);
        ET<T>::st(row + dim + h * 64 + 2 * pr, kr0); ET<T>::st(row + dim + h * 64 + 2 * pr + 1, kr1);
        T* kc = (T*)kcache + ((bb * H + h) * S_max + t) * 64; T* vc = (T*)vcache + ((bb * H + h) * S_max + t) * 64;
        ET<T>::st(kc + 2 * pr, kr0); ET<T>::st(kc + 2 * pr + 1, kr1);
        vc[2 * pr] = row[2 * dim + h * 64 + 2 * pr]; vc[2 * pr + 1] = row[2 * dim + h * 64 + 2 * pr + 1];
    }
}
extern "C" void car_launch_prefill_rope_kv(int mode, void* qkv, void* kc, void* vc, const float* rope, int b, int Tn, int H, int dim, int S_max, hipStream_t st) {
    long total = (long)b * Tn * H * 32; int g = (int)((total + 255) / 256); if (g > 4096) g = 4096;
    if (mode == 1) hipLaunchKernelGGL(prefill_rope_kv_kernel<bf16_t>, dim3(g), dim3(256), 0, st, qkv, kc, vc, rope, b, Tn, H, dim, S_max);
    else hipLaunchKernelGGL(prefill_rope_kv_kernel<float>, dim3(g), dim3(256), 0, st, qkv, kc, vc, rope, b, Tn, H, dim, S_max);
}

