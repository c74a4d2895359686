// ops.hip — bandwidth-bound helper kernels of the path: norms, softmax, resize/patchify,
// GroupNorm+swish on NHWC, codebook lookup, final RGB conv, embedding gathers, greedy sampling.
// Kernels are templated on the element type T (float = exact mode, bf16_t = fast mode), take
// untyped pointers, reduce in fp32 with wave64 shuffles in a fixed order (deterministic), and
// round where the reference's torch ops round.
#include "car_common.h"
#include "kernel_params.h"

#define LAUNCH_T(mode, KERNEL, grid, block, st, ...)                                          \
    do {                                                                                      \
        if ((mode) == 1) hipLaunchKernelGGL((KERNEL<bf16_t>), grid, block, 0, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<float>), grid, block, 0, st, __VA_ARGS__);            \
    } while (0)

__device__ __forceinline__ void ld8bf_ops(const bf16_t* p, float (&v)[8]) {
    const uint4 u = *(const uint4*)p; const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
}

// ------------------------------------------------------------------ dtype conversion (inputs arrive as f32 or bf16)
template <typename T>
__global__ void convert_kernel(const void* src, int src_dtype, void* dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float v = src_dtype == 1 ? bf2f(((const bf16_t*)src)[i]) : ((const float*)src)[i];
        ET<T>::st((T*)dst + i, v);
    }
}
extern "C" void car_launch_convert(int mode, const void* src, int src_dtype, void* dst, long n, hipStream_t st) {
    int g = (int)((n + 255) / 256); if (g > 4096) g = 4096; if (g < 1) g = 1;
    LAUNCH_T(mode, convert_kernel, dim3(g), dim3(256), st, src, src_dtype, dst, n);
}

// text input: rows [0,B) = cond * 1 (already masked by the caller), rows [B,2B) = uncond_embedding
// (reference generate.py:156-158)
template <typename T>
// `per` elements per sequence are written; they are the window [src_off, src_off + per) of the sequence's `per_src` source elements (the prefill keeps only the
// last rows of the left-padded prefix: engine_generate.hip, "prefill window")
__global__ void build_text_kernel(const void* cond, int src_dtype, const void* uncond, void* dst, int B, long per, long per_src, long src_off, int use_cfg) {
    const long n = (long)(use_cfg ? 2 * B : B) * per;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const long b = i / per, r = i - b * per;
        float v;
        if (b < B) { const long si = b * per_src + src_off + r; v = src_dtype == 1 ? bf2f(((const bf16_t*)cond)[si]) : ((const float*)cond)[si]; }
        else v = ET<T>::rnd(0.f + ET<T>::ld((const T*)uncond + src_off + r));
        ET<T>::st((T*)dst + i, v);
    }
}
extern "C" void car_launch_build_text(int mode, const void* cond, int src_dtype, const void* uncond, void* dst, int B, long per, long per_src, long src_off, int use_cfg, hipStream_t st) {
    LAUNCH_T(mode, build_text_kernel, dim3(2048), dim3(256), st, cond, src_dtype, uncond, dst, B, per, per_src, src_off, use_cfg);
}

// out[r, :] = table[idx[r], :]   (LabelEmbedder / tok_embeddings gather)
template <typename T>
__global__ void gather_rows_kernel(const void* table, const int* idx, void* out, long rows, int D) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x, n = rows * D;
    for (; i < n; i += stride) { const long r = i / D; const int d = (int)(i - r * D); ((T*)out)[i] = ((const T*)table)[(long)idx[r] * D + d]; }
}
// c2i prefix: table row of every decode row = its image's class label, or the null class `num_classes` for the unconditional CFG half
// (generate.py:141, gpt.py:89-96).  Out-of-range labels: clamped to the null class + sticky error flag (reported by car_get_stats).
__global__ void label_index_kernel(const int64_t* labels, const int* row_img, const int* row_unc, int num_classes, int* idx, int* err_flag, int b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b) return;
    int64_t l = row_unc[i] ? (int64_t)num_classes : labels[row_img[i]];
    if (l < 0 || l > num_classes) { __hip_atomic_store(err_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); l = num_classes; }   // host-mapped sticky flag (engine_internal.h check_sticky)
    idx[i] = (int)l;
}
extern "C" void car_launch_label_index(const int64_t* labels, const int* row_img, const int* row_unc, int num_classes, int* idx, int* err_flag, int b, hipStream_t st) {
    hipLaunchKernelGGL(label_index_kernel, dim3((b + 255) / 256), dim3(256), 0, st, labels, row_img, row_unc, num_classes, idx, err_flag, b);
}

extern "C" void car_launch_gather_rows(int mode, const void* table, const int* idx, void* out, long rows, int D, hipStream_t st) {
    long n = rows * D; int g = (int)((n + 255) / 256); if (g > 2048) g = 2048;
    LAUNCH_T(mode, gather_rows_kernel, dim3(g), dim3(256), st, table, idx, out, rows, D);
}

// ------------------------------------------------------------------ LayerNorm (HF Dinov2Layer norm1/norm2/layernorm, eps 1e-6)
template <typename T>
__global__ __launch_bounds__(128) void layernorm_kernel(const void* x_, const void* w_, const void* b_, void* y_, int D, float eps) {
    __shared__ float sm[20];
    const T* x = (const T*)x_ + (long)blockIdx.x * D; T* y = (T*)y_ + (long)blockIdx.x * D;
    const T* w = (const T*)w_; const T* b = (const T*)b_;
    float s = 0.f;
    for (int i = threadIdx.x; i < D; i += blockDim.x) s += ET<T>::ld(x + i);
    const float mean = block_sum(s, sm) / D;
    float v = 0.f;
    for (int i = threadIdx.x; i < D; i += blockDim.x) { float d = ET<T>::ld(x + i) - mean; v += d * d; }
    const float rstd = rsqrtf(block_sum(v, sm) / D + eps);
    for (int i = threadIdx.x; i < D; i += blockDim.x)
        ET<T>::st(y + i, (ET<T>::ld(x + i) - mean) * rstd * ET<T>::ld(w + i) + ET<T>::ld(b + i));
}
extern "C" void car_launch_layernorm(int mode, const void* x, const void* w, const void* b, void* y, long rows, int D, float eps, hipStream_t st) {
    LAUNCH_T(mode, layernorm_kernel, dim3(rows), dim3(128), st, x, w, b, y, D, eps);
}

template <typename T>
__device__ inline void ld4(const T* p, float (&v)[4]);
template <> __device__ inline void ld4<float>(const float* p, float (&v)[4]) { const float4 u = *(const float4*)p; v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w; }
template <> __device__ inline void ld4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    const uint2 u = *(const uint2*)p;
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u); v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
}
template <typename T>
__device__ inline void st4(T* p, const float (&v)[4]);
template <> __device__ inline void st4<float>(float* p, const float (&v)[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
template <> __device__ inline void st4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    uint2 u; u.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16); u.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    *(uint2*)p = u;
}

// one block per row; each thread owns groups of 4 consecutive columns (D % 4 == 0), values stay in registers
// (up to 4 groups per thread: D <= 16 * blockDim)
template <typename T>
__global__ __launch_bounds__(1024) void rmsnorm_kernel(NormP p) {
    __shared__ float sm[20];
    const long r = blockIdx.x;
    const int D = p.D, ng = D >> 2;
    const T* src = p.idx ? (const T*)p.emb + (long)p.idx[r] * D : (const T*)p.h_in + r * D;
    const T* add = nullptr;
    if (p.add_mode == 1) add = (const T*)p.ctrl + (r * p.n_tok + (*p.pos - p.T + 1)) * D;
    else if (p.add_mode == 2 && (r % p.T) == p.T - 1) add = (const T*)p.ctrl + ((r / p.T) * (long)p.n_tok) * D;
    float val[4][4];
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int gi = threadIdx.x + q * blockDim.x;
        if (gi < ng) {
            float v[4]; ld4<T>(src + gi * 4, v);
            if (p.parts) {      // h = h + linear_out: the linear output (sum of split-K slices) is rounded to T first
                float a[4] = {0.f, 0.f, 0.f, 0.f};
                const float* pp = p.parts + r * D + gi * 4;
                int s = 0;
                for (; s + 4 <= p.parts_ks; s += 4) {
                    const float4 a0 = *(const float4*)(pp + (s + 0) * p.parts_stride), a1 = *(const float4*)(pp + (s + 1) * p.parts_stride);
                    const float4 a2 = *(const float4*)(pp + (s + 2) * p.parts_stride), a3 = *(const float4*)(pp + (s + 3) * p.parts_stride);
                    a[0] += a0.x; a[1] += a0.y; a[2] += a0.z; a[3] += a0.w;  a[0] += a1.x; a[1] += a1.y; a[2] += a1.z; a[3] += a1.w;
                    a[0] += a2.x; a[1] += a2.y; a[2] += a2.z; a[3] += a2.w;  a[0] += a3.x; a[1] += a3.y; a[2] += a3.z; a[3] += a3.w;
                }
                for (; s < p.parts_ks; ++s) { const float4 a0 = *(const float4*)(pp + s * p.parts_stride); a[0] += a0.x; a[1] += a0.y; a[2] += a0.z; a[3] += a0.w; }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ET<T>::rnd(v[e] + ET<T>::rnd(a[e]));
            }
            if (add) {
                float c[4]; ld4<T>(add + gi * 4, c);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ET<T>::rnd(v[e] + ET<T>::rnd(p.cs * c[e]));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { val[q][e] = v[e]; ss += v[e] * v[e]; }
        }
    }
    if (!p.xn) {        // gather / control-add only (exact-mode decode, round 5: the norm itself runs inside the consuming GEMM, decode_f32.hip NX)
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int gi = threadIdx.x + q * blockDim.x; if (gi < ng && p.h_out) st4<T>((T*)p.h_out + r * D + gi * 4, val[q]); }
        return;
    }
    const float rstd = rsqrtf(block_sum(ss, sm) / D + p.eps);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int gi = threadIdx.x + q * blockDim.x;
        if (gi < ng) {
            if (p.h_out) st4<T>((T*)p.h_out + r * D + gi * 4, val[q]);
            float w[4], o[4]; ld4<T>((const T*)p.w + gi * 4, w);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ET<T>::rnd(val[q][e] * rstd) * w[e];
            st4<T>((T*)p.xn + r * D + gi * 4, o);
        }
    }
}
extern "C" void car_launch_rmsnorm(int mode, const NormP* p, long rows, hipStream_t st) {
    int ng = p->D / 4, th = ((ng + 63) / 64) * 64; if (th > 1024) th = 1024; if (th < 64) th = 64;
    if (mode == 1) hipLaunchKernelGGL(rmsnorm_kernel<bf16_t>, dim3(rows), dim3(th), 0, st, *p);
    else hipLaunchKernelGGL(rmsnorm_kernel<float>, dim3(rows), dim3(th), 0, st, *p);
}

// ------------------------------------------------------------------ row softmax  S fp32 [rows, lds] -> P T [rows, ldp] (zero padded)
// mask_mode 0: none (ViT, VQ attention).  mask_mode 1: LlamaGen prefill mask (generate.py:184-193):
//   rows ordered (b, head, i); key j allowed iff j <= i and (emb_mask[b, j] != 0 or j == i).
template <typename T>
__global__ __launch_bounds__(256) void softmax_kernel(const float* S, long lds, void* P_, long ldp, int ncols, int mask_mode,
                                                      const unsigned char* emb_mask, int Tq, int n_head) {
    __shared__ float sm[20];
    const long r = blockIdx.x;
    const float* s = S + r * lds; T* P = (T*)P_ + r * ldp;
    const int i = mask_mode ? (int)(r % Tq) : 0;
    const unsigned char* mk = mask_mode ? emb_mask + (r / Tq / n_head) * Tq : nullptr;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < ncols; j += blockDim.x) {
        bool ok = !mask_mode || (j <= i && (mk[j] || j == i));
        if (ok) mx = fmaxf(mx, s[j]);
    }
    mx = block_max(mx, sm);
    float sum = 0.f;
    for (int j = threadIdx.x; j < ncols; j += blockDim.x) {
        bool ok = !mask_mode || (j <= i && (mk[j] || j == i));
        if (ok) sum += expf(s[j] - mx);
    }
    sum = block_sum(sum, sm);
    const float inv = 1.0f / sum;
    for (int j = threadIdx.x; j < ldp; j += blockDim.x) {
        float v = 0.f;
        if (j < ncols) { bool ok = !mask_mode || (j <= i && (mk[j] || j == i)); if (ok) v = expf(s[j] - mx) * inv; }
        ET<T>::st(P + j, v);
    }
}
// Single pass, the row in registers (rows of up to 8 x 256 columns: the ViT encoder's 1025, the VQ AttnBlock's 1024): S is read once instead of three times.
// Thread t owns columns t, t + 256, ... and sums them in that order, then block_sum — the statements of softmax_kernel, hence its bits.
template <typename T>
__global__ __launch_bounds__(256) void softmax_reg_kernel(const float* S, long lds, void* P_, long ldp, int ncols, int mask_mode,
                                                          const unsigned char* emb_mask, int Tq, int n_head) {
    __shared__ float sm[20];
    const long r = blockIdx.x;
    const float* s = S + r * lds; T* P = (T*)P_ + r * ldp;
    const int i = mask_mode ? (int)(r % Tq) : 0;
    const unsigned char* mk = mask_mode ? emb_mask + (r / Tq / n_head) * Tq : nullptr;
    float v[8]; bool ok[8];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int j = threadIdx.x + k * 256;
        ok[k] = j < ncols && (!mask_mode || (j <= i && (mk[j] || j == i)));
        v[k] = ok[k] ? s[j] : 0.f;
        if (ok[k]) mx = fmaxf(mx, v[k]);
    }
    mx = block_max(mx, sm);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = ok[k] ? expf(v[k] - mx) : 0.f; if (ok[k]) sum += v[k]; }
    sum = block_sum(sum, sm);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int j = threadIdx.x + k * 256; if (j < ldp) ET<T>::st(P + j, ok[k] ? v[k] * inv : 0.f); }
}

// Rows of at most 128 columns (the prefill's Tv x Tv scores): one WAVE per row, and a column is summed in the lane of its ABSOLUTE position col0 + j — lane
// (col0 + j) % 64, the two halves of a lane in position order, then a fixed butterfly.  A masked column adds an exact zero, so a valid row's probabilities are the
// same bits whatever window [col0, col0 + ncols) of the prefix the prefill runs on (engine_generate.hip "prefill window": the batch-mates decide col0).
template <typename T>
__global__ __launch_bounds__(256) void softmax_wave_kernel(const float* S, long lds, void* P_, long ldp, long rows, int ncols, int mask_mode,
                                                           const unsigned char* emb_mask, int Tq, int n_head, int col0) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* s = S + r * lds; T* P = (T*)P_ + r * ldp;
    const int i = mask_mode ? (int)(r % Tq) : 0;
    const unsigned char* mk = mask_mode ? emb_mask + (r / Tq / n_head) * Tq : nullptr;
    float v[2]; bool ok[2]; int jj[2];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int j = lane + 64 * k - col0; jj[k] = j;
        ok[k] = j >= 0 && j < ncols && (!mask_mode || (j <= i && (mk[j] || j == i)));
        v[k] = ok[k] ? s[j] : 0.f;
        if (ok[k]) mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
#pragma unroll
    for (int k = 0; k < 2; ++k) v[k] = ok[k] ? expf(v[k] - mx) : 0.f;
    const float sum = wave_sum(v[0] + v[1]);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int k = 0; k < 2; ++k) if (jj[k] >= 0 && jj[k] < ncols) ET<T>::st(P + jj[k], v[k] * inv);
    for (int j = ncols + lane; j < ldp; j += 64) ET<T>::st(P + j, 0.f);      // zero padding of the row
}

// `col0`: absolute position of column 0 (the prefill window's start; 0 elsewhere)
extern "C" void car_launch_softmax_at(int mode, const float* S, long lds, void* P, long ldp, long rows, int ncols, int mask_mode,
                                      const unsigned char* emb_mask, int Tq, int n_head, int col0, hipStream_t st) {
    if (col0 >= 0 && col0 + ncols <= 128) {
        const dim3 g((unsigned)((rows + 3) / 4));
        if (mode == 1) hipLaunchKernelGGL(softmax_wave_kernel<bf16_t>, g, dim3(256), 0, st, S, lds, P, ldp, rows, ncols, mask_mode, emb_mask, Tq, n_head, col0);
        else hipLaunchKernelGGL(softmax_wave_kernel<float>, g, dim3(256), 0, st, S, lds, P, ldp, rows, ncols, mask_mode, emb_mask, Tq, n_head, col0);
        return;
    }
    if (ncols <= 2048 && ldp <= 2048) { LAUNCH_T(mode, softmax_reg_kernel, dim3(rows), dim3(256), st, S, lds, P, ldp, ncols, mask_mode, emb_mask, Tq, n_head); return; }
    LAUNCH_T(mode, softmax_kernel, dim3(rows), dim3(256), st, S, lds, P, ldp, ncols, mask_mode, emb_mask, Tq, n_head);
}
extern "C" void car_launch_softmax(int mode, const float* S, long lds, void* P, long ldp, long rows, int ncols, int mask_mode,
                                   const unsigned char* emb_mask, int Tq, int n_head, hipStream_t st) {
    car_launch_softmax_at(mode, S, lds, P, ldp, rows, ncols, mask_mode, emb_mask, Tq, n_head, ncols <= 128 ? 0 : -1, st);
}

// ------------------------------------------------------------------ resize + patch unfold (dinov2_adapter.py:16-24 + HF patch conv as matmul)
// out[b, gy*gw+gx, c*p*p + py*p + px] = resized[b, c, gy*p+py, gx*p+px], zero padded to Kpad.
// nearest: iy/ix hold source indices.  bicubic: iy/ix hold 4 clamped indices per output index, wy/wx 4 weights.
template <typename T>
__global__ void patchify_kernel(const void* img, int img_dtype, void* out_, int B, int H, int W, int gh, int gw, int p, int Kpad,
                                int bicubic, const int* iy, const int* ix, const float* wy, const float* wx) {
    const long total = (long)B * gh * gw * Kpad;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    const int pp = p * p;
    for (; i < total; i += stride) {
        const int k = (int)(i % Kpad); const long t = i / Kpad;
        float v = 0.f;
        if (k < 3 * pp) {
            const int c = k / pp, py = (k % pp) / p, px = k % p;
            const int gx = (int)(t % gw), gy = (int)((t / gw) % gh), b = (int)(t / ((long)gw * gh));
            const int oy = gy * p + py, ox = gx * p + px;
            const long base = ((long)b * 3 + c) * H * W;
            auto ld = [&](int y, int x) -> float {
                return img_dtype == 1 ? bf2f(((const bf16_t*)img)[base + (long)y * W + x]) : ((const float*)img)[base + (long)y * W + x];
            };
            if (!bicubic) v = ld(iy[oy], ix[ox]);
            else {
                // x first, then y (ATen nested cubic_interp1d); fp32
                float acc = 0.f;
                for (int a = 0; a < 4; ++a) {
                    const int yy = iy[oy * 4 + a];
                    float row = 0.f;
                    for (int q = 0; q < 4; ++q) row += ld(yy, ix[ox * 4 + q]) * wx[ox * 4 + q];
                    acc += row * wy[oy * 4 + a];
                }
                v = img_dtype == 1 ? bf2f(f2bf(acc)) : acc;   // interpolate returns the input dtype
            }
        }
        ET<T>::st((T*)out_ + i, v);
    }
}
extern "C" void car_launch_patchify(int mode, const void* img, int img_dtype, void* out, int B, int H, int W, int gh, int gw, int p, int Kpad,
                                    int bicubic, const int* iy, const int* ix, const float* wy, const float* wx, hipStream_t st) {
    LAUNCH_T(mode, patchify_kernel, dim3(4096), dim3(256), st, img, img_dtype, out, B, H, W, gh, gw, p, Kpad, bicubic, iy, ix, wy, wx);
}

// h[b,0] = cls + pos[0];  h[b,1+t] = tok[b,t] + pos[1+t]      (HF Dinov2Embeddings.forward :97-113)
template <typename T>
__global__ void vit_assemble_kernel(const void* tok_, const void* cls_, const void* pos_, void* h_, int B, int n, int D) {
    const long total = (long)B * (n + 1) * D;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int d = (int)(i % D); const long r = i / D; const int t = (int)(r % (n + 1)); const long b = r / (n + 1);
        float v = t == 0 ? ET<T>::ld((const T*)cls_ + d) : ET<T>::ld((const T*)tok_ + (b * n + t - 1) * D + d);
        ET<T>::st((T*)h_ + i, v + ET<T>::ld((const T*)pos_ + (long)t * D + d));
    }
}
extern "C" void car_launch_vit_assemble(int mode, const void* tok, const void* cls, const void* pos, void* h, int B, int n, int D, hipStream_t st) {
    LAUNCH_T(mode, vit_assemble_kernel, dim3(2048), dim3(256), st, tok, cls, pos, h, B, n, D);
}

// ------------------------------------------------------------------ GroupNorm(32, eps) [+ swish] on NHWC (vq_model.py:355-363)
// stage 1: per (b, pixel-chunk) per-channel partial sums (coalesced full-row reads); stage 2: fixed-order
// reduction to (mean, rstd) per (b, group) in double; stage 3: apply.
#define GN_CHUNK 256   // pixels per block in stage 1
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const void* x_, float* part, int HW, int C) {
    extern __shared__ float sh[];            // [2][256] then per-channel accumulators [2][C]
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const T* x = (const T*)x_ + (long)b * HW * C;
    const int p0 = chunk * GN_CHUNK, p1 = min(HW, p0 + GN_CHUNK);
    // thread t owns channel (t % C) if C <= 256 else loops; pixels strided by 256 / C
    float* accS = sh; float* accQ = sh + C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) { accS[c] = 0.f; accQ[c] = 0.f; }
    __syncthreads();
    if (C <= 256) {
        const int c = threadIdx.x % C, slot = threadIdx.x / C, nslot = 256 / C;
        float s = 0.f, q = 0.f;
        for (int p = p0 + slot; p < p1; p += nslot) { float v = ET<T>::ld(x + (long)p * C + c); s += v; q += v * v; }
        // reduce over slots in fixed order via LDS staging
        float* stS = sh + 2 * C; float* stQ = stS + 256;
        stS[threadIdx.x] = s; stQ[threadIdx.x] = q;
        __syncthreads();
        if (slot == 0) {
            for (int k = 1; k < nslot; ++k) { s += stS[k * C + c]; q += stQ[k * C + c]; }
            accS[c] = s; accQ[c] = q;
        }
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float s = 0.f, q = 0.f;
            for (int p = p0; p < p1; ++p) { float v = ET<T>::ld(x + (long)p * C + c); s += v; q += v * v; }
            accS[c] = s; accQ[c] = q;
        }
    }
    __syncthreads();
    float* o = part + ((long)b * nchunk + chunk) * 2 * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) { o[c] = accS[c]; o[C + c] = accQ[c]; }
}
// one wave per (b, group): lanes stride over (chunk, channel-in-group) pairs, double accumulation, fixed-order wave reduce
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* part, float* stats, int nchunk, int C, int G, int HW, float eps) {
    const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
    const int cpg = C / G;
    double s = 0.0, q = 0.0;
    for (int i = lane; i < nchunk * cpg; i += 64) {
        const int k = i / cpg, c = g * cpg + (i - k * cpg);
        const float* o = part + ((long)b * nchunk + k) * 2 * C;
        s += o[c]; q += o[C + c];
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); q += __shfl_xor(q, off, 64); }
    if (lane == 0) {
        const double n = (double)HW * cpg, mean = s / n;
        double var = q / n - mean * mean; if (var < 0) var = 0;
        stats[((long)b * G + g) * 2] = (float)mean;
        stats[((long)b * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
template <typename T>
__global__ void gn_apply_kernel(const void* x_, const float* stats, const void* gamma_, const void* beta_, void* y_, int B, int HW, int C, int G, int swish) {
    const long total = (long)B * HW * C;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    const int cpg = C / G;
    for (; i < total; i += stride) {
        const int c = (int)(i % C); const long b = i / ((long)HW * C);
        const float* stt = stats + (b * G + c / cpg) * 2;
        float v = (ET<T>::ld((const T*)x_ + i) - stt[0]) * stt[1] * ET<T>::ld((const T*)gamma_ + c) + ET<T>::ld((const T*)beta_ + c);
        v = ET<T>::rnd(v);
        if (swish) v = v / (1.0f + expf(-v));
        ET<T>::st((T*)y_ + i, v);
    }
}
// bf16, C a power of two <= 2048: a thread owns 8 fixed channels (its group statistics, gamma and beta stay in registers) and walks
// pixels with 16-byte loads / stores; same fp32 op order as gn_apply_kernel ((x - mean) * rstd * gamma + beta, one rounding, swish).
__global__ __launch_bounds__(256) void gn_apply_vec_kernel(const bf16_t* __restrict__ x_, const float* __restrict__ stats, const bf16_t* __restrict__ gamma,
                                                           const bf16_t* __restrict__ beta, bf16_t* __restrict__ y_, int HW, int C, int G, int swish, int ppb) {
    const int b = blockIdx.y, cpv = C >> 3, nslot = 256 / cpv, v = threadIdx.x % cpv, slot = threadIdx.x / cpv, cpg = C / G;
    float mu[8], rs[8], gm[8], bt[8];
    ld8bf_ops(gamma + v * 8, gm); ld8bf_ops(beta + v * 8, bt);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float* stt = stats + ((long)b * G + (v * 8 + e) / cpg) * 2; mu[e] = stt[0]; rs[e] = stt[1]; }
    const bf16_t* x = x_ + (long)b * HW * C + v * 8; bf16_t* y = y_ + (long)b * HW * C + v * 8;
    const int p0 = blockIdx.x * ppb, p1 = min(HW, p0 + ppb);
    for (int p = p0 + slot; p < p1; p += nslot) {
        float xv[8]; ld8bf_ops(x + (long)p * C, xv);
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            float a = bf2f(f2bf((xv[e] - mu[e]) * rs[e] * gm[e] + bt[e])), c = bf2f(f2bf((xv[e + 1] - mu[e + 1]) * rs[e + 1] * gm[e + 1] + bt[e + 1]));
            if (swish) { a = a / (1.0f + __expf(-a)); c = c / (1.0f + __expf(-c)); }
            o[e >> 1] = (unsigned)f2bf(a) | ((unsigned)f2bf(c) << 16);
        }
        *(uint4*)(y + (long)p * C) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
// bf16, C a power of two <= 2048: 16-byte loads (8 channels per lane), the pixels of a chunk dealt round-robin to 256 / (C/8)
// lane slots, slot sums folded in fixed order through LDS.  Same output as gn_partial_kernel (per-chunk per-channel sum, sum-sq).
__global__ __launch_bounds__(256) void gn_partial_vec_kernel(const bf16_t* __restrict__ x_, float* __restrict__ part, int HW, int C) {
    extern __shared__ float sh[];            // [256][16]: 8 sums + 8 square sums per thread
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const bf16_t* x = x_ + (long)b * HW * C;
    const int p0 = chunk * GN_CHUNK, p1 = min(HW, p0 + GN_CHUNK);
    const int cpv = C >> 3, nslot = 256 / cpv, v = threadIdx.x % cpv, slot = threadIdx.x / cpv;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    for (int p = p0 + slot; p < p1; p += nslot) {
        const uint4 u = *(const uint4*)(x + (long)p * C + v * 8);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = __uint_as_float(w[i] << 16), c = __uint_as_float(w[i] & 0xffff0000u);
            s[2 * i] += a; q[2 * i] += a * a; s[2 * i + 1] += c; q[2 * i + 1] += c * c;
        }
    }
    float* mine = sh + threadIdx.x * 16;
#pragma unroll
    for (int e = 0; e < 8; ++e) { mine[e] = s[e]; mine[8 + e] = q[e]; }
    __syncthreads();
    if (slot == 0) {
        for (int k = 1; k < nslot; ++k) {
            const float* o = sh + (k * cpv + v) * 16;
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += o[e]; q[e] += o[8 + e]; }
        }
        float* o = part + ((long)b * nchunk + chunk) * 2 * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[v * 8 + e] = s[e]; o[C + v * 8 + e] = q[e]; }
    }
}

// have_part != 0: `part` already holds the stage-1 partials of x (written by the producing conv's epilogue, GemmP::gn_part: one 256-pixel chunk per
// 16x16 tile, the same [B][nchunk][2][C] layout): the read-only pass over x is skipped.
extern "C" void car_launch_groupnorm_ex(int mode, const void* x, const void* gamma, const void* beta, void* y, float* part, float* stats,
                                        int B, int HW, int C, int G, float eps, int swish, int have_part, hipStream_t st) {
    const int nchunk = (HW + GN_CHUNK - 1) / GN_CHUNK;
    const size_t shb = (2 * C + 512) * sizeof(float);
    if (have_part) { /* stage 1 done by the producer */ }
    else if (mode == 1 && C >= 8 && C <= 2048 && (C & (C - 1)) == 0)
        hipLaunchKernelGGL(gn_partial_vec_kernel, dim3(nchunk, B), dim3(256), (size_t)256 * 16 * 4, st, (const bf16_t*)x, part, HW, C);
    else if (mode == 1) hipLaunchKernelGGL(gn_partial_kernel<bf16_t>, dim3(nchunk, B), dim3(256), shb, st, x, part, HW, C);
    else hipLaunchKernelGGL(gn_partial_kernel<float>, dim3(nchunk, B), dim3(256), shb, st, x, part, HW, C);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, B), dim3(64), 0, st, part, stats, nchunk, C, G, HW, eps);
    if (!y) return;              // statistics only
    if (mode == 1 && C >= 8 && C <= 2048 && (C & (C - 1)) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && !CAR_KNOB("CAR_GN_SCALAR")) {
        const int nslot = 256 / (C >> 3); int ppb = nslot * 8; if (ppb > HW) ppb = HW;
        hipLaunchKernelGGL(gn_apply_vec_kernel, dim3((HW + ppb - 1) / ppb, B), dim3(256), 0, st, (const bf16_t*)x, stats, (const bf16_t*)gamma, (const bf16_t*)beta,
                           (bf16_t*)y, HW, C, G, swish, ppb);
        return;
    }
    long total = (long)B * HW * C; int g = (int)((total + 255) / 256); if (g > 8192) g = 8192;
    LAUNCH_T(mode, gn_apply_kernel, dim3(g), dim3(256), st, x, stats, gamma, beta, y, B, HW, C, G, swish);
}
extern "C" void car_launch_groupnorm(int mode, const void* x, const void* gamma, const void* beta, void* y, float* part, float* stats,
                                     int B, int HW, int C, int G, float eps, int swish, hipStream_t st) {
    car_launch_groupnorm_ex(mode, x, gamma, beta, y, part, stats, B, HW, C, G, eps, swish, 0, st);
}

// ------------------------------------------------------------------ codebook lookup + post_quant_conv (vq_model.py:262-277, :49)
// z[b,p,:] = Wpq[zc, cd] @ (cb[tok] / max(||cb[tok]||, 1e-12)) + bias     -> NHWC [B, hw, zc]
template <typename T>
__global__ void vq_lookup_kernel(const int* tok, const float* cb, const float* wpq, const float* bpq, void* z_, long npix, int cd, int zc, int ncode) {
    const long total = npix * zc;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c = (int)(i % zc); const long pix = i / zc;
        int t = tok[pix]; t = t < 0 ? 0 : (t >= ncode ? ncode - 1 : t);
        const float* e = cb + (long)t * cd;
        float nn = 0.f;
        for (int k = 0; k < cd; ++k) nn += e[k] * e[k];
        const float inv = 1.0f / fmaxf(sqrtf(nn), 1e-12f);
        float acc = 0.f;
        for (int k = 0; k < cd; ++k) acc = fmaf(wpq[c * cd + k], e[k] * inv, acc);
        ET<T>::st((T*)z_ + i, acc + bpq[c]);
    }
}
extern "C" void car_launch_vq_lookup(int mode, const int* tok, const float* cb, const float* wpq, const float* bpq, void* z, long npix, int cd, int zc, int ncode, hipStream_t st) {
    long total = npix * zc; int g = (int)((total + 255) / 256); if (g > 4096) g = 4096;
    LAUNCH_T(mode, vq_lookup_kernel, dim3(g), dim3(256), st, tok, cb, wpq, bpq, z, npix, cd, zc, ncode);
}

// ------------------------------------------------------------------ conv_out 3x3 C->3 on NHWC input, fp32 NCHW output (vq_model.py:168,194)
// weights packed [3][9][C]; one lane per output pixel, channels in the inner loop.
template <typename T> struct LdVec;
template <> struct LdVec<bf16_t> { static constexpr int N = 8;
    __device__ static inline void ld(const bf16_t* p, float (&v)[8]) {
        const uint4 u = *(const uint4*)p; const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); } } };
template <> struct LdVec<float> { static constexpr int N = 4;
    __device__ static inline void ld(const float* p, float (&v)[4]) { const float4 u = *(const float4*)p; v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w; } };

// one lane per output pixel; the 27*C weights sit in LDS as fp32 (wave-uniform reads broadcast); 16-byte activation loads
template <typename T>
__global__ __launch_bounds__(256) void conv_out_kernel(const void* x_, const void* w_, const float* bias, float* out, int B, int H, int W, int C) {
    extern __shared__ float wsm[];       // [3][9][C]
    for (int i = threadIdx.x; i < 27 * C; i += blockDim.x) wsm[i] = ET<T>::ld((const T*)w_ + i);
    __syncthreads();
    const long npix = (long)B * H * W;
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    const int x = (int)(pix % W), y = (int)((pix / W) % H); const long b = pix / ((long)W * H);
    const T* X = (const T*)x_;
    constexpr int VN = LdVec<T>::N;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const T* px = X + ((b * H + yy) * W + xx) * C;
        const float* w0 = wsm + (0 * 9 + tap) * C; const float* w1 = wsm + (1 * 9 + tap) * C; const float* w2 = wsm + (2 * 9 + tap) * C;
        for (int c = 0; c < C; c += VN) {
            float v[VN]; LdVec<T>::ld(px + c, v);
#pragma unroll
            for (int e = 0; e < VN; ++e) { a0 = fmaf(v[e], w0[c + e], a0); a1 = fmaf(v[e], w1[c + e], a1); a2 = fmaf(v[e], w2[c + e], a2); }
        }
    }
    const long hw = (long)H * W, o = b * 3 * hw + (long)y * W + x;
    out[o] = a0 + bias[0]; out[o + hw] = a1 + bias[1]; out[o + 2 * hw] = a2 + bias[2];
}
// bf16 fast path: the 3x3 C->3 convolution as ONE pass over the activation.  A workgroup owns a 16x16 output tile: for its
// 18x18 halo of input pixels it computes the 27 per-tap partial products  Y[p][tap*3+o] = sum_c x[p][c] w[o][tap][c]  on the
// matrix cores (A = 16 pixels x 32 channels straight from HBM, 16 B per lane; B = the 27 x C weights held in registers),
// parks Y (fp32) in LDS, then every thread sums the 9 taps of its own pixel.  The activation is read 1.27x (halo) instead
// of 9x through the caches, and the 3-wide output never occupies a 128-wide GEMM tile.
__global__ __launch_bounds__(256) void conv_out_mfma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ out, int B, int H, int W, int C) {
    __shared__ float ys[336 * 28];                     // 21 m-blocks of 16 halo pixels x 27 (+1 pad) partial products
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q4 = lane >> 4, c16 = lane & 15;
    const int tx0 = blockIdx.x * 16, ty0 = blockIdx.y * 16, b = blockIdx.z;
    const int nkb = C >> 5;                            // <= 8 (C <= 256)
    // B fragments: lane (q4, c16) holds w[n = nb*16 + c16][c = kb*32 + q4*8 .. +8], n = tap*3 + o (n >= 27: zero)
    bf16x8 wf[2][8];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int n = nb * 16 + c16;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kb < nkb && n < 27) { const int o = n % 3, tap = n / 3; v = *(const uint4*)(w + ((long)(o * 9 + tap)) * C + kb * 32 + q4 * 8); }
            wf[nb][kb] = *(const bf16x8*)&v;
        }
    }
    const bf16_t* xb = x + (long)b * H * W * C;
    for (int mb = wave; mb < 21; mb += 4) {
        const int p = mb * 16 + c16;                   // halo pixel of this lane's A row
        const int hy = p / 18, hx = p - hy * 18, yy = ty0 + hy - 1, xx = tx0 + hx - 1;
        const bool ok = p < 324 && yy >= 0 && yy < H && xx >= 0 && xx < W;      // zero padding of the conv input
        const bf16_t* px = xb + ((long)yy * W + xx) * C + q4 * 8;
        f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            if (kb < nkb) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ok) v = *(const uint4*)(px + kb * 32);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&v, wf[0][kb], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&v, wf[1][kb], a1, 0, 0, 0);
            }
        }
        // D[row = pixel q4*4 + r][col = n]: n = c16 (a0) and 16 + c16 (a1)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* d = ys + (mb * 16 + q4 * 4 + r) * 28;
            d[c16] = a0[r];
            if (c16 < 12) d[16 + c16] = a1[r];
        }
    }
    __syncthreads();
    const int ly = tid >> 4, lx = tid & 15, y = ty0 + ly, xo = tx0 + lx;
    if (y < H && xo < W) {
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* d = ys + ((ly + tap / 3) * 18 + lx + tap % 3) * 28 + tap * 3;
            o0 += d[0]; o1 += d[1]; o2 += d[2];
        }
        const long hw = (long)H * W, o = (long)b * 3 * hw + (long)y * W + xo;
        out[o] = o0 + bias[0]; out[o + hw] = o1 + bias[1]; out[o + 2 * hw] = o2 + bias[2];
    }
}

extern "C" void car_launch_conv_out(int mode, const void* x, const void* w, const float* bias, float* out, int B, int H, int W, int C, hipStream_t st) {
    if (mode == 1 && C % 32 == 0 && C <= 256) {
        hipLaunchKernelGGL(conv_out_mfma_kernel, dim3((W + 15) / 16, (H + 15) / 16, B), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, bias, out, B, H, W, C);
        return;
    }
    const long npix = (long)B * H * W;
    const size_t shb = (size_t)27 * C * sizeof(float);
    if (mode == 1) hipLaunchKernelGGL(conv_out_kernel<bf16_t>, dim3((npix + 255) / 256), dim3(256), shb, st, x, w, bias, out, B, H, W, C);
    else hipLaunchKernelGGL(conv_out_kernel<float>, dim3((npix + 255) / 256), dim3(256), shb, st, x, w, bias, out, B, H, W, C);
}

// ------------------------------------------------------------------ VQ encoder entry conv: fp32 NCHW image [B,3,H,W] -> NHWC T [B,H,W,Co]
// (vq_model.py:68,108 conv_in 3x3 pad 1).  weights [Co][27] in PyTorch order (ci*9 + ky*3 + kx); one lane per pixel, weights in LDS.
template <typename T>
__global__ __launch_bounds__(256) void conv_in3_kernel(const float* img, const void* w_, const void* b_, void* out_, int B, int H, int W, int Co) {
    extern __shared__ float wsm[];       // [Co][27] + [Co]
    for (int i = threadIdx.x; i < Co * 27; i += blockDim.x) wsm[i] = ET<T>::ld((const T*)w_ + i);
    for (int i = threadIdx.x; i < Co; i += blockDim.x) wsm[Co * 27 + i] = ET<T>::ld((const T*)b_ + i);
    __syncthreads();
    const long npix = (long)B * H * W, pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    const int x = (int)(pix % W), y = (int)((pix / W) % H); const long b = pix / ((long)W * H);
    float in[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            in[ci * 9 + t] = (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0.f : ET<T>::rnd(img[((b * 3 + ci) * H + yy) * W + xx]);
        }
    T* o = (T*)out_ + pix * Co;
    for (int co = 0; co < Co; ++co) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 27; ++k) a = fmaf(in[k], wsm[co * 27 + k], a);
        ET<T>::st(o + co, a + wsm[Co * 27 + co]);
    }
}
extern "C" void car_launch_conv_in3(int mode, const float* img, const void* w, const void* b, void* out, int B, int H, int W, int Co, hipStream_t st) {
    const long npix = (long)B * H * W; const size_t shb = (size_t)Co * 28 * sizeof(float);
    if (mode == 1) hipLaunchKernelGGL(conv_in3_kernel<bf16_t>, dim3((npix + 255) / 256), dim3(256), shb, st, img, w, b, out, B, H, W, Co);
    else hipLaunchKernelGGL(conv_in3_kernel<float>, dim3((npix + 255) / 256), dim3(256), shb, st, img, w, b, out, B, H, W, Co);
}

// ------------------------------------------------------------------ VectorQuantizer.forward arg-min (vq_model.py:216-232)
// z NHWC T [npix, cd] ; both z and the codebook are l2-normalised (F.normalize eps 1e-12);
// d_j = |z|^2 + |e_j|^2 - 2 z.e_j in fp32 ; token = first index of the minimum.  One block per pixel.
template <typename T>
__global__ __launch_bounds__(256) void vq_argmin_kernel(const void* z_, const float* cb, int* tok, int cd, int ncode) {
    __shared__ float smv[4]; __shared__ int smi[4];
    const long pix = blockIdx.x;
    float z[16]; float zn = 0.f;
    for (int k = 0; k < cd; ++k) { z[k] = ET<T>::ld((const T*)z_ + pix * cd + k); zn += z[k] * z[k]; }
    const float zi = 1.0f / fmaxf(sqrtf(zn), 1e-12f);
    float z2 = 0.f;
    for (int k = 0; k < cd; ++k) { z[k] *= zi; z2 += z[k] * z[k]; }
    float best = INFINITY; int bi = 0x7fffffff;
    for (int j = threadIdx.x; j < ncode; j += blockDim.x) {
        const float* e = cb + (long)j * cd;
        float en = 0.f;
        for (int k = 0; k < cd; ++k) en += e[k] * e[k];
        const float ei = 1.0f / fmaxf(sqrtf(en), 1e-12f);
        float e2 = 0.f, dot = 0.f;
        for (int k = 0; k < cd; ++k) { const float v = e[k] * ei; e2 += v * v; dot = fmaf(z[k], v, dot); }
        const float d = z2 + e2 - 2.0f * dot;
        if (d < best) { best = d; bi = j; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smv[w] = best; smi[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) if (smv[k] < best || (smv[k] == best && smi[k] < bi)) { best = smv[k]; bi = smi[k]; }
        tok[pix] = bi;
    }
}
extern "C" void car_launch_vq_argmin(int mode, const void* z, const float* cb, int* tok, long npix, int cd, int ncode, hipStream_t st) {
    if (mode == 1) hipLaunchKernelGGL(vq_argmin_kernel<bf16_t>, dim3(npix), dim3(256), 0, st, z, cb, tok, cd, ncode);
    else hipLaunchKernelGGL(vq_argmin_kernel<float>, dim3(npix), dim3(256), 0, st, z, cb, tok, cd, ncode);
}

// ------------------------------------------------------------------ exact-mode SwiGLU on the block-16 interleaved w1|w3 layout
template <typename T>
__global__ void swiglu_kernel(const void* in_, void* out_, long rows, int hidden) {
    const long total = rows * hidden;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c = (int)(i % hidden); const long r = i / hidden;
        const T* row = (const T*)in_ + r * 2 * hidden + (c >> 4) * 32 + (c & 15);
        const float a = ET<T>::ld(row), g = ET<T>::ld(row + 16);
        ET<T>::st((T*)out_ + i, ET<T>::rnd(silu_f(a)) * g);
    }
}
extern "C" void car_launch_swiglu(int mode, const void* in, void* out, long rows, int hidden, hipStream_t st) {
    long total = rows * hidden; int g = (int)((total + 255) / 256); if (g > 4096) g = 4096;
    LAUNCH_T(mode, swiglu_kernel, dim3(g), dim3(256), st, in, out, rows, hidden);
}

// decode fast path: mid[m][k] = rnd(rnd(silu(rnd(sum_s a))) * rnd(sum_s c)) from the w13 split-K partials [ks][b][2*hidden]
// (block-16 interleaved w1|w3 columns); 8 hidden units per thread, 16-byte bf16 store.  gpt_t2i.py:217 rounding points.
__global__ __launch_bounds__(256) void swiglu_parts_kernel(const float* parts, int ks, long stride, bf16_t* out, int rows, int hidden) {
    const int gpr = hidden >> 3;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)rows * gpr) return;
    const int m = (int)(gid / gpr), k = (int)(gid - (long)m * gpr) * 8, col = (k >> 4) * 32 + (k & 15);
    float a[8], g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = 0.f; g[e] = 0.f; }
    for (int s = 0; s < ks; ++s) {
        const float* src = parts + s * stride + (long)m * 2 * hidden + col;
        const float4 a0 = *(const float4*)src, a1 = *(const float4*)(src + 4), g0 = *(const float4*)(src + 16), g1 = *(const float4*)(src + 20);
        a[0] += a0.x; a[1] += a0.y; a[2] += a0.z; a[3] += a0.w; a[4] += a1.x; a[5] += a1.y; a[6] += a1.z; a[7] += a1.w;
        g[0] += g0.x; g[1] += g0.y; g[2] += g0.z; g[3] += g0.w; g[4] += g1.x; g[5] += g1.y; g[6] += g1.z; g[7] += g1.w;
    }
    unsigned o[4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const float s0 = bf2f(f2bf(silu_f(bf2f(f2bf(a[e]))))) * bf2f(f2bf(g[e]));
        const float s1 = bf2f(f2bf(silu_f(bf2f(f2bf(a[e + 1]))))) * bf2f(f2bf(g[e + 1]));
        o[e >> 1] = (unsigned)f2bf(s0) | ((unsigned)f2bf(s1) << 16);
    }
    *(uint4*)(out + (long)m * hidden + k) = make_uint4(o[0], o[1], o[2], o[3]);
}
extern "C" void car_launch_swiglu_parts(const float* parts, int ks, long stride, void* out, int rows, int hidden, hipStream_t st) {
    const long n = (long)rows * (hidden >> 3);
    hipLaunchKernelGGL(swiglu_parts_kernel, dim3((n + 255) / 256), dim3(256), 0, st, parts, ks, stride, (bf16_t*)out, rows, hidden);
}

// 4 consecutive logits of `row` starting at column j (V % 4 == 0)
__device__ inline void sample_logit4(const SampleP& p, long row, int j, float (&v)[4]) {
    if (p.logits_ks <= 0) { const float4 u = *(const float4*)(p.logits + row * p.V + j); v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w; return; }
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.logits_ks; ++s) {
        const float4 u = *(const float4*)(p.logits + s * p.logits_stride + row * p.V + j);
        a[0] += u.x; a[1] += u.y; a[2] += u.z; a[3] += u.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = p.round_bf16 ? bf2f(f2bf(a[e])) : a[e];
}
__global__ __launch_bounds__(1024) void sample_greedy_kernel(SampleP p) {
    __shared__ float smv[16]; __shared__ int smi[16];
    const int i = blockIdx.x, step = *p.step_ptr;
    // cfg_flag of decode_n_tokens: loop index = step-1; flag drops once (step-1) > cfg_interval
    const bool mix = p.use_cfg && !(p.cfg_interval > -1 && (step - 1) > p.cfg_interval);
    float best = -INFINITY; int bi = 0x7fffffff;
    float* lo = p.logits_out ? p.logits_out + ((long)i * p.n_new + step) * p.V : nullptr;
    for (int j = threadIdx.x * 4; j < p.V; j += blockDim.x * 4) {
        float v[4]; sample_logit4(p, i, j, v);
        if (mix) {
            float u[4]; sample_logit4(p, i + p.B, j, u);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = u[e] + (v[e] - u[e]) * p.cfg_scale;
        }
        if (lo) *(float4*)(lo + j) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) if (v[e] > best) { best = v[e]; bi = j + e; }   // ascending j per thread: first max kept
    }
    // wave reduce (max value, then min index)
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { smv[w] = best; smi[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < nw; ++k) if (smv[k] > best || (smv[k] == best && smi[k] < bi)) { best = smv[k]; bi = smi[k]; }
        p.out_tokens[(long)i * p.n_new + step] = bi;
        const int fb = p.forced ? p.forced[(long)i * p.n_new + step] : bi;
        p.cur_tok[i] = fb;
        if (p.use_cfg) p.cur_tok[i + p.B] = fb;
    }
}
// ---- stochastic sampling (generate.py:17-74): temperature, top-k (k-th value threshold), top-p (nucleus on the sorted softmax),
// softmax, multinomial(1).  One 1024-thread block per image; the sorted copy lives in LDS (bitonic sort, descending).
// RNG: Philox4x32-10 keyed by the caller's seed, counter = (global image row, step): reproducible and order-independent.
// Ties: filtering is by VALUE threshold (logits >= k-th / nucleus-cut value are kept), which equals the reference's
// index-based removal whenever the boundary values are distinct.
__device__ inline void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ inline float philox_uniform(unsigned long long seed, unsigned a, unsigned b) {
    unsigned c0 = a, c1 = b, c2 = 0x243F6A88u, c3 = 0x85A308D3u, k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    for (int r = 0; r < 10; ++r) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    return (float)(c0 >> 8) * (1.0f / 16777216.0f);      // [0, 1)
}
// exclusive prefix sum of one value per thread over a 1024-thread block (16 waves); returns the block total via `total`
__device__ inline float block_excl_scan(float v, float* sm /* >= 34 floats */, float& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float inc = v;
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    __syncthreads();
    if (lane == 63) sm[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) { float a = 0.f; for (int k = 0; k < nw; ++k) { const float t = sm[k]; sm[k] = a; a += t; } sm[32] = a; }
    __syncthreads();
    total = sm[32];
    return sm[w] + inc - v;
}
__global__ __launch_bounds__(1024) void sample_stochastic_kernel(SampleP p, int Vp) {
    extern __shared__ float keys[];          // Vp floats (Vp = pow2 >= V), sorted descending
    __shared__ float sm[40]; __shared__ int cut_s; __shared__ int tok_s;
    const int i = blockIdx.x, step = *p.step_ptr, tid = threadIdx.x, nt = blockDim.x;
    if (p.dyn) { p.seed = p.dyn->seed; p.temperature = p.dyn->temperature; p.top_k = p.dyn->top_k; p.top_p = p.dyn->top_p; }
    const bool mix = p.use_cfg && !(p.cfg_interval > -1 && (step - 1) > p.cfg_interval);
    const float invT = 1.0f / fmaxf(p.temperature, 1e-5f);
    float* lo = p.logits_out ? p.logits_out + ((long)i * p.n_new + step) * p.V : nullptr;
    auto mixed4 = [&](int j, float (&v)[4]) {
        sample_logit4(p, i, j, v);
        if (mix) { float u[4]; sample_logit4(p, i + p.B, j, u);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = u[e] + (v[e] - u[e]) * p.cfg_scale; }
    };
    for (int j = tid * 4; j < Vp; j += nt * 4) {
        float v[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (j < p.V) { mixed4(j, v); if (lo) *(float4*)(lo + j) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= invT; }
        keys[j] = v[0]; keys[j + 1] = v[1]; keys[j + 2] = v[2]; keys[j + 3] = v[3];
    }
    __syncthreads();
    const bool filt = p.top_k > 0 || p.top_p < 1.0f;
    float thr = -INFINITY;
    float mx_sel = -INFINITY;
    const bool select_only = p.top_k > 0 && !(p.top_p < 1.0f);
    if (select_only) {
        // top-k alone (every script of the reference: top_k = 2000, top_p = 1.0): the filter needs ONE number, the k-th largest logit — a 4-pass radix
        // select over the keys in LDS (8 bits per pass from the top, integer histogram) finds exactly the value a full sort would put at position k-1,
        // in ~10 us instead of the ~200 us of the 105-pass bitonic sort of 16384 keys.  The nucleus filter needs the sorted order and keeps the sort below.
        __shared__ unsigned hist[256]; __shared__ unsigned part16[16]; __shared__ unsigned sel_prefix, sel_rank;
        auto key2u = [](float f) { const unsigned b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); };      // order-preserving
        const int kk = p.top_k < 1 ? 1 : (p.top_k > p.V ? p.V : p.top_k);
        unsigned prefix = 0, maskb = 0, rank = (unsigned)kk;
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int t = tid; t < 256; t += nt) hist[t] = 0;
            __syncthreads();
            for (int a = tid; a < Vp; a += nt) { const unsigned u = key2u(keys[a]); if ((u & maskb) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u); }
            __syncthreads();
            if (tid < 16) { unsigned a = 0; for (int t = 0; t < 16; ++t) a += hist[tid * 16 + t]; part16[tid] = a; }
            __syncthreads();
            if (tid == 0) {
                unsigned r = rank; int grp = 15;
                for (; grp > 0; --grp) { if (part16[grp] >= r) break; r -= part16[grp]; }
                int bkt = grp * 16 + 15;
                for (; bkt > grp * 16; --bkt) { if (hist[bkt] >= r) break; r -= hist[bkt]; }
                sel_prefix = prefix | ((unsigned)bkt << shift); sel_rank = r;
            }
            __syncthreads();
            prefix = sel_prefix; rank = sel_rank; maskb |= 0xffu << shift;
            __syncthreads();
        }
        thr = __uint_as_float((prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix);
        for (int j = tid; j < p.V; j += nt) mx_sel = fmaxf(mx_sel, keys[j]);
        mx_sel = block_max(mx_sel, sm);
    } else if (filt) {
        for (int k = 2; k <= Vp; k <<= 1)
            for (int jj = k >> 1; jj > 0; jj >>= 1) {
                for (int a = tid; a < Vp; a += nt) {
                    const int b = a ^ jj;
                    if (b > a) { const float x = keys[a], y = keys[b]; const bool desc = (a & k) == 0; if ((x < y) == desc) { keys[a] = y; keys[b] = x; } }
                }
                __syncthreads();
            }
        int kk = p.V;
        if (p.top_k > 0) { kk = p.top_k < 1 ? 1 : (p.top_k > p.V ? p.V : p.top_k); thr = keys[kk - 1]; }
        if (p.top_p < 1.0f) {
            // nucleus over the top-k-filtered sorted logits (values < thr are already -inf there)
            const float mx = keys[0];
            const int per = (Vp + nt - 1) / nt, a0 = tid * per;
            float loc = 0.f;
            for (int a = a0; a < a0 + per && a < Vp; ++a) { const float v = keys[a]; if (v >= thr && v > -INFINITY) loc += expf(v - mx); }
            float total; const float excl = block_excl_scan(loc, sm, total);
            if (tid == 0) cut_s = Vp;
            __syncthreads();
            // sorted position a (>= 1) is removed iff cumprob[a-1] > top_p
            float run = excl;
            for (int a = a0; a < a0 + per && a < Vp; ++a) {
                const float v = keys[a];
                if (a >= 1 && run / total > p.top_p) { atomicMin(&cut_s, a); break; }
                if (v >= thr && v > -INFINITY) run += expf(v - mx);
            }
            __syncthreads();
            const int cut = cut_s;
            if (cut < Vp && cut >= 1) thr = fmaxf(thr, keys[cut - 1]);
        }
        (void)kk;
    }
    // softmax + multinomial over the filtered logits in index order
    float mx = -INFINITY;
    if (select_only) mx = mx_sel;
    else if (filt) mx = keys[0];
    else { for (int j = tid; j < p.V; j += nt) mx = fmaxf(mx, keys[j]); mx = block_max(mx, sm); }
    __syncthreads();
    const int per = (p.V / 4 + nt - 1) / nt * 4, j0 = tid * per;      // contiguous chunk per thread, multiple of 4
    float loc = 0.f;
    for (int j = j0; j < j0 + per && j < p.V; j += 4) {
        float v[4]; mixed4(j, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float x = v[e] * invT; if (x >= thr) loc += expf(x - mx); }
    }
    float total; const float excl = block_excl_scan(loc, sm, total);
    const float target = philox_uniform(p.seed, (unsigned)(p.row0 + i), (unsigned)step) * total;
    // owner of the draw = the LAST thread with a non-empty chunk whose exclusive prefix is <= target.  (Testing
    // excl <= target < excl + loc per thread can select nobody: the scan's excl(t+1) is not bit-equal to excl(t) + loc(t).)
    // The first non-empty chunk has excl == 0 exactly, so an owner always exists and the token is always in the support.
    if (tid == 0) tok_s = -1;
    __syncthreads();
    if (loc > 0.f && excl <= target) atomicMax(&tok_s, tid);
    __syncthreads();
    const int owner = tok_s;
    __syncthreads();
    if (tid == 0) tok_s = -1;
    __syncthreads();
    if (tid == owner) {
        float run = excl; int pick = -1, last = -1;
        for (int j = j0; j < j0 + per && j < p.V; j += 4) {
            float v[4]; mixed4(j, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float x = v[e] * invT; if (x >= thr) { run += expf(x - mx); last = j + e; if (pick < 0 && run > target) pick = j + e; } }
        }
        if (pick < 0) pick = last;                // rounding left the running sum at or below target: the chunk's last kept token
        tok_s = pick;
    }
    __syncthreads();
    if (tid == 0) {
        int t = tok_s;
        if (t < 0) { t = 0; }       // only if every logit is -inf / NaN (no support at all)
        p.out_tokens[(long)i * p.n_new + step] = t;
        const int fb = p.forced ? p.forced[(long)i * p.n_new + step] : t;
        p.cur_tok[i] = fb;
        if (p.use_cfg) p.cur_tok[i + p.B] = fb;
    }
}
extern "C" void car_launch_sample_greedy(const SampleP* p, hipStream_t st) {
    if (p->stochastic) {
        int Vp = 4; while (Vp < p->V) Vp <<= 1;
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)sample_stochastic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024); attr_set = true; }
        hipLaunchKernelGGL(sample_stochastic_kernel, dim3(p->B), dim3(1024), (size_t)Vp * 4, st, *p, Vp);
        return;
    }
    int th = p->V / 4; th = ((th + 63) / 64) * 64; if (th > 1024) th = 1024; if (th < 64) th = 64;
    hipLaunchKernelGGL(sample_greedy_kernel, dim3(p->B), dim3(th), 0, st, *p);
}

// step bookkeeping: pos += 1, step += 1 (device-side so a captured hipGraph can be replayed)
__global__ void advance_kernel(int* pos, int* step) { if (threadIdx.x == 0) { *pos += 1; *step += 1; } }
extern "C" void car_launch_advance(int* pos, int* step, hipStream_t st) { hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, st, pos, step); }
// (pos, step) of all 8 chain slots from kernel ARGUMENTS (the copy from pageable context memory it replaces could block the host until the stream
// drained and, under a caller's stream capture, read the host words at replay time)
__global__ void set_pos_step_kernel(int* dst, int pos, int step) { if (threadIdx.x < 16) dst[threadIdx.x] = (threadIdx.x & 1) ? step : pos; }
extern "C" void car_launch_set_pos_step(int* dst, int pos, int step, hipStream_t st) { hipLaunchKernelGGL(set_pos_step_kernel, dim3(1), dim3(64), 0, st, dst, pos, step); }

// ------------------------------------------------------------------ V^T builder for the GEMM-form attention
// src [B, Tn, ld] (column offset already applied), C channels -> dst [B, C, Tpad], zero padded in t.
template <typename T>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const void* src_, long ld, long sb, void* dst_, int Tn, int Tpad, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const T* src = (const T*)src_ + (long)b * sb; T* dst = (T*)dst_ + (long)b * C * Tpad;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + tx;
        tile[r][tx] = (t < Tn && c < C) ? ET<T>::ld(src + (long)t * ld + c) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        if (c < C && t < Tpad) ET<T>::st(dst + (long)c * Tpad + t, tile[tx][r]);
    }
}
extern "C" void car_launch_transpose_pad(int mode, const void* src, long ld, long sb, void* dst, int B, int Tn, int Tpad, int C, hipStream_t st) {
    dim3 g((Tpad + 31) / 32, (C + 31) / 32, B);
    LAUNCH_T(mode, transpose_pad_kernel, g, dim3(256), st, src, ld, sb, dst, Tn, Tpad, C);
}
