// pack.hip — device-side weight packing for the decode linears (checkpoint I/O).
// The reference loads a state dict and keeps nn.Linear weights row-major (sample_t2i.py:64-83); the decode kernels of
// decode2.hip stream MFMA-fragment images instead.  These kernels build those images on the GPU straight from the
// checkpoint tensor (f32 or bf16, host-staged or already resident), so loading a model costs one pass over the weights at
// HBM speed instead of scalar host loops:
//   rows_to_bf16      src [N][K] (f32 | bf16) -> row-major bf16 (the prefill GEMM operand), optionally interleaving w1 | w3
//                     rows in blocks of 16 (the SwiGLU epilogues need the (a, c) pair in adjacent row-blocks)
//   pack_frag_bf16    row-major bf16 [N][K] -> [N/16][K/32][64 lanes][8] (lane l: row l&15, k (l>>4)*8..+8)
//   row_amax_scale    per-output-row scale = amax / 448 of the SOURCE values (OCP e4m3fn max; 1 where the row is zero)
//   quant_pack_fp8    row-major bf16 (interleaved image) + scales -> e4m3 image [N/16][K/64][64 lanes][16 B] (8 bytes of k-block
//                     2j then 8 bytes of k-block 2j+1) and the DEQUANTISED values written back over the row-major copy, so that
//                     prefill and decode see one set of effective weights
#include "car_common.h"

typedef __attribute__((ext_vector_type(2))) float pk_f32x2;

__device__ inline float src_ld(const void* p, int dtype, long i) { return dtype == 1 ? bf2f(((const bf16_t*)p)[i]) : ((const float*)p)[i]; }

// dst_row(r): ileave 0 -> r; 1 (w1) -> (r/16)*32 + r%16; 2 (w3) -> (r/16)*32 + 16 + r%16
__global__ void rows_to_bf16_kernel(const void* src, int dtype, bf16_t* dst, long N, long K, int ileave) {
    const long n8 = N * (K >> 3);
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x; const long st = (long)gridDim.x * blockDim.x;
    for (; i < n8; i += st) {
        const long r = i / (K >> 3), c = (i - r * (K >> 3)) << 3;
        const long dr = ileave == 0 ? r : ((r >> 4) * 32 + (ileave == 2 ? 16 : 0) + (r & 15));
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) o[e >> 1] = (unsigned)f2bf(src_ld(src, dtype, r * K + c + e)) | ((unsigned)f2bf(src_ld(src, dtype, r * K + c + e + 1)) << 16);
        *(uint4*)(dst + dr * K + c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

__global__ void pack_frag_bf16_kernel(const bf16_t* src, bf16_t* dst, long N, long K) {
    const long nkb = K >> 5, nch = (N >> 4) * nkb * 64;          // one thread per 16-byte lane slot
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x; const long st = (long)gridDim.x * blockDim.x;
    for (; i < nch; i += st) {
        const int l = (int)(i & 63); const long ck = i >> 6, rb = ck / nkb, kb = ck - rb * nkb;
        *(uint4*)(dst + i * 8) = *(const uint4*)(src + (rb * 16 + (l & 15)) * K + kb * 32 + (l >> 4) * 8);
    }
}

// one wave per row; src is the ORIGINAL tensor (f32 | bf16), row r of the source goes to scale[dst_row(r)]
__global__ __launch_bounds__(64) void row_amax_scale_kernel(const void* src, int dtype, float* scale, long K, int ileave) {
    const long r = blockIdx.x;
    float a = 0.f;
    for (long k = threadIdx.x; k < K; k += 64) a = fmaxf(a, fabsf(src_ld(src, dtype, r * K + k)));
    a = wave_max(a);
    const long dr = ileave == 0 ? r : ((r >> 4) * 32 + (ileave == 2 ? 16 : 0) + (r & 15));
    if (threadIdx.x == 0) scale[dr] = a > 0.f ? a / 448.0f : 1.0f;
}

// quantise the source rows (original precision) to e4m3 with their row scale; write (a) the dequantised bf16 value into the
// row-major image and (b) the byte into the packed fp8 image.  One thread per 8 consecutive k of one row.
__global__ void quant_pack_fp8_kernel(const void* src, int dtype, const float* scale, bf16_t* rowmajor, unsigned char* pk, long N, long K, int ileave) {
    const long n8 = N * (K >> 3), nkp = K >> 6;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x; const long st = (long)gridDim.x * blockDim.x;
    for (; i < n8; i += st) {
        const long r = i / (K >> 3), c = (i - r * (K >> 3)) << 3;
        const long dr = ileave == 0 ? r : ((r >> 4) * 32 + (ileave == 2 ? 16 : 0) + (r & 15));
        const float s = scale[dr], inv = 1.0f / s;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src_ld(src, dtype, r * K + c + e) / s;
        (void)inv;
        // v_cvt_pk_fp8_f32: OCP e4m3fn, round-to-nearest-even, saturating on gfx950
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], w1, true);
        const pk_f32x2 d0 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, false), d1 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, true);
        const pk_f32x2 d2 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, false), d3 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, true);
        unsigned o[4];
        o[0] = (unsigned)f2bf(d0[0] * s) | ((unsigned)f2bf(d0[1] * s) << 16); o[1] = (unsigned)f2bf(d1[0] * s) | ((unsigned)f2bf(d1[1] * s) << 16);
        o[2] = (unsigned)f2bf(d2[0] * s) | ((unsigned)f2bf(d2[1] * s) << 16); o[3] = (unsigned)f2bf(d3[0] * s) | ((unsigned)f2bf(d3[1] * s) << 16);
        *(uint4*)(rowmajor + dr * K + c) = make_uint4(o[0], o[1], o[2], o[3]);
        // packed slot: row-block dr/16, k-block pair (c/64), lane ((c%32)/8)*16 + dr%16, half (c/32)&1
        const long rb = dr >> 4, kp = c >> 6; const int half = (int)((c >> 5) & 1), lane = (int)(((c & 31) >> 3) * 16 + (dr & 15));
        *(uint2*)(pk + ((((rb * nkp + kp) * 64 + lane) * 2 + half) << 3)) = make_uint2((unsigned)w0, (unsigned)w1);
    }
}

static inline int grid_for(long n) { long g = (n + 255) / 256; if (g > 16384) g = 16384; if (g < 1) g = 1; return (int)g; }

extern "C" void car_launch_rows_to_bf16(const void* src, int dtype, void* dst, long N, long K, int ileave, hipStream_t st) {
    hipLaunchKernelGGL(rows_to_bf16_kernel, dim3(grid_for(N * (K >> 3))), dim3(256), 0, st, src, dtype, (bf16_t*)dst, N, K, ileave);
}
extern "C" void car_launch_pack_frag_bf16(const void* src, void* dst, long N, long K, hipStream_t st) {
    hipLaunchKernelGGL(pack_frag_bf16_kernel, dim3(grid_for((N >> 4) * (K >> 5) * 64)), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, N, K);
}
extern "C" void car_launch_row_amax_scale(const void* src, int dtype, float* scale, long N, long K, int ileave, hipStream_t st) {
    hipLaunchKernelGGL(row_amax_scale_kernel, dim3((unsigned)N), dim3(64), 0, st, src, dtype, scale, K, ileave);
}
extern "C" void car_launch_quant_pack_fp8(const void* src, int dtype, const float* scale, void* rowmajor, void* pk, long N, long K, int ileave, hipStream_t st) {
    hipLaunchKernelGGL(quant_pack_fp8_kernel, dim3(grid_for(N * (K >> 3))), dim3(256), 0, st, src, dtype, scale, (bf16_t*)rowmajor, (unsigned char*)pk, N, K, ileave);
}
