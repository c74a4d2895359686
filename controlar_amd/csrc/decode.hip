// decode.hip — prefill KV scatter of the exact mode: RoPE on q, k of the T prefix rows and the K / V rows
// into the [b, head, S_max, 64] cache that decode_f32.hip's attention walks.  (The round-1 VALU decode attention that lived here was replaced by
// decode_f32.hip in round 4.)
#include "car_common.h"
#include "kernel_params.h"

// ---------------------------------------------------------------------------------------------
// Prefill KV scatter: k/v of the T prefix rows -> cache, with RoPE on q,k in place.
// qkv [b*T, 3*dim] T;  writes q (rotated) back in place, K/V into cache [b,H,S_max,64] at rows 0..T-1.
// reference: gpt_t2i.py:266-277.  Prefix RoPE rows are zero, so rotated q,k of the prefix are 0 —
// reproduced by arithmetic, not special-cased.
template <typename T>
__global__ void prefill_rope_kv_kernel(void* qkv_, void* kcache, void* vcache, const float* rope, int b, int Tn, int H, int dim, int S_max, int t0) {
    const long total = (long)b * Tn * H * 32;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int pr = (int)(i % 32); const int h = (int)((i / 32) % H); const int t = (int)((i / (32L * H)) % Tn); const long bb = i / (32L * H * Tn);
        T* row = (T*)qkv_ + (bb * Tn + t) * 3 * dim;
        const float cs = rope[((long)(t + t0) * 32 + pr) * 2], sn = rope[((long)(t + t0) * 32 + pr) * 2 + 1];      // row t of the window is prefix position t0 + t
        const float q0 = ET<T>::ld(row + h * 64 + 2 * pr), q1 = ET<T>::ld(row + h * 64 + 2 * pr + 1);
        const float k0 = ET<T>::ld(row + dim + h * 64 + 2 * pr), k1 = ET<T>::ld(row + dim + h * 64 + 2 * pr + 1);
        ET<T>::st(row + h * 64 + 2 * pr, q0 * cs - q1 * sn); ET<T>::st(row + h * 64 + 2 * pr + 1, q1 * cs + q0 * sn);
        const float kr0 = ET<T>::rnd(k0 * cs - k1 * sn), kr1 = ET<T>::rnd(k1 * cs + k0 * sn);
        ET<T>::st(row + dim + h * 64 + 2 * pr, kr0); ET<T>::st(row + dim + h * 64 + 2 * pr + 1, kr1);
        T* kc = (T*)kcache + ((bb * H + h) * S_max + t + t0) * 64; T* vc = (T*)vcache + ((bb * H + h) * S_max + t + t0) * 64;
        ET<T>::st(kc + 2 * pr, kr0); ET<T>::st(kc + 2 * pr + 1, kr1);
        vc[2 * pr] = row[2 * dim + h * 64 + 2 * pr]; vc[2 * pr + 1] = row[2 * dim + h * 64 + 2 * pr + 1];
    }
}
extern "C" void car_launch_prefill_rope_kv(int mode, void* qkv, void* kc, void* vc, const float* rope, int b, int Tn, int H, int dim, int S_max, int t0, hipStream_t st) {
    long total = (long)b * Tn * H * 32; int g = (int)((total + 255) / 256); if (g > 4096) g = 4096;
    if (mode == 1) hipLaunchKernelGGL(prefill_rope_kv_kernel<bf16_t>, dim3(g), dim3(256), 0, st, qkv, kc, vc, rope, b, Tn, H, dim, S_max, t0);
    else hipLaunchKernelGGL(prefill_rope_kv_kernel<float>, dim3(g), dim3(256), 0, st, qkv, kc, vc, rope, b, Tn, H, dim, S_max, t0);
}

