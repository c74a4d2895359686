// t5.hip — kernels specific to the caption encoder front-end: the reference's T5Embedder
// (language/t5.py:58-79, get_text_embeddings :185-201) runs HF T5EncoderModel (Flan-T5-XL) over 120 padded token ids and
// hands `last_hidden_state` + the attention mask to generate().  The encoder stack itself is GEMMs + RMSNorm from gemm.hip /
// ops.hip (car_t5_encode in engine_t5.hip); what is T5-only lives here:
//   t5_prep          int64 ids / mask (tokenizer output) -> int32 row indices of `shared.weight` + uint8 key mask
//   t5_softmax       P = softmax(S + position_bias[h] + (1 - mask[b]) * finfo.min) over keys, the three-term sum of HF's
//                    T5Attention.forward / eager path (modeling_t5.py: scores += position_bias; mask added to position_bias),
//                    scaling 1.0 (T5 folds 1/sqrt(d) into its initialisation), bf16 roundings at the torch op boundaries
//   t5_gated_act     exact-mode T5DenseGatedActDense gate: out = rnd(gelu_new(a)) * g on the block-16 interleaved wi_0|wi_1 image
#include "car_common.h"

__global__ void t5_prep_kernel(const long long* ids, const long long* mask, int* ids32, unsigned char* mk, long n, int vocab) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long v = ids[i]; if (v < 0) v = 0; if (v >= vocab) v = vocab - 1;     // torch raises on out-of-range ids; clamp keeps the gather in bounds
    ids32[i] = (int)v; mk[i] = mask ? (mask[i] != 0) : 1;
}
extern "C" void car_launch_t5_prep(const long long* ids, const long long* mask, int* ids32, unsigned char* mk, long n, int vocab, hipStream_t st) {
    hipLaunchKernelGGL(t5_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ids, mask, ids32, mk, n, vocab);
}

// rows ordered (b, head, i).  bias: T-rounded values as fp32 [heads][Tq][ncols].  A row whose keys are all masked degenerates to the
// uniform distribution exactly as the additive finfo.min form does.
template <typename T>
__global__ __launch_bounds__(128) void t5_softmax_kernel(const float* S, long lds, void* P_, long ldp, int ncols, const float* bias,
                                                         const unsigned char* mask, int Tq, int n_head) {
    __shared__ float sm[20];
    const long r = blockIdx.x;
    const int i = (int)(r % Tq), h = (int)((r / Tq) % n_head); const long b = r / Tq / n_head;
    const float* s = S + r * lds; T* P = (T*)P_ + r * ldp;
    const float* bi = bias + ((long)h * Tq + i) * ncols; const unsigned char* mk = mask + b * ncols;
    float mx = -INFINITY, mx_all = -INFINITY;
    for (int j = threadIdx.x; j < ncols; j += blockDim.x) {
        const float v = ET<T>::rnd(ET<T>::rnd(s[j]) + bi[j]);
        mx_all = fmaxf(mx_all, v);
        if (mk[j]) mx = fmaxf(mx, v);
    }
    mx = block_max(mx, sm); (void)mx_all;
    const bool none = mx == -INFINITY;
    float sum = 0.f;
    for (int j = threadIdx.x; j < ncols; j += blockDim.x) {
        const float v = ET<T>::rnd(ET<T>::rnd(s[j]) + bi[j]);
        if (none) sum += 1.f; else if (mk[j]) sum += expf(v - mx);
    }
    sum = block_sum(sum, sm);
    const float inv = 1.0f / sum;
    for (int j = threadIdx.x; j < ldp; j += blockDim.x) {
        float v = 0.f;
        if (j < ncols) {
            if (none) v = inv;
            else if (mk[j]) v = expf(ET<T>::rnd(ET<T>::rnd(s[j]) + bi[j]) - mx) * inv;
        }
        ET<T>::st(P + j, v);
    }
}
extern "C" void car_launch_t5_softmax(int mode, const float* S, long lds, void* P, long ldp, long rows, int ncols, const float* bias,
                                      const unsigned char* mask, int Tq, int n_head, hipStream_t st) {
    if (mode == 1) hipLaunchKernelGGL(t5_softmax_kernel<bf16_t>, dim3((unsigned)rows), dim3(128), 0, st, S, lds, P, ldp, ncols, bias, mask, Tq, n_head);
    else hipLaunchKernelGGL(t5_softmax_kernel<float>, dim3((unsigned)rows), dim3(128), 0, st, S, lds, P, ldp, ncols, bias, mask, Tq, n_head);
}

template <typename T>
__global__ void t5_gated_act_kernel(const void* in_, void* out_, long rows, int hidden) {
    const long total = rows * hidden;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c = (int)(i % hidden); const long r = i / hidden;
        const T* row = (const T*)in_ + r * 2 * hidden + (c >> 4) * 32 + (c & 15);
        const float a = ET<T>::ld(row), g = ET<T>::ld(row + 16);
        ET<T>::st((T*)out_ + i, ET<T>::rnd(gelu_tanh_f(a)) * g);
    }
}
extern "C" void car_launch_t5_gated_act(int mode, const void* in, void* out, long rows, int hidden, hipStream_t st) {
    long total = rows * hidden; int g = (int)((total + 255) / 256); if (g > 4096) g = 4096;
    if (mode == 1) hipLaunchKernelGGL(t5_gated_act_kernel<bf16_t>, dim3(g), dim3(256), 0, st, in, out, rows, hidden);
    else hipLaunchKernelGGL(t5_gated_act_kernel<float>, dim3(g), dim3(256), 0, st, in, out, rows, hidden);
}
