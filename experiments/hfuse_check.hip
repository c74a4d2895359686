// experiments/hfuse_check.hip — the one form of co-scheduling the round-2 sweeps could not try: the attention of one decode chain and a
// linear of the other chain as workgroups of the SAME launch (horizontal fusion), so that nothing at a kernel boundary or between
// hardware queues stands between them.  profiles/r02_ws_check_half_period.txt: on two streams the attention (166 us) and the other
// chain's six linears / norms (84 us) take 255 us — their serial sum — whatever the launch schedule or residency.
//
// Method (timing only — test infrastructure, not product): the product kernels of controlar_amd/csrc/decode2.hip are compiled here as
// DEVICE functions (the macros below turn `__global__` into `__device__ __forceinline__` and route blockIdx / gridDim through two
// __shared__ variables that a wrapper kernel sets per workgroup), and six wrapper launches per half-period each carry one linear / norm
// of chain B in their first workgroups plus a slice of chain A's attention items in the rest:
//     [wo | 15 %] [ffn_norm | 10 %] [w1|w3 | 25 %] [w2 | 15 %] [attention_norm | 10 %] [wqkv | 25 %]
// Printed: attention alone, the six ops alone (as six small launches of the same wrapper), and the fused half-period.  If the fused
// half-period approaches the attention's own time, the decode step of two chains can drop from 15.6 towards 12.5 ms.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I controlar_amd/csrc experiments/hfuse_check.hip -o experiments/hfuse_check && experiments/hfuse_check
#include <hip/hip_runtime.h>

__shared__ uint3 hf_vblk;        // the block index / grid size the inlined kernel bodies see
__shared__ uint3 hf_vgrid;

#define __global__ __device__ __forceinline__
#define __launch_bounds__(...)
#define blockIdx hf_vblk
#define gridDim hf_vgrid
#define hipLaunchKernelGGL(...) ((void)0)          /* decode2.hip's host launchers become no-ops: only the kernel bodies are wanted */
#define hipFuncSetAttribute(...) hipSuccess
#include "../controlar_amd/csrc/decode2.hip"
#undef __global__
#undef __launch_bounds__
#undef blockIdx
#undef gridDim
#undef hipLaunchKernelGGL
#undef hipFuncSetAttribute
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...) \
    do { kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__); } while (0)
#define __global__ __attribute__((global))
#define __launch_bounds__(...) __attribute__((amdgpu_flat_work_group_size(1, __VA_ARGS__)))

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
template <typename T> static T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); return p; }

enum { OP_NONE = 0, OP_NORM = 1, OP_GEMM = 2 };
struct FusedP {
    int op, n_op_blocks;         // chain B's op in workgroups [0, n_op_blocks)
    GemmDP g; Norm2P n; int rows;
    Attn2P a; int item0, n_items;    // chain A's attention items [item0, item0 + n_items) in the workgroups after them
};

// GEMM tile configurations with 4 waves (256 threads) so that every workgroup of the wrapper has the attention's block size
template <int I, int J, int EPI>
__attribute__((global)) __attribute__((amdgpu_flat_work_group_size(1, 256))) void fused_kernel(FusedP f) {
    const unsigned bx = __builtin_amdgcn_workgroup_id_x();
    const bool is_op = (int)bx < f.n_op_blocks;
    if (threadIdx.x == 0) {
        if (is_op) { hf_vblk = make_uint3(bx, 0, 0); hf_vgrid = make_uint3((unsigned)f.n_op_blocks, 1, 1); }
        else { const int it = f.item0 + (int)bx - f.n_op_blocks; hf_vblk = make_uint3((unsigned)(it % f.a.H), (unsigned)(it / f.a.H), 0); hf_vgrid = make_uint3((unsigned)f.a.H, 1, 1); }
    }
    __syncthreads();
    if (is_op) {
        if (f.op == OP_GEMM) dec_gemm_kernel<I, J, 4, EPI, 0, 0>(f.g);
        else if (f.op == OP_NORM) rmsnorm2_kernel<8>(f.n, f.rows);
    } else {
        dec_attn2_kernel<4, 0, 0>(f.a);
    }
}

__attribute__((global)) void hf_fill_kernel(unsigned* p, size_t n, unsigned seed) {
    size_t i = (size_t)__builtin_amdgcn_workgroup_id_x() * 256 + threadIdx.x; const size_t st = (size_t)4096 * 256;
    for (; i < n; i += st) { unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        const unsigned a = (x & 0x807fu) | (((x >> 7) & 0x3f) + 64) << 7, b2 = ((x >> 16) & 0x807fu) | ((((x >> 23) & 0x3f) + 64) << 7);
        p[i] = a | (b2 << 16); }
}
static void fill(void* p, size_t n_u32, unsigned seed) { hipLaunchKernelGGL(hf_fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned*)p, n_u32, seed); }

template <int I, int J>
static void launch_fused(const FusedP& f, int epi, hipStream_t st) {
    const int nb = f.n_op_blocks + f.n_items;
    if (nb <= 0) return;
    const size_t sh = f.op == OP_GEMM ? (size_t)4 * I * J * 64 * 16 : 0;
#define LF(E) do { static bool attr = false; if (sh > 48 * 1024 && !attr) { (void)hipFuncSetAttribute((const void*)fused_kernel<I, J, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; } \
                   hipLaunchKernelGGL((fused_kernel<I, J, E>), dim3(nb), dim3(256), sh, st, f); } while (0)
    if (epi == EPI_QKV) LF(EPI_QKV); else if (epi == EPI_RESID) LF(EPI_RESID); else if (epi == EPI_SWIGLU) LF(EPI_SWIGLU); else LF(EPI_LOGITS);
#undef LF
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 384, D = 1280, Fh = 3584, H = 20, T = 120, SA = 1152, pos = 631, NL = 6, NKV = 2, HP = 36;
    const size_t per_layer = (size_t)(3 * D * D + D * D + 2 * Fh * D + D * Fh);
    bf16_t* dW = dalloc<bf16_t>(per_layer * NL); fill(dW, per_layer * NL / 2, 12345u);
    const size_t kvper = (size_t)M * H * SA * 64;
    bf16_t* dKV = dalloc<bf16_t>(kvper * 2 * NKV); fill(dKV, kvper * 2 * NKV / 2, 999u);
    const size_t M16 = (size_t)((M + 15) / 16) * 16;
    bf16_t *xn = dalloc<bf16_t>(M16 * D), *att = dalloc<bf16_t>(M16 * D), *mid = dalloc<bf16_t>(M16 * Fh), *hbuf = dalloc<bf16_t>((size_t)M * D), *qb = dalloc<bf16_t>((size_t)M * D), *qa = dalloc<bf16_t>((size_t)M * D), *oa = dalloc<bf16_t>(M16 * D), *nw = dalloc<bf16_t>(D);
    fill(xn, M16 * D / 2, 7u); fill(att, M16 * D / 2, 8u); fill(mid, M16 * Fh / 2, 9u); fill(qa, (size_t)M * D / 2, 10u); fill(nw, D / 2, 11u);
    CK(hipMemset(hbuf, 0, (size_t)M * D * 2));
    float* rope = dalloc<float>((size_t)1200 * 64); CK(hipMemset(rope, 0, 1200 * 64 * 4));
    int* dPos = dalloc<int>(1); CK(hipMemcpy(dPos, &pos, 4, hipMemcpyHostToDevice));
    std::vector<unsigned char> mask((size_t)M * T, 0); std::vector<int> jm(M);
    for (int i = 0; i < M; ++i) { const int Lv = 8 + (i * 13) % 33; for (int t = T - Lv; t < T; ++t) mask[(size_t)i * T + t] = 1; jm[i] = T - Lv; }
    unsigned char* dM = dalloc<unsigned char>(mask.size()); CK(hipMemcpy(dM, mask.data(), mask.size(), hipMemcpyHostToDevice));
    int* dJ = dalloc<int>(M); CK(hipMemcpy(dJ, jm.data(), (size_t)M * 4, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const int items = M * H;

    // the six ops of chain B's layer `it` with 4-wave tiles: (I, J) = (2, 4) for wqkv / wo / w2, (4, 4) for w1|w3 (car_pick_gemm_cfg's shapes at >= 192 rows)
    struct Op { int kind, epi, N, K, I; float share; };
    const Op ops[6] = {{OP_GEMM, EPI_RESID, D, D, 2, 0.15f}, {OP_NORM, 0, 0, 0, 2, 0.10f}, {OP_GEMM, EPI_SWIGLU, 2 * Fh, D, 4, 0.25f},
                       {OP_GEMM, EPI_RESID, D, Fh, 2, 0.15f}, {OP_NORM, 0, 0, 0, 2, 0.10f}, {OP_GEMM, EPI_QKV, 3 * D, D, 2, 0.25f}};
    auto attn_params = [&](int it) {
        Attn2P a; memset(&a, 0, sizeof(a)); a.q = qa; a.pos = dPos; a.mask = dM; a.jmin = dJ; a.out = oa; a.H = H; a.SA = SA; a.T = T; a.dim = D; a.nsplit = 1; a.out_packed = 1;
        a.kc = dKV + kvper * 2 * (it % NKV); a.vc = a.kc + kvper; return a;
    };
    auto make = [&](int it, int k, bool with_op, int item0, int n_items) {
        FusedP f; memset(&f, 0, sizeof(f));
        const Op& o = ops[k];
        bf16_t* w = dW + per_layer * (it % NL);
        bf16_t *wqkv = w, *wo = w + (size_t)3 * D * D, *w13 = wo + (size_t)D * D, *w2 = w13 + (size_t)2 * Fh * D;
        if (with_op && o.kind == OP_GEMM) {
            f.op = OP_GEMM; GemmDP& g = f.g; g.M = M; g.N = o.N; g.K = o.K; g.w_nt = 0;
            const int J = 4, Mb = (M + 15) / 16, MT = (Mb + J - 1) / J;
            f.n_op_blocks = (o.N / (16 * o.I)) * MT;
            if (k == 0) { g.W = wo; g.X = att; g.h = hbuf; }
            else if (k == 2) { g.W = w13; g.X = xn; g.outp = mid; }
            else if (k == 3) { g.W = w2; g.X = mid; g.h = hbuf; }
            else { g.W = wqkv; g.X = xn; g.qout = qb; g.kc = dKV + kvper * 2 * ((it + 1) % NKV); g.vc = g.kc + kvper; g.rope = rope; g.pos = dPos; g.H = H; g.SA = SA; g.dim = D; }
        } else if (with_op) {
            f.op = OP_NORM; f.rows = M; f.n_op_blocks = (M + 3) / 4;
            f.n.h_in = hbuf; f.n.xn = xn; f.n.w = nw; f.n.D = D; f.n.eps = 1e-5f;
        }
        f.a = attn_params(it); f.item0 = item0; f.n_items = n_items;
        return f;
    };
    auto launch = [&](const FusedP& f, int k) { if (ops[k].I == 4) launch_fused<4, 4>(f, ops[k].epi, st); else launch_fused<2, 4>(f, ops[k].epi, st); };
    auto timed = [&](const std::function<void(int)>& body) {
        for (int i = 0; i < 3; ++i) body(i);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(t0, st));
        for (int i = 0; i < HP; ++i) body(i);
        CK(hipEventRecord(t1, st)); CK(hipEventSynchronize(t1)); CK(hipGetLastError());
        float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
        return ms * 1000.f / HP;
    };
    {   // fusion must not change a bit: the six ops + the attention as separate launches vs the six fused launches, from identical state
        std::vector<std::vector<unsigned short>> got[2];
        for (int v = 0; v < 2; ++v) {
            CK(hipMemset(hbuf, 0, (size_t)M * D * 2)); fill(xn, M16 * D / 2, 7u); fill(mid, M16 * Fh / 2, 9u); CK(hipMemset(oa, 0, M16 * D * 2)); CK(hipMemset(qb, 0, (size_t)M * D * 2));
            CK(hipDeviceSynchronize());
            int i0 = 0;
            if (v == 0) { launch(make(0, 1, false, 0, items), 1); for (int k = 0; k < 6; ++k) launch(make(0, k, true, 0, 0), k); }
            else for (int k = 0; k < 6; ++k) { const int n = k == 5 ? items - i0 : (int)(items * ops[k].share); launch(make(0, k, true, i0, n), k); i0 += n; }
            CK(hipStreamSynchronize(st)); CK(hipGetLastError());
            for (auto pr : {std::make_pair(hbuf, (size_t)M * D), std::make_pair(mid, M16 * Fh), std::make_pair(qb, (size_t)M * D), std::make_pair(oa, M16 * D), std::make_pair(xn, M16 * D)}) {
                std::vector<unsigned short> hb(pr.second); CK(hipMemcpy(hb.data(), pr.first, pr.second * 2, hipMemcpyDeviceToHost)); got[v].push_back(std::move(hb)); }
        }
        bool same = true; for (size_t i = 0; i < got[0].size(); ++i) same = same && got[0][i] == got[1][i];
        printf("fused launches vs separate launches of the same bodies (h, mid, q, attention out, xn): %s\n", same ? "bit-identical" : "DIFFERENT");
    }
    const float ta = timed([&](int it) { launch(make(it, 1, false, 0, items), 1); });                       // one launch of all items
    const float ta6 = timed([&](int it) { int i0 = 0; for (int k = 0; k < 6; ++k) { const int n = k == 5 ? items - i0 : (int)(items * ops[k].share); launch(make(it, k, false, i0, n), k); i0 += n; } });
    const float tl = timed([&](int it) { for (int k = 0; k < 6; ++k) launch(make(it, k, true, 0, 0), k); });
    const float tf = timed([&](int it) { int i0 = 0; for (int k = 0; k < 6; ++k) { const int n = k == 5 ? items - i0 : (int)(items * ops[k].share); launch(make(it, k, true, i0, n), k); i0 += n; } });
    printf("M=%d position %d: attention alone (one launch) %6.1f us | attention in the six slices %6.1f us | the six ops alone %6.1f us\n", M, pos, ta, ta6, tl);
    printf("M=%d fused half-period (each launch = one op of chain B + a slice of chain A's attention) %6.1f us  (serial sum %6.1f, two streams: see ws_check)\n", M, tf, ta + tl);
    return 0;
}
