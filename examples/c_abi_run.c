/* Runnable plain-C client of libcontrolar_hip.so (include/controlar_hip.h): reads a model + inputs dump written by
 * tests/test_c_example_gpu.py, runs   car_create -> car_load_tensor* -> car_finalize_weights -> car_encode_control -> car_generate
 * (greedy) -> car_vq_decode   with device buffers it allocates itself through the HIP runtime C API, and writes the tokens and a
 * pixel checksum back.  No Python, no C++, no torch on this side of the boundary.
 * Build:  gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_run.c \
 *             -Lcontrolar_amd/csrc -lcontrolar_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,controlar_amd/csrc -o c_abi_run
 * Run:    ./c_abi_run dump.bin tokens.bin
 * Dump layout (little endian): int32 magic 0x43415231 | car_config | int32 n_tensors | n x { int32 name_len, name, int32 ndim,
 * int64 shape[ndim], float data[] } | int32 B, H, W, T, cap, n_new | float img[B*3*H*W] | float emb[B*T*cap] | int64 mask[B*T]
 * | float cfg_scale */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "controlar_hip.h"

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } while (0)
static int rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n ? 0 : 1; }

int main(int argc, char** argv) {
    if (argc < 3) DIE("usage: %s dump.bin tokens.bin", argv[0]);
    FILE* f = fopen(argv[1], "rb");
    if (!f) DIE("cannot open %s", argv[1]);
    int32_t magic = 0, nt = 0;
    car_config cfg;
    if (rd(f, &magic, 4) || magic != 0x43415231 || rd(f, &cfg, sizeof(cfg)) || rd(f, &nt, 4)) DIE("bad dump header");
    car_ctx* ctx = NULL;
    if (car_create(&ctx, &cfg)) DIE("car_create: %s", car_last_error(NULL));
    for (int i = 0; i < nt; ++i) {
        int32_t nl = 0, nd = 0; char name[512]; int64_t shape[8]; int64_t n = 1;
        if (rd(f, &nl, 4) || nl <= 0 || nl >= (int32_t)sizeof(name) || rd(f, name, (size_t)nl) || rd(f, &nd, 4) || nd < 0 || nd > 8) DIE("bad tensor record %d", i);
        name[nl] = 0;
        if (nd && rd(f, shape, (size_t)nd * 8)) DIE("bad shape of %s", name);
        for (int d = 0; d < nd; ++d) n *= shape[d];
        float* data = (float*)malloc((size_t)n * 4 + 4);
        if (!data || rd(f, data, (size_t)n * 4)) DIE("bad data of %s", name);
        if (car_load_tensor(ctx, name, data, shape, nd, CAR_DT_F32)) DIE("car_load_tensor(%s): %s", name, car_last_error(ctx));
        free(data);
    }
    if (car_finalize_weights(ctx)) DIE("car_finalize_weights: %s", car_last_error(ctx));
    int32_t dims[6];
    if (rd(f, dims, sizeof(dims))) DIE("bad input header");
    const int B = dims[0], H = dims[1], W = dims[2], T = dims[3], cap = dims[4], n_new = dims[5];
    const size_t n_img = (size_t)B * 3 * H * W, n_emb = (size_t)B * T * cap, n_mask = (size_t)B * T, n_px = (size_t)B * 3 * H * W;
    float* h_img = (float*)malloc(n_img * 4); float* h_emb = (float*)malloc(n_emb * 4); int64_t* h_mask = (int64_t*)malloc(n_mask * 8);
    float cfg_scale = 1.0f;
    if (!h_img || !h_emb || !h_mask || rd(f, h_img, n_img * 4) || rd(f, h_emb, n_emb * 4) || rd(f, h_mask, n_mask * 8) || rd(f, &cfg_scale, 4)) DIE("bad inputs");
    fclose(f);
    void *d_img = NULL, *d_emb = NULL, *d_mask = NULL, *d_tok = NULL, *d_px = NULL;
    hipStream_t st = NULL;
    if (hipMalloc(&d_img, n_img * 4) || hipMalloc(&d_emb, n_emb * 4) || hipMalloc(&d_mask, n_mask * 8) || hipMalloc(&d_tok, (size_t)B * n_new * 4) ||
        hipMalloc(&d_px, n_px * 4) || hipStreamCreate(&st)) DIE("hipMalloc / hipStreamCreate failed");
    if (hipMemcpy(d_img, h_img, n_img * 4, hipMemcpyHostToDevice) || hipMemcpy(d_emb, h_emb, n_emb * 4, hipMemcpyHostToDevice) ||
        hipMemcpy(d_mask, h_mask, n_mask * 8, hipMemcpyHostToDevice)) DIE("hipMemcpy failed");
    car_sampling sp;
    memset(&sp, 0, sizeof(sp));
    sp.cfg_scale = cfg_scale; sp.cfg_interval = -1; sp.temperature = 1.0f; sp.top_k = 0; sp.top_p = 1.0f; sp.sample_logits = 0; sp.control_strength = 1.0f;
    if (car_encode_control(ctx, d_img, CAR_DT_F32, B, H, W, NULL, st)) DIE("car_encode_control: %s", car_last_error(ctx));
    if (car_generate(ctx, d_emb, CAR_DT_F32, (const int64_t*)d_mask, B, n_new, 1, &sp, (int32_t*)d_tok, NULL, NULL, st)) DIE("car_generate: %s", car_last_error(ctx));
    if (car_vq_decode(ctx, (const int32_t*)d_tok, B, H / 16, W / 16, (float*)d_px, st)) DIE("car_vq_decode: %s", car_last_error(ctx));
    if (hipStreamSynchronize(st)) DIE("stream sync failed");
    int32_t* h_tok = (int32_t*)malloc((size_t)B * n_new * 4); float* h_px = (float*)malloc(n_px * 4);
    if (hipMemcpy(h_tok, d_tok, (size_t)B * n_new * 4, hipMemcpyDeviceToHost) || hipMemcpy(h_px, d_px, n_px * 4, hipMemcpyDeviceToHost)) DIE("copy back failed");
    double sum = 0.0;
    for (size_t i = 0; i < n_px; ++i) sum += (double)h_px[i];
    FILE* o = fopen(argv[2], "wb");
    if (!o || fwrite(h_tok, 4, (size_t)B * n_new, o) != (size_t)B * n_new || fwrite(&sum, 8, 1, o) != 1 || fclose(o)) DIE("cannot write %s", argv[2]);
    car_stats stt;
    if (!car_get_stats(ctx, &stt)) printf("decode loop: %.2f ms for %lld steps, graph=%d\n", stt.decode_ms, (long long)stt.decode_steps, stt.graph_used);
    car_destroy(ctx);
    return 0;
}
