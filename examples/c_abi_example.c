/* Plain-C use of libcontrolar_hip.so (include/controlar_hip.h): what a non-Python caller of the reference path would write.
 * Build:  hipcc/gcc -Iinclude examples/c_abi_example.c -Lcontrolar_amd/csrc -lcontrolar_hip -o c_abi_example
 * The CPU test-suite only syntax-checks this file as C99 (the header must stay a C header, not a C++ one). */
#include <stdio.h>
#include <string.h>
#include "controlar_hip.h"

static int fail(car_ctx* ctx, const char* what) {
    fprintf(stderr, "%s: %s\n", what, car_last_error(ctx));
    if (ctx) car_destroy(ctx);
    return 1;
}

/* weights: caller-provided (name, host pointer, shape) triples keyed by the reference state_dict names */
typedef struct { const char* name; const float* data; int64_t shape[4]; int ndim; } named_tensor;

int run_xl_canny(const named_tensor* weights, int n_weights,
                 const void* d_control_img_bf16 /* [B,3,512,512] device */, const void* d_text_emb_bf16 /* [B,120,2048] device */,
                 const int64_t* d_emb_mask /* [B,120] device */, int B, int32_t* d_tokens /* [B,1024] device */,
                 float* d_pixels /* [B,3,512,512] device */, void* hip_stream) {
    car_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.abi_version = CAR_ABI_VERSION; cfg.mode = CAR_BF16;
    cfg.dim = 1280; cfg.n_layer = 36; cfg.n_head = 20; cfg.ffn_hidden = 3584; cfg.vocab_size = 16384;      /* GPT-XL, gpt_t2i.py:556 */
    cfg.cls_token_num = 120; cfg.block_size = 1024; cfg.caption_dim = 2048; cfg.norm_eps = 1e-5f; cfg.rope_base = 10000.f;
    cfg.vit_hidden = 384; cfg.vit_layers = 12; cfg.vit_heads = 6; cfg.vit_mlp = 1536; cfg.vit_patch = 14; cfg.vit_pos_grid = 37;
    cfg.vit_ln_eps = 1e-6f; cfg.resize_mode = CAR_RESIZE_NEAREST;                                            /* condition_type 'canny' */
    cfg.codebook_size = 16384; cfg.codebook_dim = 8; cfg.z_channels = 256; cfg.vq_ch = 128; cfg.vq_num_res_blocks = 2; cfg.vq_n_mult = 5;
    cfg.vq_ch_mult[0] = 1; cfg.vq_ch_mult[1] = 1; cfg.vq_ch_mult[2] = 2; cfg.vq_ch_mult[3] = 2; cfg.vq_ch_mult[4] = 4; cfg.gn_eps = 1e-6f;

    car_ctx* ctx = NULL;
    if (car_create(&ctx, &cfg)) return fail(NULL, "car_create");
    for (int i = 0; i < n_weights; ++i)
        if (car_load_tensor(ctx, weights[i].name, weights[i].data, weights[i].shape, weights[i].ndim, CAR_DT_F32)) return fail(ctx, weights[i].name);
    if (car_finalize_weights(ctx)) return fail(ctx, "car_finalize_weights");

    car_sampling sp;
    memset(&sp, 0, sizeof(sp));
    sp.cfg_scale = 4.0f; sp.cfg_interval = -1; sp.temperature = 1.0f; sp.top_k = 2000; sp.top_p = 1.0f;   /* sample_t2i.py:207-211 defaults */
    sp.sample_logits = 1; sp.seed = 0; sp.control_strength = 1.0f;

    if (car_encode_control(ctx, d_control_img_bf16, CAR_DT_BF16, B, 512, 512, NULL, hip_stream)) return fail(ctx, "car_encode_control");
    if (car_generate(ctx, d_text_emb_bf16, CAR_DT_BF16, d_emb_mask, B, 1024, 1, &sp, d_tokens, NULL, NULL, hip_stream)) return fail(ctx, "car_generate");
    if (car_vq_decode(ctx, d_tokens, B, 32, 32, d_pixels, hip_stream)) return fail(ctx, "car_vq_decode");

    car_stats st;
    if (!car_get_stats(ctx, &st)) printf("decode loop: %.1f ms for %lld steps\n", st.decode_ms, (long long)st.decode_steps);
    car_destroy(ctx);
    return 0;
}
