// engine_encode.hip — stages A-C: resize / position-embedding tables, the DINOv2 / ViT control encoder (car_encode_control), Canny control extraction (car_canny)
// (one of the translation units behind include/controlar_hip.h; shared declarations: engine_internal.h)
#include "engine_internal.h"

// ------------------------------------------------------------------------------------- small host-side tables
static void cubic_coeffs(float t, float w[4]) {   // ATen get_cubic_upsample_coefficients, A = -0.75
    const float A = -0.75f;
    float x0 = t + 1.0f; w[0] = ((A * x0 - 5 * A) * x0 + 8 * A) * x0 - 4 * A;
    w[1] = ((A + 2) * t - (A + 3)) * t * t + 1;
    float x2 = 1.0f - t; w[2] = ((A + 2) * x2 - (A + 3)) * x2 * x2 + 1;
    float x3 = 2.0f - t; w[3] = ((A * x3 - 5 * A) * x3 + 8 * A) * x3 - 4 * A;
}
static void bicubic_tab(int out, int in, bool align, std::vector<int>& idx, std::vector<float>& wt) {
    idx.resize((size_t)out * 4); wt.resize((size_t)out * 4);
    for (int d = 0; d < out; ++d) {
        float src;
        if (align) { float sc = out > 1 ? (float)((double)(in - 1) / (double)(out - 1)) : 0.f; src = (float)d * sc; }
        else { float sc = (float)((double)in / (double)out); src = ((float)d + 0.5f) * sc - 0.5f; }
        float fl = std::floor(src); float t = src - fl; int ix = (int)fl;
        cubic_coeffs(t, &wt[(size_t)d * 4]);
        for (int k = 0; k < 4; ++k) { int v = ix - 1 + k; v = v < 0 ? 0 : (v > in - 1 ? in - 1 : v); idx[(size_t)d * 4 + k] = v; }
    }
}

int get_resize(car_ctx* c, int H, int W, int nh, int nw, car_ctx::ResizeTab* out) {
    auto key = std::make_pair(H, W);
    auto it = c->resize_cache.find(key);
    if (it != c->resize_cache.end()) { *out = it->second; return 0; }
    car_ctx::ResizeTab t{nullptr, nullptr, nullptr, nullptr};
    if (c->cfg.resize_mode == CAR_RESIZE_NEAREST) {
        // ATen nearest: floor(dst * (float)in/out) in fp32, clamped (dinov2_adapter.py:20)
        std::vector<int> iy(nh), ix(nw);
        const float sy = (float)((double)H / (double)nh), sx = (float)((double)W / (double)nw);
        for (int i = 0; i < nh; ++i) { int v = (int)std::floor((float)i * sy); iy[i] = v > H - 1 ? H - 1 : v; }
        for (int i = 0; i < nw; ++i) { int v = (int)std::floor((float)i * sx); ix[i] = v > W - 1 ? W - 1 : v; }
        HIPCHK(c, hipMalloc((void**)&t.iy, nh * 4)); HIPCHK(c, hipMalloc((void**)&t.ix, nw * 4));
        HIPCHK(c, hipMemcpy(t.iy, iy.data(), nh * 4, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(t.ix, ix.data(), nw * 4, hipMemcpyHostToDevice));
    } else {
        std::vector<int> iy, ix; std::vector<float> wy, wx;
        bicubic_tab(nh, H, true, iy, wy); bicubic_tab(nw, W, true, ix, wx);
        HIPCHK(c, hipMalloc((void**)&t.iy, iy.size() * 4)); HIPCHK(c, hipMalloc((void**)&t.ix, ix.size() * 4));
        HIPCHK(c, hipMalloc((void**)&t.wy, wy.size() * 4)); HIPCHK(c, hipMalloc((void**)&t.wx, wx.size() * 4));
        HIPCHK(c, hipMemcpy(t.iy, iy.data(), iy.size() * 4, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(t.ix, ix.data(), ix.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(t.wy, wy.data(), wy.size() * 4, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(t.wx, wx.data(), wx.size() * 4, hipMemcpyHostToDevice));
    }
    c->resize_cache[key] = t; *out = t;
    return 0;
}

// HF Dinov2Embeddings.interpolate_pos_encoding (:57-95): bicubic align_corners=False in fp32, per (gh,gw), cached
int get_pos_embed(car_ctx* c, int gh, int gw, void** out) {
    auto key = std::make_pair(gh, gw);
    auto it = c->pos_cache.find(key);
    if (it != c->pos_cache.end()) { *out = it->second; return 0; }
    const std::vector<float>& pe = c->host_keep["adapter.model.embeddings.position_embeddings"];
    const int D = c->cfg.vit_hidden, G = c->cfg.vit_pos_grid;
    if ((int64_t)pe.size() != (int64_t)(G * G + 1) * D) FAIL(c, "position_embeddings has %zu elements, expected %d", pe.size(), (G * G + 1) * D);
    std::vector<float> o((size_t)(gh * gw + 1) * D);
    memcpy(o.data(), pe.data(), (size_t)D * 4);
    if (gh == G && gw == G) memcpy(o.data() + D, pe.data() + D, (size_t)G * G * D * 4);
    else {
        std::vector<int> iy, ix; std::vector<float> wy, wx;
        bicubic_tab(gh, G, false, iy, wy); bicubic_tab(gw, G, false, ix, wx);
        std::vector<float> rows((size_t)G * gw);
        for (int d = 0; d < D; ++d) {
            for (int y = 0; y < G; ++y) for (int x = 0; x < gw; ++x) {
                float acc = 0.f;
                for (int k = 0; k < 4; ++k) acc += pe[(size_t)(1 + y * G + ix[x * 4 + k]) * D + d] * wx[x * 4 + k];
                rows[(size_t)y * gw + x] = acc;
            }
            for (int y = 0; y < gh; ++y) for (int x = 0; x < gw; ++x) {
                float acc = 0.f;
                for (int k = 0; k < 4; ++k) acc += rows[(size_t)iy[y * 4 + k] * gw + x] * wy[y * 4 + k];
                o[(size_t)(1 + y * gw + x) * D + d] = acc;
            }
        }
    }
    void* dp = nullptr;
    const size_t bytes = o.size() * c->esz;
    HIPCHK(c, hipMalloc(&dp, bytes));
    if (c->mode == CAR_F32) { HIPCHK(c, hipMemcpy(dp, o.data(), bytes, hipMemcpyHostToDevice)); }
    else { std::vector<bf16_t> hb(o.size()); for (size_t i = 0; i < o.size(); ++i) hb[i] = f2bf(o[i]); HIPCHK(c, hipMemcpy(dp, hb.data(), bytes, hipMemcpyHostToDevice)); }
    c->pos_cache[key] = dp; *out = dp;
    return 0;
}

// ------------------------------------------------------------------------------------- control encoder
extern "C" int car_encode_control(car_ctx* c, const void* img, int32_t img_dtype, int32_t B, int32_t H, int32_t W, void* out, void* stream_) {
    if (c && check_sticky(c)) return -1;
    if (!c) return -1;
    if (!c->finalized) FAIL(c, "car_encode_control: call car_finalize_weights first");
    if (!c->has_gpt) FAIL(c, "car_encode_control: this context holds VQ weights only");
    if (!img || B <= 0 || H < 16 || W < 16) FAIL(c, "car_encode_control: bad arguments");
    if (img_dtype != CAR_DT_F32 && img_dtype != CAR_DT_BF16) FAIL(c, "car_encode_control: image dtype must be F32 or BF16");
    const car_config& g = c->cfg;
    hipStream_t caller = (hipStream_t)stream_, st = c->stream;
    const int mode = c->mode; const size_t e = c->esz;
    const int p = g.vit_patch, gh = H / 16, gw = W / 16, n = gh * gw, Tn = n + 1, D = g.vit_hidden, nh = g.vit_heads, hd = D / nh;
    const int Kp = (int)rup(3 * p * p, 32), Tpad = (int)rup(Tn, 32);
    car_ctx::ResizeTab rt; if (get_resize(c, H, W, gh * p, gw * p, &rt)) return -1;
    void* pos = nullptr; if (get_pos_embed(c, gh, gw, &pos)) return -1;
    NEED(c, c->ctrl_in, (size_t)B * n * g.dim * e);
    c->ctrl_B = B; c->ctrl_ntok = n;
    const bool flash = use_flash(c, hd);
    // images per chunk.  The unfused (exact) form holds fp32 scores and probabilities, CH*heads*Tn*(Tn + Tpad)*4 B — 4.9 GB at 96 images of 1025 tokens x 6 heads;
    // 384 images, ms per encode: 16 per chunk 501, 32 471, 48 452, 96 436 (profiles/r05_exact_probe_v7.txt); capped so the two tensors stay below 6 GB
    int chmax = flash ? 64 : 96;
    if (!flash) while (chmax > 8 && (size_t)chmax * nh * Tn * (size_t)(Tn + Tpad) * 4 > ((size_t)6 << 30)) chmax /= 2;
    { const char* ev = CAR_KNOB("CAR_ENC_CHUNK"); if (ev && atoi(ev) >= 1 && atoi(ev) <= 256) chmax = atoi(ev); }
    const int CH = B < chmax ? B : chmax;
    NEED(c, c->ws[0], (size_t)CH * n * Kp * e);            // patches, later ctx
    NEED(c, c->ws[1], (size_t)CH * Tn * D * e);            // h
    NEED(c, c->ws[2], (size_t)CH * Tn * D * e);            // y (normed) / tok
    NEED(c, c->ws[3], (size_t)CH * Tn * 3 * D * e);        // q | k | v (separate planes)
    if (!flash) {
        NEED(c, c->ws[4], (size_t)CH * nh * Tn * Tn * 4);      // S fp32
        NEED(c, c->ws[5], (size_t)CH * nh * Tn * Tpad * e);    // P
    }
    NEED(c, c->ws[6], (size_t)CH * D * Tpad * e);          // V^T
    NEED(c, c->ws[7], (size_t)CH * Tn * (g.vit_mlp > g.dim ? g.vit_mlp : g.dim) * e);   // mlp mid / adapter mid
    NEED(c, c->ws[8], (size_t)CH * Tn * D * e);            // ctx
    fence_in(c, caller);
    const std::string a = "adapter.model.";
    for (int b0 = 0; b0 < B; b0 += CH) {
        const int nb = (B - b0) < CH ? (B - b0) : CH;
        const size_t ibytes = img_dtype == CAR_DT_BF16 ? 2 : 4;
        const void* im = (const char*)img + (size_t)b0 * 3 * H * W * ibytes;
        void *patches = c->ws[0].p, *h = c->ws[1].p, *y = c->ws[2].p, *qkv = c->ws[3].p, *P = c->ws[5].p, *vT = c->ws[6].p, *mid = c->ws[7].p, *ctx = c->ws[8].p;
        float* S = (float*)c->ws[4].p;
        car_launch_patchify(mode, im, img_dtype, patches, nb, H, W, gh, gw, p, Kp, g.resize_mode == CAR_RESIZE_BICUBIC_AC, rt.iy, rt.ix, rt.wy, rt.wx, st);
        {   // patch projection (HF :119-149) -> y used as tok buffer
            GemmP q = gp(patches, Kp, Wp(c, a + "embeddings.patch_embeddings.projection.weight"), Kp, y, D, nb * n, D, Kp);
            q.bias = Wp(c, a + "embeddings.patch_embeddings.projection.bias"); q.bias_mode = BIAS_N;
            car_launch_gemm(mode, AMODE_PLAIN, &q, st);
        }
        car_launch_vit_assemble(mode, y, Wp(c, a + "embeddings.cls_token"), pos, h, nb, n, D, st);
        const long rows = (long)nb * Tn;
        void* qp = qkv; void* kp = off(qkv, (size_t)rows * D, e); void* vp = off(qkv, (size_t)2 * rows * D, e);
        for (int l = 0; l < g.vit_layers; ++l) {
            const std::string L = a + "encoder.layer." + std::to_string(l) + ".";
            car_launch_layernorm(mode, h, Wp(c, L + "norm1.weight"), Wp(c, L + "norm1.bias"), y, rows, D, g.vit_ln_eps, st);
            const char* names[3] = {"query", "key", "value"}; void* dst[3] = {qp, kp, vp};
            for (int t = 0; t < 3; ++t) {
                GemmP q = gp(y, D, Wp(c, L + "attention.attention." + names[t] + ".weight"), D, dst[t], D, (int)rows, D, D);
                q.bias = Wp(c, L + "attention.attention." + std::string(names[t]) + ".bias"); q.bias_mode = BIAS_N;
                car_launch_gemm(mode, AMODE_PLAIN, &q, st);
            }
            car_launch_transpose_pad(mode, vp, D, (long)Tn * D, vT, nb, Tn, Tpad, D, st);
            bool fused = false;
            if (flash) {
                FlashP f; memset(&f, 0, sizeof(f));
                f.q = (const bf16_t*)qp; f.k = (const bf16_t*)kp; f.vt = (const bf16_t*)vT; f.o = (bf16_t*)ctx;
                f.q_sb = f.k_sb = f.o_sb = (long)Tn * D; f.q_st = f.k_st = f.o_st = D; f.vt_sb = (long)D * Tpad; f.vt_ld = Tpad;
                f.Tq = f.Tk = Tn; f.H = nh; f.scale = 1.0f / std::sqrt((float)hd); f.mode = 0;
                fused = car_launch_flash64(&f, nb, st) == 0;
            }
            if (!fused) {   // S[b,h] = (Q K^T) * hd^-0.5   (HF eager_attention_forward :153-179; softmax internals fp32)
                GemmP q = gp(qp, D, kp, D, S, Tn, Tn, Tn, hd);
                q.alpha = 1.0f / std::sqrt((float)hd); q.out_f32 = 1; q.nb0 = nb; q.nb1 = nh;
                q.sA0 = (long)Tn * D; q.sA1 = hd; q.sW0 = (long)Tn * D; q.sW1 = hd; q.sC0 = (long)nh * Tn * Tn; q.sC1 = (long)Tn * Tn;
                car_launch_gemm(mode, AMODE_PLAIN, &q, st);
                car_launch_softmax(mode, S, Tn, P, Tpad, (long)nb * nh * Tn, Tn, 0, nullptr, 0, 0, st);
            }
            if (!fused) {   // ctx[b, t, h*hd + d] = P[b,h] @ V[b,h]
                GemmP q = gp(P, Tpad, vT, Tpad, ctx, D, Tn, hd, Tpad);
                q.nb0 = nb; q.nb1 = nh;
                q.sA0 = (long)nh * Tn * Tpad; q.sA1 = (long)Tn * Tpad; q.sW0 = (long)D * Tpad; q.sW1 = (long)hd * Tpad; q.sC0 = (long)Tn * D; q.sC1 = hd;
                car_launch_gemm(mode, AMODE_PLAIN, &q, st);
            }
            {   // h = layer_scale1(dense(ctx)) + h   (HF :342-363)
                GemmP q = gp(ctx, D, Wp(c, L + "attention.output.dense.weight"), D, h, D, (int)rows, D, D);
                q.bias = Wp(c, L + "attention.output.dense.bias"); q.bias_mode = BIAS_N; q.scale = g.vit_variant == 0 ? Wp(c, L + "layer_scale1.lambda1") : nullptr; q.R = h; q.ldr = D;
                car_launch_gemm(mode, AMODE_PLAIN, &q, st);
            }
            car_launch_layernorm(mode, h, Wp(c, L + "norm2.weight"), Wp(c, L + "norm2.bias"), y, rows, D, g.vit_ln_eps, st);
            {   // erf-GELU MLP (HF :281-297)
                GemmP q = gp(y, D, Wp(c, L + "mlp.fc1.weight"), D, mid, g.vit_mlp, (int)rows, g.vit_mlp, D);
                q.bias = Wp(c, L + "mlp.fc1.bias"); q.bias_mode = BIAS_N; q.act = ACT_GELU_ERF;
                car_launch_gemm(mode, AMODE_PLAIN, &q, st);
                GemmP r = gp(mid, g.vit_mlp, Wp(c, L + "mlp.fc2.weight"), g.vit_mlp, h, D, (int)rows, D, g.vit_mlp);
                r.bias = Wp(c, L + "mlp.fc2.bias"); r.bias_mode = BIAS_N; r.scale = g.vit_variant == 0 ? Wp(c, L + "layer_scale2.lambda1") : nullptr; r.R = h; r.ldr = D;
                car_launch_gemm(mode, AMODE_PLAIN, &r, st);
            }
        }
        car_launch_layernorm(mode, h, Wp(c, a + "layernorm.weight"), Wp(c, a + "layernorm.bias"), y, rows, D, g.vit_ln_eps, st);
        // drop CLS (dinov2_adapter.py:29) by addressing, then adapter_mlp (generate.py:138)
        mlp_tanh(c, off(y, (size_t)D, e), D, (long)Tn * D, nb, n, D, "adapter_mlp.", mid, off(c->ctrl_in.p, (size_t)b0 * n * g.dim, e), g.dim, st);
    }
    if (out) HIPCHK(c, hipMemcpyAsync(out, c->ctrl_in.p, (size_t)B * n * g.dim * e, hipMemcpyDeviceToDevice, st));
    fence_out(c, caller);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------- Canny control extraction
// cv2.Canny(img, low, high) of condition/canny.py:6-14 for a batch of 8-bit RGB photos [B,H,W,3] (device).  edges_out: uint8 [B,H,W] in
// {0,255} or NULL; control_out: [B,3,H,W] in the context's element type = 2*(edges/255 - 0.5) replicated over 3 channels
// (sample_t2i.py:125,141) or NULL — ready for car_encode_control.  The hysteresis fixed point is checked on the host between launches
// (this runs in front of the path, not inside the token loop).
extern "C" int car_canny(car_ctx* c, const uint8_t* img_hwc, int32_t B, int32_t H, int32_t W, float low_threshold, float high_threshold,
                         uint8_t* edges_out, void* control_out, void* stream_) {
    if (!c) return -1;
    if (check_sticky(c)) return -1;
    if (!img_hwc || B <= 0 || H <= 0 || W <= 0 || (!edges_out && !control_out)) FAIL(c, "car_canny: bad arguments");
    if (low_threshold > high_threshold) { const float t = low_threshold; low_threshold = high_threshold; high_threshold = t; }
    hipStream_t caller = (hipStream_t)stream_, st = c->stream;
    const long HW = (long)H * W;
    const size_t map_bytes = ((size_t)B * HW + 3) & ~(size_t)3;
    // hysteresis: tile-local fixed points swept to a global one.  The sweeps are enqueued in batches of kSweeps; sweep i looks at the "changed" flag
    // of sweep i-1 and exits at once when the fixed point was already reached, and the host looks at the LAST flag of a batch only: one wait per
    // call for any ordinary picture (a weak-edge chain has to cross tile borders more than kSweeps times to need a second batch), instead of
    // one host round trip per sweep.
    constexpr int kSweeps = 24;
    NEED(c, c->canny_map, map_bytes + 4 * (kSweeps + 1));
    unsigned char* map = (unsigned char*)c->canny_map.p; int* flags = (int*)(map + map_bytes);
    fence_in(c, caller);
    car_launch_canny_grad_nms(img_hwc, map, B, H, W, (int)std::floor(low_threshold), (int)std::floor(high_threshold), st);
    for (int batch = 0; batch < 100000; ++batch) {
        int h = 0;
        HIPCHK(c, hipMemsetAsync(flags, 0, 4 * (kSweeps + 1), st));
        HIPCHK(c, hipMemsetAsync(flags, 1, 1, st));                     // flags[0] = 1 (little-endian byte): the first sweep always runs
        for (int i = 1; i <= kSweeps; ++i) car_launch_canny_hyst(map, B, H, W, flags + i - 1, flags + i, st);
        HIPCHK(c, hipMemcpyAsync(&h, flags + kSweeps, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (!h) break;
    }
    car_launch_canny_finish(c->mode, map, edges_out, control_out, B, HW, st);
    fence_out(c, caller);
    HIPCHK(c, hipGetLastError());
    return 0;
}
