// engine_vq.hip — stage H and its inverse: the VQGAN decoder (car_vq_decode) and encoder (car_vq_encode) over shared building blocks
// (one of the translation units behind include/controlar_hip.h; shared declarations: engine_internal.h)
#include "engine_internal.h"

// ------------------------------------------------------------------------------------- VQ building blocks (shared by decode and encode)
struct VqOps {
    car_ctx* c; int mode; size_t e; hipStream_t st;
    // GroupNorm stage 1 for free: a 3x3 conv that takes conv3_halo64_kernel also writes the per-tile sum / sum-of-squares partials of its OUTPUT from the
    // epilogue (GemmP::gn_part -> ws[8], the layout of gn_partial_vec_kernel), and a GroupNorm whose input is that very tensor skips its read-only pass.
    // `part_of` = the tensor whose partials ws[8] currently holds (null: none); every other writer of a tensor clears it.
    mutable const void* part_of = nullptr;
    void conv3(const void* x, void* y, const std::string& name, int nb, int Ho, int Wo, int Cin, int Cout, int ups, const void* R, int amode = AMODE_CONV3) const {
        GemmP q = gp(x, 0, Wp(c, name + ".weight"), 9 * (long)Cin, y, Cout, nb * Ho * Wo, Cout, 9 * Cin);
        q.bias = Wp(c, name + ".bias"); q.bias_mode = BIAS_N; q.Ho = Ho; q.Wo = Wo; q.Cin = Cin; q.ups = ups; q.R = R; q.ldr = Cout;
        q.patch = 1;        // 16x16 spatial patch order of the GEMM rows where the launcher can use it (bf16, Ho and Wo multiples of 16)
        part_of = nullptr;
        if (amode == AMODE_CONV3 && car_conv3_halo64_ok(mode, &q) && Cout <= 512) { q.gn_part = (float*)c->ws[8].p; part_of = y; }
        if (car_launch_gemm(mode, amode, &q, st) != 0) {      // the launcher refused the fused partials (cannot happen after the same predicate said yes): plain conv,
            q.gn_part = nullptr; part_of = nullptr;            // and the next GroupNorm runs its own stage 1
            (void)car_launch_gemm(mode, amode, &q, st);
        }
    }
    void conv1(const void* x, void* y, const std::string& name, int M, int Cin, int Cout, const void* R) const {
        GemmP q = gp(x, Cin, Wp(c, name + ".weight"), Cin, y, Cout, M, Cout, Cin);
        q.bias = Wp(c, name + ".bias"); q.bias_mode = BIAS_N; q.R = R; q.ldr = Cout;
        part_of = nullptr;
        car_launch_gemm(mode, AMODE_PLAIN, &q, st);
    }
    void gn(const void* x, void* y, const std::string& name, int nb, int HW, int C, int swish) const {
        const int have = (part_of != nullptr && part_of == x) ? 1 : 0;
        car_launch_groupnorm_ex(mode, x, Wp(c, name + ".weight"), Wp(c, name + ".bias"), y, (float*)c->ws[8].p, (float*)c->ws[9].p, nb, HW, C, 32, c->cfg.gn_eps, swish, have, st);
        part_of = nullptr;
    }
    // kinds: 0 ResnetBlock (vq_model.py:300-315), 1 AttnBlock (:328-352, single head over HW positions),
    //        2 Upsample (nearest x2 folded into the conv gather, :375-379), 3 Downsample (pad (0,1,0,1) + conv stride 2, :382-396)
    void blocks(const std::vector<VqItem>& layout, int nb, void*& x, void*& t1, void*& t2, void*& t3, int& Hc, int& Wc) const {
        for (auto& it : layout) {
            const int HW = Hc * Wc;
            if (it.kind == 0) {
                gn(x, t1, it.name + ".norm1", nb, HW, it.cin, 1);
                conv3(t1, t2, it.name + ".conv1", nb, Hc, Wc, it.cin, it.cout, 0, nullptr);
                gn(t2, t1, it.name + ".norm2", nb, HW, it.cout, 1);
                if (it.cin != it.cout) { conv1(x, t3, it.name + ".nin_shortcut", nb * HW, it.cin, it.cout, nullptr); conv3(t1, t2, it.name + ".conv2", nb, Hc, Wc, it.cout, it.cout, 0, t3); std::swap(x, t2); }
                else { conv3(t1, x, it.name + ".conv2", nb, Hc, Wc, it.cout, it.cout, 0, x); }
            } else if (it.kind == 1) {
                const int C = it.cin; const int Tp = (int)rup(HW, 32);
                gn(x, t1, it.name + ".norm", nb, HW, C, 0);
                void* qb = c->ws[7].p; void* kb = off(qb, (size_t)nb * HW * C, e); void* vb = off(qb, (size_t)2 * nb * HW * C, e);
                conv1(t1, qb, it.name + ".q", nb * HW, C, C, nullptr); conv1(t1, kb, it.name + ".k", nb * HW, C, C, nullptr); conv1(t1, vb, it.name + ".v", nb * HW, C, C, nullptr);
                float* S = (float*)c->ws[4].p;
                { GemmP q = gp(qb, C, kb, C, S, HW, HW, HW, C); q.alpha = 1.0f / std::sqrt((float)C); q.out_f32 = 1; q.nb0 = nb; q.sA0 = (long)HW * C; q.sW0 = (long)HW * C; q.sC0 = (long)HW * HW; car_launch_gemm(mode, AMODE_PLAIN, &q, st); }
                car_launch_softmax(mode, S, HW, c->ws[5].p, Tp, (long)nb * HW, HW, 0, nullptr, 0, 0, st);
                car_launch_transpose_pad(mode, vb, C, (long)HW * C, c->ws[6].p, nb, HW, Tp, C, st);
                { GemmP q = gp(c->ws[5].p, Tp, c->ws[6].p, Tp, t2, C, HW, C, Tp); q.nb0 = nb; q.sA0 = (long)HW * Tp; q.sW0 = (long)C * Tp; q.sC0 = (long)HW * C; car_launch_gemm(mode, AMODE_PLAIN, &q, st); }
                conv1(t2, x, it.name + ".proj_out", nb * HW, C, C, x);
            } else if (it.kind == 2) {
                Hc *= 2; Wc *= 2;
                conv3(x, t1, it.name + ".conv", nb, Hc, Wc, it.cin, it.cin, 1, nullptr);
                std::swap(x, t1);
            } else {
                Hc /= 2; Wc /= 2;
                conv3(x, t1, it.name + ".conv", nb, Hc, Wc, it.cin, it.cin, 0, nullptr, AMODE_CONV3S2);
                std::swap(x, t1);
            }
        }
    }
};

// VQModel.encode (vq_model.py:41-46) -> min_encoding_indices: img fp32 NCHW [B,3,H,W] (H, W multiples of 16) -> tokens int32 [B, (H/16)(W/16)]
extern "C" int car_vq_encode(car_ctx* c, const float* img, int32_t B, int32_t H, int32_t W, int32_t* out_tokens, void* stream_) {
    if (c && check_sticky(c)) return -1;
    if (!c) return -1;
    if (!c->finalized) FAIL(c, "car_vq_encode: call car_finalize_weights first");
    if (!Wp(c, "encoder.conv_in.weight") || !Wp(c, "quantize.embedding.weight")) FAIL(c, "car_vq_encode: VQ encoder weights were not loaded into this context");
    const car_config& g = c->cfg; const int mode = c->mode; const size_t e = c->esz;
    const int ndown = g.vq_n_mult - 1, div = 1 << ndown;
    if (!img || !out_tokens || B <= 0 || H <= 0 || W <= 0 || H % div || W % div) FAIL(c, "car_vq_encode: bad arguments (H, W must be multiples of %d)", div);
    if (g.codebook_dim > 16) FAIL(c, "car_vq_encode: codebook_embed_dim > 16 unsupported");
    hipStream_t caller = (hipStream_t)stream_, st = c->stream;
    int last_c = 0;
    const std::vector<VqItem> layout = vq_enc_layout(g, &last_c);
    const int hh = H / div, ww = W / div, HW0 = hh * ww, HWp = (int)rup(HW0, 32);
    size_t max_el = (size_t)H * W * g.vq_ch;
    { size_t hw = (size_t)H * W; for (auto& it : layout) { if (it.kind == 3) hw /= 4; size_t cc = it.cin > it.cout ? it.cin : it.cout; if (hw * cc > max_el) max_el = hw * cc; } }
    int CH = B; while (CH > 1 && (size_t)CH * max_el * e * 4 > ((size_t)8 << 30)) CH = (CH + 1) / 2;
    for (int i = 0; i < 4; ++i) NEED(c, c->ws[i], (size_t)CH * max_el * e);
    NEED(c, c->ws[4], (size_t)CH * HW0 * HW0 * 4);
    NEED(c, c->ws[5], (size_t)CH * HW0 * HWp * e);
    NEED(c, c->ws[6], (size_t)CH * last_c * HWp * e);
    NEED(c, c->ws[7], (size_t)CH * 3 * HW0 * last_c * e);
    NEED(c, c->ws[8], (size_t)CH * ((size_t)(H * W + 255) / 256) * 2 * 512 * 4 + 1024);
    NEED(c, c->ws[9], (size_t)CH * 32 * 2 * 4 + 64);
    fence_in(c, caller);
    VqOps ops{c, mode, e, st};
    for (int b0 = 0; b0 < B; b0 += CH) {
        const int nb = (B - b0) < CH ? (B - b0) : CH;
        void *x = c->ws[0].p, *t1 = c->ws[1].p, *t2 = c->ws[2].p, *t3 = c->ws[3].p;
        int Hc = H, Wc = W;
        car_launch_conv_in3(mode, img + (size_t)b0 * 3 * H * W, Wp(c, "encoder.conv_in.weight"), Wp(c, "encoder.conv_in.bias"), x, nb, H, W, g.vq_ch, st);
        ops.blocks(layout, nb, x, t1, t2, t3, Hc, Wc);
        ops.gn(x, t1, "encoder.norm_out", nb, Hc * Wc, last_c, 1);
        ops.conv3(t1, t2, "encoder.conv_out", nb, Hc, Wc, last_c, g.z_channels, 0, nullptr);
        ops.conv1(t2, t3, "quant_conv", nb * Hc * Wc, g.z_channels, g.codebook_dim, nullptr);
        car_launch_vq_argmin(mode, t3, (const float*)Wp(c, "quantize.embedding.weight"), out_tokens + (size_t)b0 * HW0, (long)nb * HW0, g.codebook_dim, g.codebook_size, st);
    }
    fence_out(c, caller);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------- VQ decode
extern "C" int car_vq_decode(car_ctx* c, const int32_t* tokens, int32_t B, int32_t hh, int32_t ww, float* out_nchw, void* stream_) {
    if (c && check_sticky(c)) return -1;
    if (!c) return -1;
    if (!c->finalized) FAIL(c, "car_vq_decode: call car_finalize_weights first");
    if (!Wp(c, "quantize.embedding.weight")) FAIL(c, "car_vq_decode: VQ weights were not loaded into this context");
    if (!tokens || !out_nchw || B <= 0 || hh <= 0 || ww <= 0) FAIL(c, "car_vq_decode: bad arguments");
    const car_config& g = c->cfg; const int mode = c->mode; const size_t e = c->esz;
    hipStream_t caller = (hipStream_t)stream_, st = c->stream;
    int last_c = 0;
    const std::vector<VqItem> layout = vq_layout(g, &last_c);
    const int nup = g.vq_n_mult - 1, Hf = hh << nup, Wf = ww << nup;
    // largest activation (elements per image): track through the layout
    size_t max_el = 0; { int ch = g.vq_ch * g.vq_ch_mult[g.vq_n_mult - 1]; size_t hw = (size_t)hh * ww; max_el = hw * (ch > g.z_channels ? ch : g.z_channels);
        for (auto& it : layout) { if (it.kind == 2) hw *= 4; size_t cc = it.kind == 0 ? (it.cin > it.cout ? it.cin : it.cout) : it.cin; if (hw * cc > max_el) max_el = hw * cc; } }
    // chunk the batch so that ~4 live activation buffers stay below ~8 GiB
    int CH = B; while (CH > 1 && (size_t)CH * max_el * e * 4 > ((size_t)8 << 30)) CH = (CH + 1) / 2;
    const size_t abytes = (size_t)CH * max_el * e;
    for (int i = 0; i < 4; ++i) NEED(c, c->ws[i], abytes);
    const int HW0 = hh * ww, C0 = g.vq_ch * g.vq_ch_mult[g.vq_n_mult - 1];
    const int HWp = (int)rup(HW0, 32);
    NEED(c, c->ws[4], (size_t)CH * HW0 * HW0 * 4);            // attention scores fp32
    NEED(c, c->ws[5], (size_t)CH * HW0 * HWp * e);            // P
    NEED(c, c->ws[6], (size_t)CH * C0 * HWp * e);             // V^T
    NEED(c, c->ws[7], (size_t)CH * 3 * HW0 * C0 * e);         // q, k, v
    NEED(c, c->ws[8], (size_t)CH * ((size_t)(Hf * Wf + 255) / 256) * 2 * 512 * 4 + 1024);   // GN partials (C <= 512)
    NEED(c, c->ws[9], (size_t)CH * 32 * 2 * 4 + 64);          // GN stats
    fence_in(c, caller);
    VqOps ops{c, mode, e, st};
    for (int b0 = 0; b0 < B; b0 += CH) {
        const int nb = (B - b0) < CH ? (B - b0) : CH;
        void *x = c->ws[0].p, *t1 = c->ws[1].p, *t2 = c->ws[2].p, *t3 = c->ws[3].p;
        int Hc = hh, Wc = ww;
        // get_codebook_entry + post_quant_conv (vq_model.py:262-277, :49) -> NHWC
        car_launch_vq_lookup(mode, tokens + (size_t)b0 * HW0, (const float*)Wp(c, "quantize.embedding.weight"), (const float*)Wp(c, "post_quant_conv.weight"),
                             (const float*)Wp(c, "post_quant_conv.bias"), t1, (long)nb * HW0, g.codebook_dim, g.z_channels, g.codebook_size, st);
        ops.conv3(t1, x, "decoder.conv_in", nb, Hc, Wc, g.z_channels, C0, 0, nullptr);
        ops.blocks(layout, nb, x, t1, t2, t3, Hc, Wc);
        ops.gn(x, t1, "decoder.norm_out", nb, Hc * Wc, last_c, 1);
        car_launch_conv_out(mode, t1, Wp(c, "decoder.conv_out.weight"), (const float*)Wp(c, "decoder.conv_out.bias"), out_nchw + (size_t)b0 * 3 * Hf * Wf, nb, Hc, Wc, last_c, st);
    }
    fence_out(c, caller);
    HIPCHK(c, hipGetLastError());
    return 0;
}
