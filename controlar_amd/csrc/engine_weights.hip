// engine_weights.hip — car_load_tensor (reference state-dict names -> packed device images), car_finalize_weights, the packed-image cache
// (one of the translation units behind include/controlar_hip.h; shared declarations: engine_internal.h)
#include "engine_internal.h"

// ------------------------------------------------------------------------------------- weights

// upload a host fp32 array as element type T (or as fp32 when force_f32)
static int upload(car_ctx* c, const std::string& name, const std::vector<float>& h, const std::vector<int64_t>& shape, bool force_f32 = false) {
    Wt t; t.shape = shape; t.numel = (int64_t)h.size();
    const bool f32 = force_f32 || c->mode == CAR_F32;
    const size_t bytes = h.size() * (f32 ? 4 : 2);
    t.bytes = bytes;
    HIPCHK(c, hipMalloc(&t.p, bytes ? bytes : 4));
    if (f32) { HIPCHK(c, hipMemcpy(t.p, h.data(), bytes, hipMemcpyHostToDevice)); }
    else {
        std::vector<bf16_t> hb(h.size());
        for (size_t i = 0; i < h.size(); ++i) hb[i] = f2bf(h[i]);
        HIPCHK(c, hipMemcpy(t.p, hb.data(), bytes, hipMemcpyHostToDevice));
    }
    auto it = c->w.find(name);
    if (it != c->w.end() && it->second.p) (void)hipFree(it->second.p);
    c->w[name] = t;
    // a re-loaded tensor invalidates its exact-mode fragment image (rebuilt by car_finalize_weights); a re-loaded RMSNorm weight invalidates the image of the
    // linear it is folded into (decode_f32.hip NX: wqkv <- attention_norm, w1|w3 <- ffn_norm, output <- norm)
    std::vector<std::string> stale = {name + "#pk32"};
    auto dep = [&](const char* sfx, const char* lin) { const size_t n = strlen(sfx); if (name.size() >= n && name.compare(name.size() - n, n, sfx) == 0) stale.push_back(name.substr(0, name.size() - n) + lin + "#pk32"); };
    dep("attention_norm.weight", "attention.wqkv.weight"); dep("ffn_norm.weight", "feed_forward.w13.weight");
    if (name == "norm.weight") stale.push_back("output.weight#pk32");
    for (auto& sname : stale) { auto pk = c->w.find(sname); if (pk != c->w.end()) { if (pk->second.p) (void)hipFree(pk->second.p); c->w.erase(pk); } }
    return 0;
}

// fp32 -> OCP e4m3fn (bias 7, max 448, no inf), round-to-nearest-even, saturating
static unsigned char f32_to_e4m3(float f) {
    if (f != f) return 0x7f;
    const unsigned char sign = std::signbit(f) ? 0x80 : 0;
    float a = std::fabs(f);
    if (a >= 464.0f) return sign | 0x7e;                       // beyond the midpoint above 448 (and inf): saturate
    if (a < 0.015625f) {                                        // below 2^-6: subnormal grid of 2^-9
        const int q = (int)std::nearbyint(a * 512.0f);
        return sign | (unsigned char)(q >= 8 ? 0x08 : q);
    }
    int e; const float m = std::frexp(a, &e);                   // a = m * 2^e, m in [0.5, 1)
    int ee = e - 1; float mm = m * 2.0f;                        // a = mm * 2^ee, mm in [1, 2)
    int mant = (int)std::nearbyint((mm - 1.0f) * 8.0f);
    if (mant == 8) { mant = 0; ++ee; }
    if (ee > 8 || (ee == 8 && mant > 6)) return sign | 0x7e;
    return sign | (unsigned char)(((ee + 7) << 3) | mant);
}
static float e4m3_to_f32(unsigned char v) {
    const int e = (v >> 3) & 15, m = v & 7; const float s = (v & 0x80) ? -1.f : 1.f;
    if (e == 15 && m == 7) return NAN;
    return s * (e == 0 ? (float)m * 0.001953125f : std::ldexp(1.0f + (float)m / 8.0f, e - 7));
}
extern "C" int car_debug_f32_to_e4m3(const float* in, unsigned char* out, int64_t n) {      // host-only helper (tests)
    if (!in || !out) return -1;
    for (int64_t i = 0; i < n; ++i) out[i] = f32_to_e4m3(in[i]);
    return 0;
}

// dec_linear weight image: [N/16][K/32] chunks of 64 lanes x 8 bf16 (lane l: row l&15, k (l>>4)*8..+8) — decode.hip
static void pack_decode_bf16(const float* h, int N, int K, bf16_t* pk) {
    const int nkb = K / 32;
    for (int rb = 0; rb < N / 16; ++rb) for (int kb = 0; kb < nkb; ++kb) for (int l = 0; l < 64; ++l) {
        const float* src = &h[(size_t)(rb * 16 + (l & 15)) * K + kb * 32 + (l >> 4) * 8];
        bf16_t* dst = &pk[(((size_t)rb * nkb + kb) * 64 + l) * 8];
        for (int e = 0; e < 8; ++e) dst[e] = f2bf(src[e]);
    }
}
extern "C" int car_debug_pack_decode_weight(const float* w, int32_t N, int32_t K, uint16_t* out) {      // host-only helper (tests)
    if (!w || !out || N <= 0 || K <= 0 || N % 16 || K % 32) return -1;
    pack_decode_bf16(w, N, K, out);
    return 0;
}
// ---- device-side packing of one decode linear (pack.hip).  `name` is the row-major image ([Ntot, K] bf16, the prefill operand:
// for w1 / w3 the 16-row interleaved "w13" image); `src` is the checkpoint tensor [Nsrc, K] in `dtype` (host or device).
// bf16 weights: name#pk = MFMA-fragment image.  fp8 weights: name#pk8 = e4m3 image, name#sc = fp32 row scales, and the
// row-major image holds the DEQUANTISED values so that prefill and decode see one set of effective weights.
static int ensure_w(car_ctx* c, const std::string& name, size_t bytes, const std::vector<int64_t>& shape, int64_t numel) {
    auto it = c->w.find(name);
    if (it != c->w.end() && it->second.p && it->second.bytes == bytes) return 0;
    if (it != c->w.end() && it->second.p) (void)hipFree(it->second.p);
    Wt t; t.shape = shape; t.numel = numel; t.bytes = bytes;
    HIPCHK(c, hipMalloc(&t.p, bytes ? bytes : 4));
    c->w[name] = t;
    return 0;
}
static int dev_linear(car_ctx* c, const std::string& name, const void* src, bool on_dev, int dtype, int Nsrc, int K, int ileave, int Ntot) {
    const bool f8 = c->cfg.decode_weight_fp8 != 0;
    if (Ntot % 16 || K % (f8 ? 64 : 32)) FAIL(c, "%s: decode packing needs N%%16==0 and K%%%d==0 (got %d x %d)", name.c_str(), f8 ? 64 : 32, Ntot, K);
    const size_t eb = dtype == CAR_DT_F32 ? 4 : 2;
    void* stage = nullptr;
    if (!on_dev) {
        HIPCHK(c, hipMalloc(&stage, (size_t)Nsrc * K * eb));
        HIPCHK(c, hipMemcpy(stage, src, (size_t)Nsrc * K * eb, hipMemcpyHostToDevice));
        src = stage;
    }
    const int dt = dtype == CAR_DT_F32 ? 0 : 1;
    int rc = ensure_w(c, name, (size_t)Ntot * K * 2, {Ntot, K}, (int64_t)Ntot * K);
    bool complete = ileave == 0;
    if (!rc && ileave) { int& seen = c->w13_seen[name]; seen |= ileave; complete = seen == 3; }
    if (!rc && !f8) {
        car_launch_rows_to_bf16(src, dt, c->w[name].p, Nsrc, K, ileave, 0);
        if (complete) {
            rc = ensure_w(c, name + "#pk", (size_t)Ntot * K * 2, {Ntot, K}, (int64_t)Ntot * K);
            if (!rc) car_launch_pack_frag_bf16(c->w[name].p, c->w[name + "#pk"].p, Ntot, K, 0);
        }
    } else if (!rc) {
        rc = ensure_w(c, name + "#sc", (size_t)Ntot * 4, {Ntot}, Ntot);
        if (!rc) rc = ensure_w(c, name + "#pk8", (size_t)Ntot * K, {Ntot, K}, (int64_t)Ntot * K);
        if (!rc) {
            car_launch_row_amax_scale(src, dt, (float*)c->w[name + "#sc"].p, Nsrc, K, ileave, 0);
            car_launch_quant_pack_fp8(src, dt, (const float*)c->w[name + "#sc"].p, c->w[name].p, c->w[name + "#pk8"].p, Nsrc, K, ileave, 0);
        }
    }
    if (!rc) { hipError_t e2 = hipStreamSynchronize(0); if (e2 == hipSuccess) e2 = hipGetLastError(); if (e2 != hipSuccess) { c->err = std::string("device packing failed: ") + hipGetErrorString(e2); rc = -1; } }
    if (stage) (void)hipFree(stage);
    return rc;
}

static void replace_all(std::string& s, const std::string& a, const std::string& b) {
    size_t p = 0; while ((p = s.find(a, p)) != std::string::npos) { s.replace(p, a.size(), b); p += b.size(); }
}
// HF ViTModel key names (transformers 5.x "layers.N.attention.q_proj", and the 4.x checkpoint names
// "encoder.layer.N.attention.attention.query / intermediate.dense / output.dense") -> the encoder's canonical names
static std::string canon_name(const std::string& in) {
    if (in.compare(0, 14, "adapter.model.") != 0) return in;
    std::string s = in;
    replace_all(s, "adapter.model.layers.", "adapter.model.encoder.layer.");
    replace_all(s, ".attention.q_proj.", ".attention.attention.query.");
    replace_all(s, ".attention.k_proj.", ".attention.attention.key.");
    replace_all(s, ".attention.v_proj.", ".attention.attention.value.");
    replace_all(s, ".attention.o_proj.", ".attention.output.dense.");
    replace_all(s, ".layernorm_before.", ".norm1.");
    replace_all(s, ".layernorm_after.", ".norm2.");
    replace_all(s, ".intermediate.dense.", ".mlp.fc1.");
    if (s.find(".attention.output.dense.") == std::string::npos) replace_all(s, ".output.dense.", ".mlp.fc2.");
    return s;
}

extern "C" int car_load_tensor(car_ctx* c, const char* cname, const void* ptr, const int64_t* shape, int32_t ndim, int32_t dtype) {
    if (!c || !cname || !ptr || (ndim > 0 && !shape)) { if (c) c->err = "car_load_tensor: null argument"; return -1; }
    if (dtype != CAR_DT_F32 && dtype != CAR_DT_BF16) FAIL(c, "car_load_tensor(%s): dtype must be F32 or BF16", cname);
    const std::string name = canon_name(cname);
    if (name.find("adapter.model.pooler.") == 0 || name == "condition_norm.weight") return 0;   // present in c2i checkpoints, unused on the path
    // reference tensors that the inference path never reads
    if (name == "condition_embeddings.weight" || name == "condition_mlp.uncond_embedding" || ends_with(name, "mask_token") ||
        name == "quantize.codebook_used") return 0;
    std::vector<int64_t> shp(shape, shape + ndim);
    int64_t n = 1; for (auto s : shp) n *= s;
    {
        // fast mode: the five decode linears are packed on the device straight from the checkpoint tensor (pack.hip)
        const car_config& g0 = c->cfg;
        const bool is13 = ends_with(name, "feed_forward.w1.weight") || ends_with(name, "feed_forward.w3.weight");
        const bool islin = ndim == 2 && (ends_with(name, "attention.wqkv.weight") || ends_with(name, "attention.wo.weight") ||
                                         ends_with(name, "feed_forward.w2.weight") || name == "output.weight");
        if (c->mode == CAR_BF16 && (is13 || islin)) {
            hipPointerAttribute_t at; bool on_dev = false;
            if (hipPointerGetAttributes(&at, ptr) == hipSuccess) on_dev = (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged);
            else (void)hipGetLastError();
            c->finalized = false;
            if (is13) {
                if (ndim != 2 || shp[0] != g0.ffn_hidden || shp[1] != g0.dim) FAIL(c, "%s: expected [%d,%d]", cname, g0.ffn_hidden, g0.dim);
                const bool is1 = ends_with(name, "w1.weight");
                const std::string base = name.substr(0, name.size() - strlen("w1.weight"));
                return dev_linear(c, base + "w13.weight", ptr, on_dev, dtype, g0.ffn_hidden, g0.dim, is1 ? 1 : 2, 2 * g0.ffn_hidden);
            }
            return dev_linear(c, name, ptr, on_dev, dtype, (int)shp[0], (int)shp[1], 0, (int)shp[0]);
        }
    }
    // bring to host fp32
    std::vector<float> h((size_t)n);
    {
        hipPointerAttribute_t at; bool on_dev = false;
        if (hipPointerGetAttributes(&at, ptr) == hipSuccess) on_dev = (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged);
        else (void)hipGetLastError();
        const size_t eb = dtype == CAR_DT_F32 ? 4 : 2;
        std::vector<unsigned char> raw;
        const void* src = ptr;
        if (on_dev) { raw.resize((size_t)n * eb); HIPCHK(c, hipMemcpy(raw.data(), ptr, raw.size(), hipMemcpyDeviceToHost)); src = raw.data(); }
        if (dtype == CAR_DT_F32) memcpy(h.data(), src, (size_t)n * 4);
        else { const bf16_t* b = (const bf16_t*)src; for (int64_t i = 0; i < n; ++i) h[(size_t)i] = bf2f(b[i]); }
    }
    const car_config& g = c->cfg;
    c->finalized = false;
    // ---- name-specific packing
    if (starts_with(name, "t5.")) {
        // caption encoder (car_t5_encode).  A full T5 state dict may be offered: the decoder half, lm_head and the tied alias are skipped.
        if (!c->has_t5) FAIL(c, "%s: call car_t5_configure before loading t5.* tensors", cname);
        if (starts_with(name, "t5.decoder.") || starts_with(name, "t5.lm_head.") || name == "t5.encoder.embed_tokens.weight") return 0;
        const car_t5_config& t = c->t5;
        if (ends_with(name, "SelfAttention.relative_attention_bias.weight")) {
            if (ndim != 2 || shp[0] != t.rel_buckets || shp[1] != t.num_heads) FAIL(c, "%s: expected [%d,%d]", cname, t.rel_buckets, t.num_heads);
            if (c->mode == CAR_BF16) for (auto& v : h) v = bf2f(f2bf(v));          // nn.Embedding weight in the model dtype
            c->host_keep[name] = h; c->t5_bias_T = 0; return 0;
        }
        if (ends_with(name, "DenseReluDense.wi_0.weight") || ends_with(name, "DenseReluDense.wi_1.weight")) {
            // wi_0 | wi_1 interleaved in blocks of 16 rows: the gated epilogue sees (gate, value) pairs (same image as w1 | w3)
            if (ndim != 2 || shp[0] != t.d_ff || shp[1] != t.d_model) FAIL(c, "%s: expected [%d,%d]", cname, t.d_ff, t.d_model);
            const bool is0 = ends_with(name, "wi_0.weight");
            const std::string base = name.substr(0, name.size() - strlen("wi_0.weight"));
            const std::string other = base + (is0 ? "wi_1.weight" : "wi_0.weight");
            auto it = c->host_keep.find(other);
            if (it == c->host_keep.end()) { c->host_keep[name] = std::move(h); return 0; }
            const std::vector<float>& w0 = is0 ? h : it->second; const std::vector<float>& w1 = is0 ? it->second : h;
            std::vector<float> pk((size_t)2 * t.d_ff * t.d_model);
            for (int r = 0; r < t.d_ff; ++r) {
                const size_t blk = (size_t)(r / 16) * 32 + (r % 16);
                memcpy(&pk[blk * t.d_model], &w0[(size_t)r * t.d_model], (size_t)t.d_model * 4);
                memcpy(&pk[(blk + 16) * t.d_model], &w1[(size_t)r * t.d_model], (size_t)t.d_model * 4);
            }
            int rc = upload(c, base + "wi.weight", pk, {2 * (int64_t)t.d_ff, t.d_model});
            c->host_keep.erase(other);
            return rc;
        }
        const int inner = t.num_heads * t.d_kv;
        int64_t e0 = -1, e1 = -1;
        if (name == "t5.shared.weight") { e0 = t.vocab_size; e1 = t.d_model; }
        else if (ends_with(name, "SelfAttention.q.weight") || ends_with(name, "SelfAttention.k.weight") || ends_with(name, "SelfAttention.v.weight")) { e0 = inner; e1 = t.d_model; }
        else if (ends_with(name, "SelfAttention.o.weight")) { e0 = t.d_model; e1 = inner; }
        else if (ends_with(name, "DenseReluDense.wo.weight")) { e0 = t.d_model; e1 = t.d_ff; }
        else if (ends_with(name, "layer_norm.weight")) { e0 = t.d_model; }
        else FAIL(c, "%s: not a tensor of the T5 encoder (gated-gelu family)", cname);
        if (shp.empty() || shp[0] != e0 || (e1 >= 0 && (ndim != 2 || shp[1] != e1)) || (e1 < 0 && ndim != 1)) FAIL(c, "%s: unexpected shape", cname);
        return upload(c, name, h, shp);
    }
    if (ends_with(name, "feed_forward.w1.weight") || ends_with(name, "feed_forward.w3.weight")) {
        // w1 | w3 interleaved in blocks of 16 rows so the GEMM epilogue sees (a, c) pairs (gemm.hip SWIGLU)
        if (ndim != 2 || shp[0] != g.ffn_hidden || shp[1] != g.dim) FAIL(c, "%s: expected [%d,%d]", cname, g.ffn_hidden, g.dim);
        const bool is1 = ends_with(name, "w1.weight");
        const std::string base = name.substr(0, name.size() - strlen("w1.weight"));
        const std::string other = base + (is1 ? "w3.weight" : "w1.weight");
        auto it = c->host_keep.find(other);
        if (it == c->host_keep.end()) { c->host_keep[name] = std::move(h); return 0; }
        const std::vector<float>& w1 = is1 ? h : it->second; const std::vector<float>& w3 = is1 ? it->second : h;
        std::vector<float> pk((size_t)2 * g.ffn_hidden * g.dim);
        for (int r = 0; r < g.ffn_hidden; ++r) {
            const size_t blk = (size_t)(r / 16) * 32 + (r % 16);
            memcpy(&pk[blk * g.dim], &w1[(size_t)r * g.dim], (size_t)g.dim * 4);
            memcpy(&pk[(blk + 16) * g.dim], &w3[(size_t)r * g.dim], (size_t)g.dim * 4);
        }
        int rc = upload(c, base + "w13.weight", pk, {2 * (int64_t)g.ffn_hidden, g.dim});        // exact mode only (fast mode: dev_linear above)
        c->host_keep.erase(other);
        return rc;
    }
    if (name == "adapter.model.embeddings.position_embeddings") { c->host_keep[name] = h; return 0; }   // interpolated per resolution
    if (name == "adapter.model.embeddings.patch_embeddings.projection.weight") {
        // [D,3,p,p] -> [D, Kpad] zero padded to a multiple of 32
        const int K = 3 * g.vit_patch * g.vit_patch, Kp = (int)rup(K, 32);
        if (n != (int64_t)g.vit_hidden * K) FAIL(c, "%s: bad shape", cname);
        std::vector<float> pk((size_t)g.vit_hidden * Kp, 0.f);
        for (int d = 0; d < g.vit_hidden; ++d) memcpy(&pk[(size_t)d * Kp], &h[(size_t)d * K], (size_t)K * 4);
        return upload(c, name, pk, {g.vit_hidden, Kp});
    }
    if (name == "quantize.embedding.weight" || starts_with(name, "post_quant_conv.")) return upload(c, name, h, shp, true);
    if (name == "decoder.conv_out.weight") {
        // [3,C,3,3] -> [3][9][C]
        const int C = (int)shp[1];
        std::vector<float> pk(h.size());
        for (int o = 0; o < 3; ++o) for (int ci = 0; ci < C; ++ci) for (int t = 0; t < 9; ++t)
            pk[((size_t)o * 9 + t) * C + ci] = h[((size_t)o * C + ci) * 9 + t];
        return upload(c, name, pk, {3, 9, C});
    }
    if (name == "decoder.conv_out.bias") return upload(c, name, h, shp, true);
    if (name == "encoder.conv_in.weight") return upload(c, name, h, {shp[0], 27});     // [Co,3,3,3] is already (ci, ky, kx)-major
    if ((starts_with(name, "decoder.") || starts_with(name, "encoder.")) && ndim == 4 && shp[2] == 3) {
        // conv3x3 [Co,Ci,3,3] -> implicit-GEMM weight [Co, 9*Ci], k = tap*Ci + ci
        const int Co = (int)shp[0], Ci = (int)shp[1];
        std::vector<float> pk(h.size());
        for (int o = 0; o < Co; ++o) for (int ci = 0; ci < Ci; ++ci) for (int t = 0; t < 9; ++t)
            pk[((size_t)o * 9 + t) * Ci + ci] = h[((size_t)o * Ci + ci) * 9 + t];
        return upload(c, name, pk, {Co, 9 * (int64_t)Ci});
    }
    if ((starts_with(name, "decoder.") || starts_with(name, "encoder.") || starts_with(name, "quant_conv.")) && ndim == 4) return upload(c, name, h, {shp[0], shp[1]});   // 1x1 conv
    return upload(c, name, h, shp);
}

extern "C" int car_finalize_weights(car_ctx* c) {
    if (!c) return -1;
    const car_config& g = c->cfg;
    std::vector<std::string> req = {
        "tok_embeddings.weight", "adapter_mlp.fc1.weight", "adapter_mlp.fc2.weight", "condition_mlp.cap_proj.fc1.weight", "condition_mlp.cap_proj.fc2.weight",
        "norm.weight", "output.weight" };
    if (g.model_type == 1) req.push_back("cls_embedding.embedding_table.weight");
    else for (const char* s : {"cls_embedding.cap_proj.fc1.weight", "cls_embedding.cap_proj.fc2.weight", "cls_embedding.uncond_embedding"}) req.push_back(s);
    for (int k = 0; k < 3; ++k) { req.push_back("condition_layers." + std::to_string(k) + ".fc1.weight"); req.push_back("condition_layers." + std::to_string(k) + ".fc2.weight"); }
    for (int i = 0; i < g.n_layer; ++i) {
        const std::string p = "layers." + std::to_string(i) + ".";
        for (const char* s : {"attention.wqkv.weight", "attention.wo.weight", "feed_forward.w13.weight", "feed_forward.w2.weight", "attention_norm.weight", "ffn_norm.weight"}) req.push_back(p + s);
    }
    const std::string a = "adapter.model.";
    for (const char* s : {"embeddings.cls_token", "embeddings.patch_embeddings.projection.weight", "embeddings.patch_embeddings.projection.bias", "layernorm.weight", "layernorm.bias"}) req.push_back(a + s);
    for (int i = 0; i < g.vit_layers; ++i) {
        const std::string p = a + "encoder.layer." + std::to_string(i) + ".";
        for (const char* s : {"norm1.weight", "norm1.bias", "attention.attention.query.weight", "attention.attention.query.bias", "attention.attention.key.weight",
                              "attention.attention.key.bias", "attention.attention.value.weight", "attention.attention.value.bias", "attention.output.dense.weight",
                              "attention.output.dense.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight",
                              "mlp.fc2.bias"}) req.push_back(p + s);
        if (g.vit_variant == 0) { req.push_back(p + "layer_scale1.lambda1"); req.push_back(p + "layer_scale2.lambda1"); }
    }
    std::string missing;
    int nmiss = 0;
    // a context may serve only decode_code (VQ weights alone) — the reference keeps GPT and VQ as separate modules
    const bool have_t5 = c->has_t5 && Wp(c, "t5.shared.weight");
    const bool vq_only = (Wp(c, "quantize.embedding.weight") || have_t5) && !Wp(c, "tok_embeddings.weight") && !Wp(c, "output.weight");
    c->has_gpt = !vq_only;
    if (have_t5) {       // the caption encoder is optional as a group, complete if present
        std::vector<std::string> tr = {"t5.encoder.final_layer_norm.weight"};
        for (int i = 0; i < c->t5.num_layers; ++i) {
            const std::string p = "t5.encoder.block." + std::to_string(i) + ".layer.";
            for (const char* s : {"0.SelfAttention.q.weight", "0.SelfAttention.k.weight", "0.SelfAttention.v.weight", "0.SelfAttention.o.weight", "0.layer_norm.weight",
                                  "1.DenseReluDense.wi.weight", "1.DenseReluDense.wo.weight", "1.layer_norm.weight"}) tr.push_back(p + s);
        }
        for (auto& r : tr) if (!Wp(c, r)) { if (nmiss < 6) missing += r + " "; ++nmiss; }
        if (c->host_keep.find("t5.encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight") == c->host_keep.end()) {
            missing += "t5.encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight "; ++nmiss; }
    }
    if (!vq_only) {
        for (auto& r : req) if (!Wp(c, r)) { if (nmiss < 6) missing += r + " "; ++nmiss; }
        if (c->host_keep.find("adapter.model.embeddings.position_embeddings") == c->host_keep.end()) { missing += "adapter.model.embeddings.position_embeddings "; ++nmiss; }
    }
    // the VQ decoder is optional as a group (a context may serve generate() only) but must be complete if present
    if (Wp(c, "quantize.embedding.weight")) {
        int last = 0;
        std::vector<std::string> vr = {"post_quant_conv.weight", "post_quant_conv.bias", "decoder.conv_in.weight", "decoder.conv_in.bias",
                                       "decoder.norm_out.weight", "decoder.norm_out.bias", "decoder.conv_out.weight", "decoder.conv_out.bias"};
        for (auto& it : vq_layout(g, &last)) {
            if (it.kind == 0) { for (const char* s : {".norm1.weight", ".norm1.bias", ".conv1.weight", ".conv1.bias", ".norm2.weight", ".norm2.bias", ".conv2.weight", ".conv2.bias"}) vr.push_back(it.name + s);
                                if (it.cin != it.cout) { vr.push_back(it.name + ".nin_shortcut.weight"); vr.push_back(it.name + ".nin_shortcut.bias"); } }
            else if (it.kind == 1) { for (const char* s : {".norm.weight", ".norm.bias", ".q.weight", ".q.bias", ".k.weight", ".k.bias", ".v.weight", ".v.bias", ".proj_out.weight", ".proj_out.bias"}) vr.push_back(it.name + s); }
            else { vr.push_back(it.name + ".conv.weight"); vr.push_back(it.name + ".conv.bias"); }
        }
        if (Wp(c, "encoder.conv_in.weight")) {       // encode side is optional as a group, complete if present
            int el = 0;
            for (const char* s : {"encoder.conv_in.bias", "encoder.norm_out.weight", "encoder.norm_out.bias", "encoder.conv_out.weight", "encoder.conv_out.bias",
                                  "quant_conv.weight", "quant_conv.bias"}) vr.push_back(s);
            for (auto& it : vq_enc_layout(g, &el)) {
                if (it.kind == 0) { for (const char* s : {".norm1.weight", ".norm1.bias", ".conv1.weight", ".conv1.bias", ".norm2.weight", ".norm2.bias", ".conv2.weight", ".conv2.bias"}) vr.push_back(it.name + s);
                                    if (it.cin != it.cout) { vr.push_back(it.name + ".nin_shortcut.weight"); vr.push_back(it.name + ".nin_shortcut.bias"); } }
                else if (it.kind == 1) { for (const char* s : {".norm.weight", ".norm.bias", ".q.weight", ".q.bias", ".k.weight", ".k.bias", ".v.weight", ".v.bias", ".proj_out.weight", ".proj_out.bias"}) vr.push_back(it.name + s); }
                else { vr.push_back(it.name + ".conv.weight"); vr.push_back(it.name + ".conv.bias"); }
            }
        }
        for (auto& r : vr) if (!Wp(c, r)) { if (nmiss < 6) missing += r + " "; ++nmiss; }
    }
    if (!vq_only && c->mode == CAR_BF16) {       // fast mode: every decode linear must have its packed image (both w1 and w3 arrived)
        const char* sfx = g.decode_weight_fp8 ? "#pk8" : "#pk";
        std::vector<std::string> lin = {"output.weight"};
        for (int i = 0; i < g.n_layer; ++i) {
            const std::string p = "layers." + std::to_string(i) + ".";
            for (const char* s : {"attention.wqkv.weight", "attention.wo.weight", "feed_forward.w13.weight", "feed_forward.w2.weight"}) lin.push_back(p + s);
            auto it = c->w13_seen.find(p + "feed_forward.w13.weight");
            if (it != c->w13_seen.end() && it->second != 3) { if (nmiss < 6) missing += p + (it->second == 1 ? "feed_forward.w3.weight " : "feed_forward.w1.weight "); ++nmiss; }
        }
        for (auto& r : lin) if (Wp(c, r) && !Wp(c, r + sfx)) { if (nmiss < 6) missing += r + sfx + " "; ++nmiss; }
    }
    if (nmiss) FAIL(c, "car_finalize_weights: %d required tensors missing, e.g. %s", nmiss, missing.c_str());
    if (!vq_only && c->mode == CAR_F32) {
        // exact mode: the five decode linears also get their fp32 MFMA-fragment image (decode_f32.hip dec_gemm_f32; the row-major copy stays the
        // prefill operand).  Built on the device from the resident row-major tensor, once.
        std::vector<std::string> lin = {"output.weight"};
        for (int i = 0; i < g.n_layer; ++i) {
            const std::string p = "layers." + std::to_string(i) + ".";
            for (const char* s : {"attention.wqkv.weight", "attention.wo.weight", "feed_forward.w13.weight", "feed_forward.w2.weight"}) lin.push_back(p + s);
        }
        for (auto& r : lin) {
            const Wt& src = c->w[r];
            if (src.shape.size() != 2 || src.shape[0] % 16 || src.shape[1] % 16) FAIL(c, "%s: exact-mode decode packing needs N%%16==0 and K%%16==0", r.c_str());
            auto it = c->w.find(r + "#pk32");
            if (it != c->w.end() && it->second.p && it->second.bytes == src.bytes) continue;
            if (ensure_w(c, r + "#pk32", src.bytes, src.shape, src.numel)) return -1;
            // the linears behind an RMSNorm get the norm weight folded into their columns: their decode GEMM multiplies the raw residual rows and applies
            // rstd in its epilogue (decode_f32.hip NX).  The row-major tensor (prefill operand) stays as loaded.
            const void* fold = nullptr;
            auto folded = [&](const char* lin, const char* nrm) { const size_t n = strlen(lin); if (r.size() >= n && r.compare(r.size() - n, n, lin) == 0) fold = Wp(c, r.substr(0, r.size() - n) + nrm); };
            folded("attention.wqkv.weight", "attention_norm.weight"); folded("feed_forward.w13.weight", "ffn_norm.weight");
            if (r == "output.weight") fold = Wp(c, "norm.weight");
            if ((r == "output.weight" || r.find("wqkv") != std::string::npos || r.find("w13") != std::string::npos) && !fold) FAIL(c, "%s: the RMSNorm weight in front of it is missing", r.c_str());
            car_launch_pack_frag_f32(c->w[r].p, c->w[r + "#pk32"].p, src.shape[0], src.shape[1], fold, 0);
        }
        hipError_t e2 = hipStreamSynchronize(0); if (e2 == hipSuccess) e2 = hipGetLastError();
        if (e2 != hipSuccess) FAIL(c, "car_finalize_weights: fp32 fragment packing failed: %s", hipGetErrorString(e2));
    }
    c->finalized = true;
    return 0;
}

// ------------------------------------------------------------------------------------- packed-image cache
// The reference re-reads and re-loads its checkpoints on every start (sample_t2i.py:64-83; demo/model.py:66-75 even per request).
// car_export_packed writes every device-resident weight image of a finalised context (row-major operands, MFMA-fragment / e4m3
// images, scales, conv layouts) plus the host-side tables into one file; car_import_packed restores them with plain copies —
// no conversion, no packing — into a context created with the SAME car_config by the SAME build.  Layout: magic, build id, car_config, entry count, then
// per entry {kind, name, shape, numel, bytes, payload}.  The caller keys the file (controlar_amd/checkpoint.py: content hash).
static const char kPackMagic[8] = {'C', 'A', 'R', 'P', 'K', '0', '3', 0};
static bool same_config(const car_config& a, const car_config& b) {
    car_config x = a, y = b; x.stream_priority = y.stream_priority = 0;
    return memcmp(&x, &y, sizeof(car_config)) == 0;
}
extern "C" int car_export_packed(car_ctx* c, const char* path) {
    if (!c || !path) return -1;
    if (!c->finalized) FAIL(c, "car_export_packed: call car_finalize_weights first");
    (void)hipDeviceSynchronize();
    FILE* f = fopen(path, "wb");
    if (!f) FAIL(c, "car_export_packed: cannot open %s for writing", path);
    char bid[48]; memset(bid, 0, sizeof(bid)); strncpy(bid, car_build_id(), sizeof(bid) - 1);
    bool ok = fwrite(kPackMagic, 1, 8, f) == 8 && fwrite(bid, 1, sizeof(bid), f) == sizeof(bid) && fwrite(&c->cfg, sizeof(car_config), 1, f) == 1;
    const uint64_t n = c->w.size() + c->host_keep.size();
    ok = ok && fwrite(&n, 8, 1, f) == 1;
    std::vector<unsigned char> buf;
    auto put = [&](uint32_t kind, const std::string& name, const std::vector<int64_t>& shape, int64_t numel, const void* data, uint64_t bytes) {
        const uint32_t nl = (uint32_t)name.size(), nd = (uint32_t)shape.size();
        ok = ok && fwrite(&kind, 4, 1, f) == 1 && fwrite(&nl, 4, 1, f) == 1 && fwrite(name.data(), 1, nl, f) == nl && fwrite(&nd, 4, 1, f) == 1;
        if (nd) ok = ok && fwrite(shape.data(), 8, nd, f) == nd;
        ok = ok && fwrite(&numel, 8, 1, f) == 1 && fwrite(&bytes, 8, 1, f) == 1;
        if (bytes) ok = ok && fwrite(data, 1, bytes, f) == bytes;
    };
    for (auto& kv : c->w) {
        const Wt& t = kv.second;
        buf.resize(t.bytes);
        if (t.bytes && hipMemcpy(buf.data(), t.p, t.bytes, hipMemcpyDeviceToHost) != hipSuccess) { fclose(f); FAIL(c, "car_export_packed: device read of %s failed", kv.first.c_str()); }
        put(0, kv.first, t.shape, t.numel, buf.data(), t.bytes);
    }
    for (auto& kv : c->host_keep) put(1, kv.first, {(int64_t)kv.second.size()}, (int64_t)kv.second.size(), kv.second.data(), kv.second.size() * 4);
    ok = (fclose(f) == 0) && ok;
    if (!ok) { remove(path); FAIL(c, "car_export_packed: short write to %s", path); }
    return 0;
}
static int import_packed_impl(car_ctx* c, const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) FAIL(c, "car_import_packed: cannot open %s", path);
    struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
    (void)fseek(f, 0, SEEK_END); const long fsize = ftell(f); (void)fseek(f, 0, SEEK_SET);
    char magic[8], bid[48]; car_config cfg; uint64_t n = 0;
    if (fsize < 0 || fread(magic, 1, 8, f) != 8 || memcmp(magic, kPackMagic, 8) || fread(bid, 1, sizeof(bid), f) != sizeof(bid) || fread(&cfg, sizeof(car_config), 1, f) != 1 || fread(&n, 8, 1, f) != 1)
        FAIL(c, "car_import_packed: %s is not a packed-weight file of this library version", path);
    { char mine[48]; memset(mine, 0, sizeof(mine)); strncpy(mine, car_build_id(), sizeof(mine) - 1);
      if (memcmp(bid, mine, sizeof(mine))) FAIL(c, "car_import_packed: %s was written by a different build of the library (packed layouts are per build)", path); }
    if (!same_config(cfg, c->cfg)) FAIL(c, "car_import_packed: %s was written for a different car_config", path);
    if (n > (1u << 20)) FAIL(c, "car_import_packed: %s is corrupt (entry count)", path);
    // two passes.  Pass 1 reads the entry HEADERS only (names, shapes, payload sizes checked against the file size; payloads are skipped with fseek), so a
    // truncated or corrupt file fails before the context is touched and without holding several GB of payload on the host (eight ranks of a node import at
    // once).  Pass 2 streams one payload at a time through a bounded staging buffer; a read error there clears the images this call had already imported.
    struct Ent { uint32_t kind; std::string name; std::vector<int64_t> shape; int64_t numel; uint64_t bytes; long offset; };
    std::vector<Ent> ents; ents.reserve((size_t)n);
    for (uint64_t i = 0; i < n; ++i) {
        Ent e; uint32_t nl = 0, nd = 0; uint64_t bytes = 0;
        bool ok = fread(&e.kind, 4, 1, f) == 1 && fread(&nl, 4, 1, f) == 1 && nl > 0 && nl < 4096 && e.kind <= 1;
        if (ok) { e.name.assign((size_t)nl, ' '); ok = fread(&e.name[0], 1, nl, f) == nl && fread(&nd, 4, 1, f) == 1 && nd <= 8; }
        if (ok && nd) { e.shape.resize(nd); ok = fread(e.shape.data(), 8, nd, f) == nd; }
        ok = ok && fread(&e.numel, 8, 1, f) == 1 && fread(&bytes, 8, 1, f) == 1;
        const long here = ok ? ftell(f) : -1;
        ok = ok && here >= 0 && e.numel >= 0 && bytes <= (uint64_t)(fsize - here);             // the payload must fit in what is left of the file
        if (ok && e.kind == 1) ok = bytes == (uint64_t)e.numel * 4;                              // host tables are fp32
        if (ok && e.kind == 0) {                                                                 // device images: 1, 2 or 4 bytes per element of the stated shape
            int64_t prod = 1; for (int64_t d : e.shape) { if (d < 0 || (d && prod > INT64_MAX / d)) { ok = false; break; } prod *= d; }
            ok = ok && prod == e.numel && (bytes == (uint64_t)e.numel || bytes == (uint64_t)e.numel * 2 || bytes == (uint64_t)e.numel * 4);
        }
        if (ok) { e.bytes = bytes; e.offset = here; ok = fseek(f, (long)bytes, SEEK_CUR) == 0; }
        if (!ok) FAIL(c, "car_import_packed: %s is truncated or corrupt (entry %llu)", path, (unsigned long long)i);
        ents.push_back(std::move(e));
    }
    std::vector<std::string> done_w, done_h;
    auto undo = [&]() {      // a failure in pass 2: drop what this call imported, so that the context does not hold half a model
        for (auto& nme : done_w) { auto it = c->w.find(nme); if (it != c->w.end()) { if (it->second.p) (void)hipFree(it->second.p); c->w.erase(it); } c->w13_seen.erase(nme); }
        for (auto& nme : done_h) c->host_keep.erase(nme);
        c->finalized = false;
    };
    std::vector<unsigned char> stage;
    const size_t kStage = (size_t)256 << 20;
    for (Ent& e : ents) {
        if (fseek(f, e.offset, SEEK_SET) != 0) { undo(); FAIL(c, "car_import_packed: seek failed in %s", path); }
        if (e.kind == 1) {
            std::vector<float> v((size_t)e.numel);
            if (e.bytes && fread(v.data(), 1, (size_t)e.bytes, f) != e.bytes) { undo(); FAIL(c, "car_import_packed: read error in %s (%s)", path, e.name.c_str()); }
            c->host_keep[e.name] = std::move(v); done_h.push_back(e.name); continue;
        }
        if (ensure_w(c, e.name, (size_t)e.bytes, e.shape, e.numel)) { undo(); return -1; }
        done_w.push_back(e.name);
        for (uint64_t o = 0; o < e.bytes; o += kStage) {
            const size_t nb = (size_t)std::min<uint64_t>(kStage, e.bytes - o);
            if (stage.size() < nb) stage.resize(nb);
            if (fread(stage.data(), 1, nb, f) != nb) { undo(); FAIL(c, "car_import_packed: read error in %s (%s)", path, e.name.c_str()); }
            if (hipMemcpy((char*)c->w[e.name].p + o, stage.data(), nb, hipMemcpyHostToDevice) != hipSuccess) { undo(); FAIL(c, "car_import_packed: upload of %s failed", e.name.c_str()); }
        }
        if (ends_with(e.name, "feed_forward.w13.weight")) c->w13_seen[e.name] = 3;
    }
    c->finalized = false;
    return car_finalize_weights(c);       // names / shapes are checked against the config there: a missing image fails the import
}
extern "C" int car_import_packed(car_ctx* c, const char* path) {
    if (!c || !path) return -1;
    try { return import_packed_impl(c, path); }
    catch (const std::exception& ex) { c->err = std::string("car_import_packed: ") + ex.what(); return -1; }   // no C++ exception crosses the C ABI
}
