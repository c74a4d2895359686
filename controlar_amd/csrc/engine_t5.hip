// engine_t5.hip — the Flan-T5 caption encoder (car_t5_configure / car_t5_encode)
// (one of the translation units behind include/controlar_hip.h; shared declarations: engine_internal.h)
#include "engine_internal.h"

// ------------------------------------------------------------------------------------- caption encoder
// HF T5EncoderModel as the reference builds it (language/t5.py:58-79) and calls it (:185-201): T5Stack of
// [T5LayerSelfAttention, T5LayerFF] blocks (modeling_t5.py), pre-RMSNorm residual layout, no biases anywhere, relative position
// bias of block 0 shared by every block, attention scaling 1.0, gated tanh-GELU feed-forward, final RMSNorm.
extern "C" int car_t5_configure(car_ctx* c, const car_t5_config* t) {
    if (!c || !t) { if (c) c->err = "car_t5_configure: null argument"; return -1; }
    if (t->vocab_size <= 0 || t->d_model <= 0 || t->d_kv <= 0 || t->num_heads <= 0 || t->d_ff <= 0 || t->num_layers <= 0 || t->rel_buckets < 4 ||
        t->rel_max_distance <= 0 || !(t->ln_eps > 0.f)) FAIL(c, "car_t5_configure: non-positive field");
    if (t->d_model % 32 || t->d_kv % 32 || t->d_ff % 32 || t->d_model > 16384) FAIL(c, "car_t5_configure: d_model, d_kv, d_ff must be multiples of 32 (d_model <= 16384)");
    if (t->rel_buckets % 4) FAIL(c, "car_t5_configure: rel_buckets must be a multiple of 4");
    c->t5 = *t; c->has_t5 = true; c->t5_bias_T = 0;
    return 0;
}

// T5Attention._relative_position_bucket, bidirectional (modeling_t5.py): rel = key - query
static int t5_bucket(int rel, int nb, int max_distance) {
    int b = 0; const int n = nb / 2;
    if (rel > 0) b += n;
    const int a = rel < 0 ? -rel : rel, max_exact = n / 2;
    if (a < max_exact) return b + a;
    int v = max_exact + (int)(std::log((double)a / max_exact) / std::log((double)max_distance / max_exact) * (n - max_exact));
    if (v > n - 1) v = n - 1;
    return b + v;
}

extern "C" int car_t5_encode(car_ctx* c, const int64_t* input_ids, const int64_t* attention_mask, int32_t B, int32_t T, void* out, void* stream_) {
    if (!c) return -1;
    if (check_sticky(c)) return -1;
    if (!c->has_t5 || !c->finalized || !Wp(c, "t5.shared.weight")) FAIL(c, "car_t5_encode: T5 weights not loaded / finalised");
    if (!input_ids || !out || B <= 0 || T <= 0) FAIL(c, "car_t5_encode: bad arguments");
    const car_t5_config& t = c->t5;
    const int mode = c->mode; const size_t e = c->esz;
    const int D = t.d_model, nh = t.num_heads, hd = t.d_kv, inner = nh * hd, F = t.d_ff, Tpad = (int)rup(T, 32);
    hipStream_t caller = (hipStream_t)stream_, st = c->stream;
    // position bias [heads][T][T] for this T (compute_bias): table[bucket(j - i)][h]
    if (c->t5_bias_T != T) {
        const std::vector<float>& tab = c->host_keep["t5.encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"];
        std::vector<float> hb((size_t)nh * T * T);
        std::vector<int> bk(2 * T - 1);
        for (int r = -(T - 1); r <= T - 1; ++r) bk[r + T - 1] = t5_bucket(r, t.rel_buckets, t.rel_max_distance);
        for (int h = 0; h < nh; ++h) for (int i = 0; i < T; ++i) for (int j = 0; j < T; ++j)
            hb[((size_t)h * T + i) * T + j] = tab[(size_t)bk[j - i + T - 1] * nh + h];
        HIPCHK(c, hipStreamSynchronize(st));
        NEED(c, c->t5_bias, hb.size() * 4);
        HIPCHK(c, hipMemcpy(c->t5_bias.p, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        c->t5_bias_T = T;
    }
    const long n_tok = (long)B * T;
    // ids int32 | mask uint8 | int64 staging for host-side inputs
    const size_t o_mk = rup((size_t)n_tok * 4, 256), o_st = o_mk + rup((size_t)n_tok, 256);
    NEED(c, c->t5_in, o_st + 2 * (size_t)n_tok * 8);
    int* ids32 = (int*)c->t5_in.p; unsigned char* mk = (unsigned char*)c->t5_in.p + o_mk; long long* stage = (long long*)((char*)c->t5_in.p + o_st);
    int CH = B; if (CH > 64) CH = 64;
    const bool flash = use_flash(c, hd);
    const long rows_max = (long)CH * T;
    NEED(c, c->ws[1], (size_t)n_tok * D * e);                 // h (all rows: gathered up front)
    NEED(c, c->ws[2], (size_t)rows_max * D * e);              // xn
    NEED(c, c->ws[3], (size_t)rows_max * 3 * inner * e);      // q | k | v planes
    NEED(c, c->ws[4], (size_t)CH * nh * T * T * 4);           // S fp32
    NEED(c, c->ws[5], (size_t)CH * nh * T * Tpad * e);        // P
    NEED(c, c->ws[6], (size_t)CH * inner * Tpad * e);         // V^T
    NEED(c, c->ws[7], (size_t)rows_max * F * e);              // gated mid
    NEED(c, c->ws[8], (size_t)rows_max * inner * e);          // ctx
    if (mode == CAR_F32) NEED(c, c->ws[9], (size_t)rows_max * 2 * F * e);   // exact mode: wi_0 | wi_1 outputs before the gate
    fence_in(c, caller);
    auto on_device = [](const void* p) { hipPointerAttribute_t at; if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
                                         return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged; };
    const long long* d_ids = (const long long*)input_ids; const long long* d_mask = (const long long*)attention_mask;
    if (!on_device(input_ids)) { HIPCHK(c, hipMemcpyAsync(stage, input_ids, (size_t)n_tok * 8, hipMemcpyHostToDevice, st)); d_ids = stage; }
    if (attention_mask && !on_device(attention_mask)) { HIPCHK(c, hipMemcpyAsync(stage + n_tok, attention_mask, (size_t)n_tok * 8, hipMemcpyHostToDevice, st)); d_mask = stage + n_tok; }
    car_launch_t5_prep(d_ids, d_mask, ids32, mk, n_tok, t.vocab_size, st);
    car_launch_gather_rows(mode, Wp(c, "t5.shared.weight"), ids32, c->ws[1].p, n_tok, D, st);
    const float* bias = (const float*)c->t5_bias.p;
    for (int b0 = 0; b0 < B; b0 += CH) {
        const int nb = (B - b0) < CH ? (B - b0) : CH;
        const long rows = (long)nb * T;
        void *h = off(c->ws[1].p, (size_t)b0 * T * D, e), *xn = c->ws[2].p, *qkv = c->ws[3].p, *P = c->ws[5].p, *vT = c->ws[6].p, *mid = c->ws[7].p, *ctx = c->ws[8].p;
        float* S = (float*)c->ws[4].p;
        void* qp = qkv; void* kp = off(qkv, (size_t)rows * inner, e); void* vp = off(qkv, (size_t)2 * rows * inner, e);
        auto norm = [&](const std::string& w, void* dst) {
            NormP np; memset(&np, 0, sizeof(np)); np.h_in = h; np.xn = dst; np.w = Wp(c, w); np.D = D; np.eps = t.ln_eps;
            car_launch_rmsnorm(mode, &np, rows, st);
        };
        for (int l = 0; l < t.num_layers; ++l) {
            const std::string L = "t5.encoder.block." + std::to_string(l) + ".layer.";
            norm(L + "0.layer_norm.weight", xn);
            const char* names[3] = {"q", "k", "v"}; void* dst[3] = {qp, kp, vp};
            for (int k = 0; k < 3; ++k) {
                GemmP q = gp(xn, D, Wp(c, L + "0.SelfAttention." + names[k] + ".weight"), D, dst[k], inner, (int)rows, inner, D);
                car_launch_gemm(mode, AMODE_PLAIN, &q, st);
            }
            car_launch_transpose_pad(mode, vp, inner, (long)T * inner, vT, nb, T, Tpad, inner, st);
            bool fused = false;
            if (flash) {
                FlashP f; memset(&f, 0, sizeof(f));
                f.q = (const bf16_t*)qp; f.k = (const bf16_t*)kp; f.vt = (const bf16_t*)vT; f.o = (bf16_t*)ctx;
                f.q_sb = f.k_sb = f.o_sb = (long)T * inner; f.q_st = f.k_st = f.o_st = inner; f.vt_sb = (long)inner * Tpad; f.vt_ld = Tpad;
                f.Tq = f.Tk = T; f.H = nh; f.scale = 1.0f; f.mode = 2; f.mask = mk + (size_t)b0 * T; f.bias = bias;
                fused = car_launch_flash64(&f, nb, st) == 0;
            }
            if (!fused) {   // scores[b,h] = Q K^T (scaling 1.0)
                GemmP q = gp(qp, inner, kp, inner, S, T, T, T, hd);
                q.out_f32 = 1; q.nb0 = nb; q.nb1 = nh;
                q.sA0 = (long)T * inner; q.sA1 = hd; q.sW0 = (long)T * inner; q.sW1 = hd; q.sC0 = (long)nh * T * T; q.sC1 = (long)T * T;
                car_launch_gemm(mode, AMODE_PLAIN, &q, st);
                car_launch_t5_softmax(mode, S, T, P, Tpad, (long)nb * nh * T, T, bias, mk + (size_t)b0 * T, T, nh, st);
            }
            if (!fused) {   // ctx[b, t, h*hd + d] = P[b,h] @ V[b,h]
                GemmP q = gp(P, Tpad, vT, Tpad, ctx, inner, T, hd, Tpad);
                q.nb0 = nb; q.nb1 = nh;
                q.sA0 = (long)nh * T * Tpad; q.sA1 = (long)T * Tpad; q.sW0 = (long)inner * Tpad; q.sW1 = (long)hd * Tpad; q.sC0 = (long)T * inner; q.sC1 = hd;
                car_launch_gemm(mode, AMODE_PLAIN, &q, st);
            }
            {   // h = h + o(ctx)   (T5LayerSelfAttention)
                GemmP q = gp(ctx, inner, Wp(c, L + "0.SelfAttention.o.weight"), inner, h, D, (int)rows, D, inner);
                q.R = h; q.ldr = D;
                car_launch_gemm(mode, AMODE_PLAIN, &q, st);
            }
            norm(L + "1.layer_norm.weight", xn);
            if (mode == CAR_BF16) {   // mid = gelu_new(wi_0 x) * wi_1 x in the GEMM epilogue
                GemmP q = gp(xn, D, Wp(c, L + "1.DenseReluDense.wi.weight"), D, mid, F, (int)rows, 2 * F, D); q.swiglu = 2;
                car_launch_gemm(mode, AMODE_PLAIN, &q, st);
            } else {
                GemmP q = gp(xn, D, Wp(c, L + "1.DenseReluDense.wi.weight"), D, c->ws[9].p, 2 * F, (int)rows, 2 * F, D);
                car_launch_gemm(mode, AMODE_PLAIN, &q, st);
                car_launch_t5_gated_act(mode, c->ws[9].p, mid, rows, F, st);
            }
            {   // h = h + wo(mid)   (T5LayerFF)
                GemmP q = gp(mid, F, Wp(c, L + "1.DenseReluDense.wo.weight"), F, h, D, (int)rows, D, F);
                q.R = h; q.ldr = D;
                car_launch_gemm(mode, AMODE_PLAIN, &q, st);
            }
        }
        norm("t5.encoder.final_layer_norm.weight", off(out, (size_t)b0 * T * D, e));
    }
    fence_out(c, caller);
    HIPCHK(c, hipGetLastError());
    return 0;
}
