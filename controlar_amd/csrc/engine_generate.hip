// engine_generate.hip — stages D-G: text prefix, control tokens, prefill, the captured decode step (bf16 fast path and the exact fp32 path), car_generate / car_generate_c2i
// (one of the translation units behind include/controlar_hip.h; shared declarations: engine_internal.h)
#include "engine_internal.h"

// ------------------------------------------------------------------------------------- decode step (one token for all b sequences)
struct StepBufs { void *h, *xn, *qkv, *att, *mid, *mid2; float* part; float* logits; int *pos, *step, *cur; };

// A chain = a contiguous slice [b0, b0+bg) of the sequences decoded as its own dependency chain.  With several chains the
// captured step has parallel branches: one chain's HBM-bound attention overlaps the other chains' latency-bound GEMMs
// (each chain re-streams the weights; a layer's 40 MB sits in the 256 MiB MALL between chains).
struct FastBufs { bf16_t *xn, *att, *mid, *q; float* logits; float* attn_part; float* ssq; };     // per-chain scratch (XP-packed activations; ssq: row sums of squares of the residual stream as per-tile partials [rows][dim/16])
struct Grp { int b0, bg, nsplit, attn_variant, attn_lds_pad, attn_pgrid; int *pos, *step; FastBufs fb; SampleP sp; };

// bf16 fast path (decode2.hip): 7 kernels per layer — norm -> wqkv(+RoPE, KV write) -> attention -> wo(+residual) ->
// norm -> w1|w3(+SwiGLU) -> w2(+residual); every linear streams the weights once for all rows of the chain.
// `phase_ev` / `phase_dst` (multi-chain capture): right after this chain's FIRST wqkv the event is recorded and `phase_dst` (the next
// chain's stream) is made to wait for it — the next chain enters the step half a layer late, so that its latency-bound linears run
// under this chain's HBM-bound attention and vice versa (chains forked at the same node run in lockstep: both do their linears at the
// same time, then both their attention, and nothing is hidden).  `prio`: the linears / norms raise their wave priority (s_setprio).
static int enqueue_decode_step_fast(car_ctx* c, const StepBufs& sb, const Grp& gr, int b_total, int SA, int n_tok, bool use_ctrl,
                                    float cs, const unsigned char* maskb, const int* jmin, hipStream_t st,
                                    hipEvent_t phase_ev = nullptr, hipStream_t phase_dst = nullptr, int prio = 0) {
    const car_config& g = c->cfg;
    const int D = g.dim, Hn = g.n_head, Fh = g.ffn_hidden, li = g.n_layer / 3, V = g.vocab_size, T = g.cls_token_num;
    const int b = gr.bg, b0 = gr.b0, nsplit = gr.nsplit;
    const FastBufs& fb = gr.fb;
    const size_t kv_layer = (size_t)b_total * Hn * SA * 64, kv_off = (size_t)b0 * Hn * SA * 64;
    bf16_t* h = (bf16_t*)sb.h + (size_t)b0 * D;
    const bool f8 = g.decode_weight_fp8 != 0;
    int normx_max = 48; bool normx_j4 = false;
    { const char* ev = CAR_KNOB("CAR_NORMX_MAX"); if (ev) normx_max = atoi(ev); ev = CAR_KNOB("CAR_NORMX_J4"); if (ev) normx_j4 = atoi(ev) != 0; }
    int nk = 0, bad_cfg = 0;
    // L2 run-ahead (round 6; chains of one m-block: BASELINE configs 2, 4, 5).  The per-XCD L2 survives a kernel boundary (experiments/xk_cache: a region the same XCD
    // read one kernel earlier streams at L2 speed), and three of a small layer's five kernels — attention, wo, w2 — occupy 40-160 of the 256 CUs.  They carry HELPER
    // workgroups that touch the weights of the kernels that follow, XCD by XCD (decode2_params.h CAR_PF_FIELDS):
    //     attention -> wo + w1|w3 of this layer,   wo -> w2 of this layer,   w2 -> wqkv of the next layer (the last layer: the first 16 MB of the vocabulary projection)
    // so that the HBM stream of a layer's 41 MB runs under its latency-bound kernels.  experiments/lat_probe, 2 rows, position 631: 35.1 -> 32.7 us per layer.
    const bool runahead = b <= 16 && c->n_cu >= 128 && !CAR_KNOB("CAR_NO_RUNAHEAD");
    auto wimg = [&](const std::string& wname, const void*& ptr, unsigned& bytes, size_t cap = (size_t)24 << 20) {
        auto it = c->w.find(wname + (f8 ? "#pk8" : "#pk"));
        if (it == c->w.end()) { ptr = nullptr; bytes = 0; return; }
        ptr = it->second.p; bytes = (unsigned)std::min(it->second.bytes, cap);
    };
    // chains of 17-48 rows (on-the-fly norm: no rmsnorm2 launches to ride on): wo (80-240 workgroups) hosts w1|w3 + w2, w2 hosts the next layer's wqkv
    const bool runahead_nx = b > 16 && b <= normx_max && b == b_total && c->n_cu >= 128 && !CAR_KNOB("CAR_NO_RUNAHEAD");
    auto helpers_for = [&](int main_wgs) -> int {      // helper workgroups beside `main_wgs` workgroups: fill the chip once, multiple of 8 (the helper's XCD arithmetic), at least 32
        if (!(runahead || runahead_nx) || (main_wgs & 7)) return 0;
        int h = ((c->n_cu - main_wgs) / 8) * 8;
        if (h > 192) h = 192;
        return h >= 32 ? h : 0;
    };
    // Mid chains (17-191 rows as ONE chain: BASELINE config 3): the rmsnorm2 launches in front of wqkv / w1|w3 / output are 4-48 workgroups on a 256-CU chip; their
    // helpers pull the weights of the linears behind them into the XCDs' L2s (attention_norm -> wqkv; ffn_norm -> w1|w3 + w2; final norm -> 16 MB of the vocabulary projection)
    const bool runahead_mid = b > 16 && b == b_total && c->n_cu >= 128 && !CAR_KNOB("CAR_NO_RUNAHEAD");
    auto norm_helpers = [&](Norm2P& np, const std::string& w0, const std::string& w1, size_t cap0 = (size_t)24 << 20) {
        if (!runahead_mid) return;
        const int main_wgs = (b + 3) / 4;
        if (main_wgs & 7) return;
        int h = ((c->n_cu - main_wgs) / 8) * 8; if (h > 192) h = 192; if (h < 32) return;
        np.pf_wgs = h; wimg(w0, np.pf_p0, np.pf_b0, cap0); if (!w1.empty()) wimg(w1, np.pf_p1, np.pf_b1);
    };
    // returns the number of sum-of-squares partials per row the kernel leaves in p.ssq_out (0 if it writes none)
    auto gemm = [&](const std::string& wname, const bf16_t* X, int N, int K, int epi, GemmDP gp_) -> int {
        GemmDP p = gp_;
        p.W = (const bf16_t*)Wp(c, wname + (f8 ? "#pk8" : "#pk")); p.X = X; p.M = b; p.N = N; p.K = K;
        p.wscale = f8 ? (const float*)Wp(c, wname + "#sc") : nullptr; p.f8_mfma = g.decode_weight_fp8 == 2;
        int cfg = car_pick_gemm_cfg(b, N, K, epi);
        // 49-64 rows behind an on-the-fly norm: ALL four m-blocks in one workgroup (J = 4), so the row statistics are folded once per weight tile instead of
        // once per (weight tile, m-block) — 240 workgroups of wqkv instead of 960, each re-reading the same 20 KB of partials (profiles/r04_lat_probe_v5_rows64.txt)
        if (p.ssq_in && b > 48 && b <= 64 && normx_j4) cfg = ((epi == EPI_SWIGLU || N >= 6144) ? 200 : 100) + 40 + 1;
        const int I = cfg / 100, J = (cfg / 10) % 10, Mb = (b + 15) / 16;
        p.w_nt = ((Mb + J - 1) / J == 1 ? 1 : 0) | (prio ? 2 : 0);      // bit 0: non-temporal weight stream, bit 1: raised wave priority
        if (p.pf_wgs > 0) { p.pf_wgs = helpers_for((N / (16 * I)) * ((Mb + J - 1) / J)); if (p.pf_wgs > 160) p.pf_wgs = 160; }      // (the caller marks the kernels that host helpers)
        if (p.ssq_out) p.ssq_ld = N / (16 * (I >= 2 ? 2 : 1));
        if (car_launch_dec_gemm_cfg(&p, epi, cfg, st)) bad_cfg = cfg;
        ++nk;
        return p.ssq_out ? p.ssq_ld : 0;
    };
    GemmDP z; memset(&z, 0, sizeof(z));
    // tiny chains (<= 8 rows): the latency-bound regime (BASELINE configs 2, 4, 5).  The two RMSNorms of a layer and the final norm run
    // in the prologue of the GEMM that consumes them (dec_gemm NORM variant), and the attention is ONE launch of 16-wave workgroups
    // (no split-KV partials, no combine kernel): 5 dependent kernels per layer instead of 8.  Measured on MI355X with the layer loop of
    // experiments/small_chain (profiles/r03_small_chain.txt, position 631, us per layer): 2 rows 40.2 -> 35.2, 4 rows 41.6 -> 35.7,
    // 8 rows 47.8 -> 37.4 with 8-wave tiles (one row of the prologue norm per wave); from 12 rows up the fused prologue (every workgroup
    // repeats the norm of all rows) no longer wins (44.1 either way at 12, 50.1 vs 49.4 at 16) and the separate norm kernels stay.
    // The floor of this structure is the kernel boundary itself: 5 EMPTY kernels per layer cost 8.3 us.
    const bool fuse_norm = b <= 8 && D <= 2048 && !CAR_KNOB("CAR_NO_SMALL_FUSE");
    // chains of up to 48 rows (round 4, experiments/lat_probe: profiles/r04_lat_probe_v5_*): the RMSNorm in front of wqkv / w1|w3 / output is applied ON THE FLY.
    // The RESID linear that produced the residual stream (wo, w2) leaves each row's sum of squares as per-tile partials; the consumer folds them into rstd
    // and normalises the bf16 residual rows it loads as its X operand in registers (dec_gemm NORM == 2).  Against the prologue form (<= 8 rows: a barrier-
    // separated norm in front of the main loop, 3.6-6.0 us of a 6-8 us kernel) and against the separate rmsnorm2 kernels (> 8 rows: two dependent launches of
    // ~6 us per layer) the measured layer goes 37.0 -> 34.4 us at 2 rows, 39.4 -> 35.7 at 8, 66.5 -> 61.6 at 32; at 64 rows it is a draw (83.3 / 82.9: the
    // 960 workgroups of wqkv each repeat the row statistics) and at 128 a loss (120 / 125), so larger chains keep the norm kernels.  The first norm of layer 0
    // (token gather) and of the three control-add layers changes the stream before it is normed: those keep the prologue / kernel form.
    const bool normx = b <= normx_max && D % 128 == 0 && D <= 2048 && fb.ssq != nullptr && !CAR_KNOB("CAR_NO_NORMX");      // D/32 and D/16 partials per row: multiples of 4, at most 128 (the fold's 16-byte loads)
    int ssq_np = 0;                                                   // partials per row currently valid in fb.ssq (0: none)
    bf16_t* hc = h;                                                  // the residual stream; ping-pongs with `halt` when a control token is added
    bf16_t* halt = (bf16_t*)sb.xn + (size_t)b0 * D;                  // (the prefill's xn buffer is idle during decode)
    auto normx_fields = [&](GemmDP& q, const std::string& wname) { q.nw = (const bf16_t*)Wp(c, wname); q.neps = g.norm_eps; q.nh_in = hc; q.ssq_in = fb.ssq; q.ssq_np = ssq_np; };
    auto norm_fields = [&](GemmDP& q, const std::string& wname, int l, bool first_of_layer) {
        q.nw = (const bf16_t*)Wp(c, wname); q.neps = g.norm_eps; q.nh_in = hc; q.pos = gr.pos;
        if (first_of_layer && l == 0) { q.nemb = (const bf16_t*)Wp(c, "tok_embeddings.weight"); q.nidx = sb.cur + b0; q.nh_out = h; }
        if (first_of_layer && use_ctrl && l % li == 0 && l / li < 3) {
            q.nadd = 1; q.nctrl = (const bf16_t*)c->ctrl[l / li].p + (size_t)b0 * n_tok * D; q.nT = T; q.n_tok = n_tok; q.ncs = cs;
            q.nh_out = l == 0 ? h : (hc == h ? halt : h);            // never in place: every workgroup re-reads the un-added stream
        }
    };
    for (int l = 0; l < g.n_layer; ++l) {
        const std::string L = "layers." + std::to_string(l) + ".";
        const size_t kvb = g.kv_cache_fp8 ? 1 : 2;          // bytes per cached element (e4m3 / bf16)
        bf16_t* kc = (bf16_t*)((char*)c->kv.p + ((size_t)(2 * l) * kv_layer + kv_off) * kvb); bf16_t* vc = (bf16_t*)((char*)c->kv.p + ((size_t)(2 * l + 1) * kv_layer + kv_off) * kvb);
        const bool special = l == 0 || (use_ctrl && l % li == 0 && l / li < 3);      // the stream changes (gather / control add) before this layer's first norm
        const bool nx1 = normx && !special && ssq_np > 0;
        if (!nx1 && !fuse_norm) {   // [token gather at layer 0] (+ control add at layers 0, n/3, 2n/3) -> h ; attention_norm -> xn (packed)
            Norm2P np; memset(&np, 0, sizeof(np));
            np.h_in = h; np.xn = fb.xn; np.w = (const bf16_t*)Wp(c, L + "attention_norm.weight"); np.D = D; np.eps = g.norm_eps; np.add = prio ? 2 : 0;
            if (l == 0) { np.emb = (const bf16_t*)Wp(c, "tok_embeddings.weight"); np.idx = sb.cur + b0; np.h_out = h; }
            if (use_ctrl && l % li == 0 && l / li < 3) {
                np.add |= 1; np.ctrl = (const bf16_t*)c->ctrl[l / li].p + (size_t)b0 * n_tok * D; np.pos = gr.pos; np.T = T; np.n_tok = n_tok; np.cs = cs; np.h_out = h;
            }
            norm_helpers(np, L + "attention.wqkv.weight", "");
            car_launch_rmsnorm2(&np, b, st); ++nk;
        }
        {
            GemmDP q = z; q.qout = fb.q; q.kc = kc; q.vc = vc; q.rope = c->rope; q.pos = gr.pos; q.H = Hn; q.SA = SA; q.dim = D; q.kv8 = g.kv_cache_fp8 ? 1 : 0;
            if (nx1) normx_fields(q, L + "attention_norm.weight");
            else if (fuse_norm) { norm_fields(q, L + "attention_norm.weight", l, true); }
            gemm(L + "attention.wqkv.weight", fb.xn, 3 * D, D, EPI_QKV, q);
            if (!nx1 && fuse_norm && q.nh_out) hc = q.nh_out;
            if (l == 0 && phase_ev) { (void)hipEventRecord(phase_ev, st); (void)hipStreamWaitEvent(phase_dst, phase_ev, 0); }
        }
        {
            Attn2P ap; memset(&ap, 0, sizeof(ap));
            ap.q = fb.q; ap.kc = kc; ap.vc = vc; ap.pos = gr.pos; ap.mask = maskb ? maskb + (size_t)b0 * T : nullptr; ap.jmin = jmin ? jmin + b0 : nullptr;
            ap.out = fb.att; ap.part = fb.attn_part; ap.H = Hn; ap.SA = SA; ap.T = T; ap.dim = D; ap.nsplit = nsplit; ap.out_packed = 1; ap.kv8 = g.kv_cache_fp8 ? 1 : 0;
            if (gr.attn_pgrid > 0 && nsplit == 1) { ap.n_seq = b; ap.pgrid = gr.attn_pgrid; }
            if (gr.attn_variant == 162) {      // the small-batch kernel hosts the run-ahead for wo and w1|w3
                ap.pf_wgs = helpers_for(b * Hn);
                if (ap.pf_wgs > 0) { wimg(L + "attention.wo.weight", ap.pf_p0, ap.pf_b0); wimg(L + "feed_forward.w13.weight", ap.pf_p1, ap.pf_b1); }
            }
            car_launch_dec_attn2_var(&ap, b, gr.attn_variant, gr.attn_lds_pad, st); nk += nsplit > 1 ? 2 : 1;
        }
        { GemmDP q = z; q.h = hc; if (normx) q.ssq_out = fb.ssq;
          if (runahead) { q.pf_wgs = 1; wimg(L + "feed_forward.w2.weight", q.pf_p0, q.pf_b0); }
          else if (runahead_nx) { q.pf_wgs = 1; wimg(L + "feed_forward.w13.weight", q.pf_p0, q.pf_b0); wimg(L + "feed_forward.w2.weight", q.pf_p1, q.pf_b1); }
          ssq_np = gemm(L + "attention.wo.weight", fb.att, D, D, EPI_RESID, q); }
        const bool nx2 = normx && ssq_np > 0;
        if (!nx2 && !fuse_norm) {
            Norm2P np; memset(&np, 0, sizeof(np));
            np.h_in = h; np.xn = fb.xn; np.w = (const bf16_t*)Wp(c, L + "ffn_norm.weight"); np.D = D; np.eps = g.norm_eps; np.add = prio ? 2 : 0;
            norm_helpers(np, L + "feed_forward.w13.weight", L + "feed_forward.w2.weight");
            car_launch_rmsnorm2(&np, b, st); ++nk;
        }
        { GemmDP q = z; q.outp = fb.mid;
          if (nx2) normx_fields(q, L + "ffn_norm.weight"); else if (fuse_norm) norm_fields(q, L + "ffn_norm.weight", l, false);
          gemm(L + "feed_forward.w13.weight", fb.xn, 2 * Fh, D, EPI_SWIGLU, q); }
        {   // w2 leaves the sums of squares for the next layer's first norm (or the final norm) unless that layer adds a control token first
            const bool next_special = l + 1 < g.n_layer && use_ctrl && (l + 1) % li == 0 && (l + 1) / li < 3;
            GemmDP q = z; q.h = hc; if (normx && !next_special) q.ssq_out = fb.ssq;
            if (runahead || runahead_nx) {
                q.pf_wgs = 1; wimg(l + 1 < g.n_layer ? "layers." + std::to_string(l + 1) + ".attention.wqkv.weight" : std::string("output.weight"), q.pf_p0, q.pf_b0, (size_t)16 << 20);
                // the KV prefixes the next layer's attention will stream — MEASURED, OFF (development switch): the attention shrinks 6.5 -> 4.4 us at 2 rows, but w2 and the two
                // boundaries behind the helpers grow by as much (32.4 vs 32.6 us per layer at position 631, 31.7 vs 32.4 at 200, 36.6 vs 39.7 at 8 rows; only past position
                // ~1000 a gain: 34.8 -> 33.3): every kernel's span is already its own critical path, and a helper load that outlives it is paid at the boundary
                if (l + 1 < g.n_layer && gr.attn_variant == 162 && ((b * Hn) & 7) == 0 && CAR_KNOB("CAR_KV_RUNAHEAD")) {
                    q.pf_kc = (const char*)c->kv.p + ((size_t)(2 * (l + 1)) * kv_layer + kv_off) * kvb; q.pf_vc = (const char*)c->kv.p + ((size_t)(2 * (l + 1) + 1) * kv_layer + kv_off) * kvb;
                    q.pf_pos = gr.pos; q.pf_items = b * Hn; q.pf_SA = SA; q.pf_kvb = (int)kvb;
                }
            }
            ssq_np = gemm(L + "feed_forward.w2.weight", fb.mid, D, Fh, EPI_RESID, q);
        }
    }
    const bool nx3 = normx && ssq_np > 0;
    if (!nx3 && !fuse_norm) {
        Norm2P np; memset(&np, 0, sizeof(np));
        np.h_in = h; np.xn = fb.xn; np.w = (const bf16_t*)Wp(c, "norm.weight"); np.D = D; np.eps = g.norm_eps; np.add = prio ? 2 : 0;
        norm_helpers(np, "output.weight", "", (size_t)16 << 20);
        car_launch_rmsnorm2(&np, b, st); ++nk;
    }
    { GemmDP q = z; q.outf = fb.logits;
      if (nx3) normx_fields(q, "norm.weight"); else if (fuse_norm) norm_fields(q, "norm.weight", g.n_layer, false);
      gemm("output.weight", fb.xn, V, D, EPI_LOGITS, q); }
    car_launch_advance(gr.pos, gr.step, st); ++nk;
    SampleP sp = gr.sp; sp.logits = fb.logits; sp.logits_ks = 0; sp.round_bf16 = 0;
    car_launch_sample_greedy(&sp, st); ++nk;
    c->n_dec_kernels = nk;
    if (bad_cfg) FAIL(c, "decode GEMM: tile configuration %d rejected for this model's dimensions (b=%d, dim=%d, ffn=%d, vocab=%d)", bad_cfg, b, D, Fh, V);
    return 0;
}

// Exact mode (decode_f32.hip), round 5: 5 kernels per layer — wqkv(+on-the-fly attention_norm, RoPE, q scale, K/V rows written at *pos) -> attention (fixed
// 512-position splits folded in one launch) -> wo(+residual) -> w1|w3(+on-the-fly ffn_norm, SwiGLU) -> w2(+residual); layer 0 (token gather) and the three
// control-add layers put one `rmsnorm` launch in its add-only form in front.  Every linear runs on the exact fp32 MFMA over the fragment-packed weights; the
// linears behind a norm multiply the RAW residual rows by the image that carries the norm weight in its columns and scale by rstd in the epilogue (round 4
// ran 8 kernels per layer: two rmsnorm launches and the split-KV combine).  Nothing here depends on the batch except the tile shape, which does not change
// an output's arithmetic: a sequence decodes to the same bits alone and in a batch of 384.
static int enqueue_decode_step(car_ctx* c, const StepBufs& sb, int b_total, int b0, int b, int S_max, int n_tok, int nsplit, bool use_ctrl,
                               float cs, const SampleP& sp_chain, int* pos, int* step, const unsigned char* maskb, hipStream_t st,
                               hipEvent_t phase_ev = nullptr, hipStream_t phase_dst = nullptr) {
    // One chain = rows [b0, b0 + b) of the b_total decoded sequences (every buffer is row-major over the sequences, so a chain is a row offset).  With two
    // chains the captured step has two branches: one chain's attention (HBM-bound) runs beside the other's linears (bound by the fp32 matrix pipe) — different
    // resources, unlike the bf16 step whose linears are latency-bound.  `phase_ev` / `phase_dst`: the next chain enters after this chain's first wqkv.
    const car_config& g = c->cfg; const int mode = c->mode; const size_t e = c->esz;
    const int D = g.dim, Hn = g.n_head, Fh = g.ffn_hidden, li = g.n_layer / 3, V = g.vocab_size, T = g.cls_token_num;
    const size_t kv_layer = (size_t)b_total * Hn * S_max * 64, kv_off = (size_t)b0 * Hn * S_max * 64;
    int nk = 0, bad = 0;
    bool lin_prio = false; { const char* ev = CAR_KNOB("CAR_LINEAR_PRIO"); if (ev) lin_prio = atoi(ev) != 0; }
    auto gemm = [&](const std::string& wname, const void* X, long ldx, int N, int K, int epi, GemmFP q) {
        q.W = (const float*)Wp(c, wname + "#pk32"); q.X = (const float*)X; q.ldx = ldx; q.M = b; q.N = N; q.K = K;
        int cfg = car_pick_gemm_f32_cfg2(b, N, K, epi, b_total / (b > 0 ? b : 1));
        const int J = cfg % 10, Mb = (b + 15) / 16;
        q.w_nt = (cfg < 1000 && (Mb + J - 1) / J == 1 ? 1 : 0) | (lin_prio ? 2 : 0);
        if (!q.W || car_launch_dec_gemm_f32_cfg(&q, epi, cfg, st)) bad = cfg ? cfg : -1;
        ++nk;
    };
    GemmFP z; memset(&z, 0, sizeof(z));
    GemmFP zn = z; zn.normx = 1; zn.neps = g.norm_eps;                  // the linears behind an RMSNorm
    float* h = (float*)sb.h + (size_t)b0 * D; float* att = (float*)sb.att + (size_t)b0 * D;
    float* mid = (float*)sb.mid + (size_t)b0 * Fh; float* logits = sb.logits + (size_t)b0 * V; float* part = sb.part + (size_t)b0 * Hn * nsplit * 66;
    float* qbuf = (float*)sb.qkv + (size_t)b0 * D;                      // [b][H][64] rotated, pre-scaled q (the prefill's qkv buffer is idle during decode)
    for (int l = 0; l < g.n_layer; ++l) {
        const std::string L = "layers." + std::to_string(l) + ".";
        float* kc = (float*)off(c->kv.p, (size_t)(2 * l) * kv_layer + kv_off, e); float* vc = (float*)off(c->kv.p, (size_t)(2 * l + 1) * kv_layer + kv_off, e);
        const bool ctrl_here = use_ctrl && l % li == 0 && l / li < 3;
        if (l == 0 || ctrl_here) {   // token gather (layer 0), control add (layers 0, n/3, 2n/3): the residual stream changes before its norm
            NormP np; memset(&np, 0, sizeof(np));
            np.h_in = h; np.h_out = h; np.xn = nullptr; np.D = D; np.eps = g.norm_eps;
            if (l == 0) { np.emb = Wp(c, "tok_embeddings.weight"); np.idx = sb.cur + b0; }
            if (ctrl_here) { np.add_mode = 1; np.ctrl = off(c->ctrl[l / li].p, (size_t)b0 * n_tok * D, e); np.pos = pos; np.T = T; np.n_tok = n_tok; np.cs = cs; }
            car_launch_rmsnorm(mode, &np, b, st); ++nk;
        }
        { GemmFP q = zn; q.qout = qbuf; q.kc = kc; q.vc = vc; q.rope = c->rope; q.pos = pos; q.H = Hn; q.S_max = S_max; q.dim = D;
          gemm(L + "attention.wqkv.weight", h, D, 3 * D, D, FEPI_QKV, q); }
        if (l == 0 && phase_ev) { (void)hipEventRecord(phase_ev, st); (void)hipStreamWaitEvent(phase_dst, phase_ev, 0); }
        {
            AttnFP ap; memset(&ap, 0, sizeof(ap));
            ap.q = qbuf; ap.kc = kc; ap.vc = vc; ap.pos = pos; ap.mask = maskb ? maskb + (size_t)b0 * T : nullptr; ap.part = part; ap.out = att;
            ap.H = Hn; ap.S_max = S_max; ap.T = T; ap.dim = D; ap.nsplit_max = nsplit;
            const bool fused = (long)Hn * b_total >= 2048 && nsplit <= 8;
            // several chains: 12-wave attention workgroups (two per CU = 24 of its 32 wave slots), so that the other chain's linears find room beside it
            int form = fused ? (b < b_total ? 3 : 1) : 0;
            { const char* ev = CAR_KNOB("CAR_ATTN_F32_FORM"); if (ev && fused) form = atoi(ev); }
            car_launch_dec_attn_f32_ex(&ap, b, form, st); nk += fused ? 1 : 2;
        }
        { GemmFP q = z; q.out = h; q.ldo = D; q.R = h; gemm(L + "attention.wo.weight", att, D, D, D, FEPI_RESID, q); }
        { GemmFP q = zn; q.out = mid; q.ldo = Fh; gemm(L + "feed_forward.w13.weight", h, D, 2 * Fh, D, FEPI_SWIGLU, q); }
        { GemmFP q = z; q.out = h; q.ldo = D; q.R = h; gemm(L + "feed_forward.w2.weight", mid, Fh, D, Fh, FEPI_RESID, q); }
    }
    { GemmFP q = zn; q.out = logits; q.ldo = V; gemm("output.weight", h, D, V, D, FEPI_PLAIN, q); }      // final norm on the fly; fp32 logits (exact mode has no bf16 round)
    car_launch_advance(pos, step, st); ++nk;      // pos = T+i+1 consumed next step; step indexes the token being sampled
    SampleP sp = sp_chain; sp.logits = logits; sp.step_ptr = step; car_launch_sample_greedy(&sp, st); ++nk;
    c->n_dec_kernels = nk;
    if (bad) FAIL(c, "exact-mode decode GEMM: tile configuration %d rejected (b=%d, dim=%d, ffn=%d, vocab=%d: N %% 32 and K %% 16 must be 0)", bad, b, D, Fh, V);
    return 0;
}

// ------------------------------------------------------------------------------------- generate
static int generate_impl(car_ctx* c, const void* text_emb, int32_t text_dtype, const int64_t* labels, const int64_t* emb_mask, int32_t B, int32_t n_new,
                         int32_t use_control, const car_sampling* sp, int32_t* out_tokens, const int32_t* forced_tokens,
                         float* logits_out, void* stream_);

extern "C" int car_generate(car_ctx* c, const void* text_emb, int32_t text_dtype, const int64_t* emb_mask, int32_t B, int32_t n_new,
                            int32_t use_control, const car_sampling* sp, int32_t* out_tokens, const int32_t* forced_tokens,
                            float* logits_out, void* stream_) {
    if (!c) return -1;
    if (c->cfg.model_type != 0) FAIL(c, "car_generate: context was created for the c2i model; use car_generate_c2i");
    if (!text_emb) FAIL(c, "car_generate: bad arguments");
    if (text_dtype != CAR_DT_F32 && text_dtype != CAR_DT_BF16) FAIL(c, "car_generate: text dtype must be F32 or BF16");
    return generate_impl(c, text_emb, text_dtype, nullptr, emb_mask, B, n_new, use_control, sp, out_tokens, forced_tokens, logits_out, stream_);
}

extern "C" int car_generate_c2i(car_ctx* c, const int64_t* labels, int32_t B, int32_t n_new, int32_t use_control, const car_sampling* sp,
                                int32_t* out_tokens, const int32_t* forced_tokens, float* logits_out, void* stream_) {
    if (!c) return -1;
    if (c->cfg.model_type != 1) FAIL(c, "car_generate_c2i: context was created for the t2i model; use car_generate");
    if (!labels) FAIL(c, "car_generate_c2i: bad arguments");
    return generate_impl(c, nullptr, CAR_DT_F32, labels, nullptr, B, n_new, use_control, sp, out_tokens, forced_tokens, logits_out, stream_);
}

static int generate_impl(car_ctx* c, const void* text_emb, int32_t text_dtype, const int64_t* labels, const int64_t* emb_mask, int32_t B, int32_t n_new,
                         int32_t use_control, const car_sampling* sp, int32_t* out_tokens, const int32_t* forced_tokens,
                         float* logits_out, void* stream_) {
    if (check_sticky(c)) return -1;
#ifdef CAR_DEV_KNOBS
    // every switch read below (and by the kernels' launchers) counts into car_stats.dev_knobs_active; the counter runs across ALL entry points and is cleared by car_get_stats
    struct KnobLatch { car_ctx* c; int at_entry; ~KnobLatch() { const int d = g_car_knob_hits - at_entry; if (d > c->knob_hits) c->knob_hits = d; } } knob_latch{c, g_car_knob_hits};
#endif
    if (!c->finalized) FAIL(c, "car_generate: call car_finalize_weights first");
    if (!c->has_gpt) FAIL(c, "car_generate: this context holds VQ weights only");
    if (!sp || !out_tokens || B <= 0 || n_new <= 0) FAIL(c, "car_generate: bad arguments");
    const car_config& g = c->cfg;
    const bool c2i = g.model_type == 1;
    if (sp->sample_logits && g.vocab_size > 32768) FAIL(c, "car_generate: stochastic sampling supports vocab_size <= 32768");
    if (g.vocab_size % 4) FAIL(c, "car_generate: vocab_size must be a multiple of 4");
    const int T = g.cls_token_num;
    if (n_new > g.block_size) FAIL(c, "car_generate: max_new_tokens %d exceeds block_size %d (rope table rows, gpt_t2i.py:454)", n_new, g.block_size);
    if (use_control && (c->ctrl_B != B || c->ctrl_ntok < n_new)) FAIL(c, "car_generate: control tokens cached for B=%d n=%d, requested B=%d n_new=%d", c->ctrl_B, c->ctrl_ntok, B, n_new);
    const bool use_cfg = sp->cfg_scale > 1.0f;
    const int b = use_cfg ? 2 * B : B;
    const float cs = (use_cfg && !c2i) ? sp->control_strength : 1.0f;   // generate.py:87-92: strength ignored when cfg <= 1; absent in gpt.py
    const int S_max = (int)rup(T + n_new, 8);                       // gpt_t2i.py:395
    const int SA = c->mode == CAR_BF16 ? (int)rup(S_max, 32) : S_max;   // fast mode: packed KV streams hold whole 32-position blocks (decode2.hip)
    const int D = g.dim, Hn = g.n_head, Fh = g.ffn_hidden, V = g.vocab_size, n_tok = c->ctrl_ntok, li = g.n_layer / 3;
    const int mode = c->mode; const size_t e = c->esz;
    hipStream_t caller = (hipStream_t)stream_, st = c->stream;
    const int Tpad = (int)rup(T, 32);

    // ---- row layout.  Images are cut into NG groups; the rows of group g are contiguous: [cond rows | uncond rows] under CFG
    // (so every group is a self-contained chain for the decode loop), plain image order otherwise.  NG = 1 reproduces the
    // reference layout [cond 0..B-1 | uncond 0..B-1] (generate.py:158-163).
    const bool fast = mode == CAR_BF16;
    const int mult = use_cfg ? 2 : 1;
    // two chains from 192 sequences up: each chain's GEMMs stream the weights once for <= 128+ rows, and one chain's HBM-bound
    // attention runs beside the other's latency-bound GEMMs (profiles/r02_decode_chain_sweep.txt)
    // (exact mode, round 4: the same cut — its linears are bound by the fp32 matrix pipe and its attention by HBM, so the two chains overlap DIFFERENT resources;
    //  the rows of a chain are computed exactly as in any other batch, so the cut does not touch the mode's batch invariance)
    int NG = b >= 192 ? 2 : 1;
    // exact mode, round 5 (profiles/r05_exact_probe_*.txt, 384 sequences, mean position, ms per step): 1 chain 18.50, 2 chains 18.10, 3 chains 17.42, 4 chains
    // 19.24, 6 chains 21.96.  The attention runs as 12-wave workgroups (decode_f32.hip: 24 of a CU's 32 wave slots), so the other chains' linears are resident
    // beside it; a chain's four linears have to finish while the OTHER chains stream their KV, and with three chains that window is two attentions long.
    if (!fast && b >= 288) NG = 3;
    { const char* ev = CAR_KNOB("CAR_CHAINS"); if (ev) { int v = atoi(ev); if (v >= 1 && v <= 8 && B / v >= 2) NG = v; } }
    if (CAR_KNOB("CAR_SINGLE_CHAIN") || NG > B) NG = 1;
    int img0[9];
    for (int gi = 0; gi <= NG; ++gi) img0[gi] = (int)((long)B * gi / NG);
    std::vector<int> row_img((size_t)b), row_unc((size_t)b);
    for (int gi = 0; gi < NG; ++gi) {
        const int ng = img0[gi + 1] - img0[gi], base = mult * img0[gi];
        for (int j = 0; j < ng; ++j) { row_img[(size_t)base + j] = img0[gi] + j; row_unc[(size_t)base + j] = 0;
                                       if (use_cfg) { row_img[(size_t)base + ng + j] = img0[gi] + j; row_unc[(size_t)base + ng + j] = 1; } }
    }

    // ---- buffers
    const size_t kv_layer = (size_t)b * Hn * SA * 64;
    const size_t kv_cap_before = c->kv.cap;        // ensure() never shrinks: a changed capacity IS a new allocation (the address may repeat)
    const size_t kv_e = (fast && g.kv_cache_fp8) ? 1 : e;        // opt-in e4m3 KV cache: one byte per element
    NEED(c, c->kv, (size_t)g.n_layer * 2 * kv_layer * kv_e);
    const bool kv_fresh = c->kv.cap != kv_cap_before;
    const long rowsP = (long)b * T;
    NEED(c, c->ws[0], (size_t)b * T * g.caption_dim * e);                     // text input (cond | uncond)
    NEED(c, c->ws[1], (size_t)rowsP * D * e);                                 // h (prefill)
    NEED(c, c->ws[2], (size_t)rowsP * D * e);                                 // xn
    NEED(c, c->ws[3], (size_t)rowsP * 3 * D * e);                             // qkv
    const bool pf_flash = use_flash(c, 64);                                   // fused prefill attention: no score / probability tensors
    if (!pf_flash) {
        NEED(c, c->ws[4], (size_t)b * Hn * T * T * 4);                        // S
        NEED(c, c->ws[5], (size_t)b * Hn * T * Tpad * e);                     // P
    }
    NEED(c, c->ws[6], (size_t)b * D * Tpad * e);                              // V^T
    NEED(c, c->ws[7], (size_t)rowsP * (mode == CAR_BF16 ? Fh : 3 * Fh) * e);  // ffn mid (+ interleaved w13 out in exact mode)
    NEED(c, c->ws[8], (size_t)rowsP * D * e);                                 // attention out
    NEED(c, c->ws[9], (size_t)b * V * 4);                                     // logits fp32
    // exact mode: KV splits with boundaries fixed in ABSOLUTE positions (AF_SPLIT rows each), whatever the batch — the split layout fixes the order in
    // which a row's softmax partial sums are folded, so a sequence decodes to the same bits in a batch of 1 and in a batch of 384 (every other
    // exact-mode kernel sums one fixed-order fp32 chain per output): tests/test_parity_gpu.py::test_exact_mode_is_batch_invariant,
    // bench.py --precision fp32 (row 0 = the XL golden).
    int nsplit = fast ? 1 : (S_max + AF_SPLIT - 1) / AF_SPLIT;
    if (fast) { const int wg = b * Hn; while (wg * nsplit < 1024 && nsplit < 16) nsplit *= 2; }
    NEED(c, c->ws[10], (size_t)b * Hn * nsplit * 66 * 4);                     // split-KV partials
    NEED(c, c->ws[11], (size_t)B * (use_control ? n_tok : 1) * D * e);        // condition_mlp output / mlp mid
    // device scalars: [16 ints: (pos, step) of up to 8 chains] [cur_tok: b ints] [jmin: b ints] [jmin_min: 4 ints] [SampleDyn, 16-byte aligned]
    NEED(c, c->scal, (size_t)(16 + 2 * b + 4) * 4 + 16 + sizeof(SampleDyn) + 16);
    NEED(c, c->rowimg, (size_t)b * 4);
    NEED(c, c->tok_out, (size_t)B * n_new * 4);
    NEED(c, c->maskb, (size_t)b * T);
    for (int k = 0; k < 3; ++k) if (use_control) NEED(c, c->ctrl[k], (size_t)b * n_tok * D * e);

    fence_in(c, caller);
    HIPCHK(c, hipEventRecord(c->ev_t0, st));
    // The reference zero-fills fresh KVCache buffers every call (gpt_t2i.py:223-225, :391-405); slots that
    // were never written are always masked there and never read here (the attention kernels walk only valid rows), so
    // no per-call memset is needed.  A FRESH allocation is cleared once: the packed V stream is
    // consumed in 32-position blocks whose tail rows meet a zero probability — they must be finite, not uninitialised bits.
    if (kv_fresh) HIPCHK(c, hipMemsetAsync(c->kv.p, 0, c->kv.cap, st));
    // text-pad mask -> uint8 [b, T] (both CFG halves share it, generate.py:188), built on the device: no host round trip
    c->h_rowimg.assign(row_img.begin(), row_img.end());
    HIPCHK(c, hipMemcpyAsync(c->rowimg.p, c->h_rowimg.data(), (size_t)b * 4, hipMemcpyHostToDevice, st));
    car_launch_build_mask(emb_mask, (const int*)c->rowimg.p, (unsigned char*)c->maskb.p, b, T, st);
    int* pos = (int*)c->scal.p; int* step = pos + 1; int* cur = pos + 16; int* jmin = cur + b; int* jmin_min = jmin + b;
    SampleDyn* dyn = (SampleDyn*)(((uintptr_t)(jmin_min + 4) + 15) & ~(uintptr_t)15);
    c->h_dyn.seed = sp->seed; c->h_dyn.temperature = sp->temperature; c->h_dyn.top_k = sp->top_k; c->h_dyn.top_p = sp->top_p;
    HIPCHK(c, hipMemcpyAsync(dyn, &c->h_dyn, sizeof(SampleDyn), hipMemcpyHostToDevice, st));
    if (emb_mask) car_launch_mask_first_valid((const unsigned char*)c->maskb.p, jmin, b, T, st);
    // ---- prefill window (round 4; SURVEY Appendix E: result-preserving).  Prompts are LEFT-padded (sample_t2i.py:146-160): a pad column is masked for every
    // row of the prefill and for every decode step, so the hidden states and K / V of pad rows influence nothing that is returned — yet the reference (and
    // rounds 1-3 here) pushed all T = 120 rows of every sequence through the 36 layers.  The prefill now runs on the LAST Tv rows only, Tv = the longest valid
    // prompt of the batch rounded up to 8 (8-40 of 120 in the reference's caption statistics): every kernel of the prefill sees sequences of Tv rows, cache rows
    // and rope rows are offset by t0 = T - Tv, the mask by the same columns.  A valid row's arithmetic is unchanged (masked keys contributed exact zeros), so
    // exact-mode tokens stay bit-identical.  Cost: the window size has to be known on the HOST before the prefill is enqueued — from the caller's
    // car_sampling.first_valid_hint (no wait at all), else by ONE 4-byte device-to-host read of the batch minimum of `jmin`, the only host wait of car_generate
    // and only when a mask is given (development build: CAR_NO_PREFILL_WINDOW=1 keeps all T rows and no wait).
    int Tv = T, t0 = 0;
    if (emb_mask && !c2i && T > 8 && T % 8 == 0 && (mode == CAR_BF16 || T <= 128) && !CAR_KNOB("CAR_NO_PREFILL_WINDOW")) {
        int hmin = 0;
        const bool hinted = sp->first_valid_hint > 0;
        if (hinted) {
            // the caller knows a lower bound of the first valid prompt position (it built the mask on the host): nothing is read back, the call only enqueues
            // (the device checks the bound below and raises a sticky flag if valid rows would fall outside the window)
            hmin = sp->first_valid_hint - 1;
        } else {
            // no hint: ONE 4-byte read of the batch minimum of `jmin` — the only host wait of car_generate (it also waits for whatever the caller had queued before)
            car_launch_min_int(jmin, b, jmin_min, -1, nullptr, st);
            HIPCHK(c, hipMemcpyAsync(&hmin, jmin_min, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
        }
        // The window starts on a multiple of 16: a key keeps its slot inside the 16-wide k-blocks of the P·V product (one MFMA chain over the k-blocks in order:
        // leading all-zero blocks add exact zeros), and the row softmax sums a column in the lane of its ABSOLUTE position (softmax_wave_kernel) — so a valid row's
        // K / V rows and logits are the same bits whatever the batch-mates' prompt lengths make of t0 (tests/test_parity_gpu.py::test_exact_mode_prefill_window_invariance).
        const int hm = hmin < 0 ? 0 : (hmin > T ? T : hmin);
        t0 = (hm / 16) * 16; if (t0 > T - 8) t0 = ((T - 8) / 16) * 16; if (t0 < 0) t0 = 0;
        Tv = T - t0;
        if (Tv % 8) { Tv = T; t0 = 0; }
        if (hinted && t0 > 0) {
            if (!c->host_flags) { HIPCHK(c, hipHostMalloc((void**)&c->host_flags, 64, hipHostMallocMapped)); memset(c->host_flags, 0, 64); }
            car_launch_min_int(jmin, b, jmin_min, t0, c->host_flags + 1, st);
        }
    }
    const unsigned char* pmask = (const unsigned char*)c->maskb.p;             // the prefill's mask: [b][Tv]
    if (t0 > 0) {
        NEED(c, c->maskw, (size_t)b * Tv);
        HIPCHK(c, hipMemcpy2DAsync(c->maskw.p, (size_t)Tv, (const unsigned char*)c->maskb.p + t0, (size_t)T, (size_t)Tv, (size_t)b, hipMemcpyDeviceToDevice, st));
        pmask = (const unsigned char*)c->maskw.p;
    }
    const long rowsW = (long)b * Tv;                                           // rows the prefill computes (rowsP = b * T sized the buffers)
    const int Tpw = (int)rup(Tv, 32);
    {
        // profiling aid (tools/pmc_workload.py): start the decode loop `skip` positions late so that a handful of steps under
        // counter collection see a long KV prefix.  The skipped cache rows hold zeros / stale rows: tokens are meaningless.
        int skip = 0; { const char* ev = CAR_KNOB("CAR_DEBUG_SKIP_STEPS"); if (ev) { skip = atoi(ev); if (skip < 0 || skip > n_new - 2) skip = 0; } }
        c->dbg_skip = skip;
        for (int i = 0; i < 8; ++i) { c->h_init[2 * i] = T + skip; c->h_init[2 * i + 1] = skip; }    // (pos, step) per chain: after prefill the first decode step runs at input_pos = T, sampling token index 1
        car_launch_set_pos_step(pos, c->h_init[0], c->h_init[1], st);
    }

    // ---- D. text prefix embed: cls_embedding.cap_proj (gpt_t2i.py:435), uncond rows = uncond_embedding (generate.py:157)
    void *text = c->ws[0].p, *h = c->ws[1].p, *xn = c->ws[2].p, *qkv = c->ws[3].p, *P = c->ws[5].p, *vT = c->ws[6].p, *mid = c->ws[7].p, *att = c->ws[8].p;
    float* S = (float*)c->ws[4].p; float* logits = (float*)c->ws[9].p;
    if (c2i) {
        // LabelEmbedder (gpt.py:89-96): h[b] = embedding_table[label]; CFG rows use the null class num_classes (generate.py:141).
        // The row -> table-index map is built on the device (no host round trip): an out-of-range label is clamped to the null class and
        // raises a sticky device flag that car_get_stats reports (the reference's nn.Embedding fails asynchronously on a GPU as well).
        c->h_rowunc.assign(row_unc.begin(), row_unc.end());
        NEED(c, c->rowunc, (size_t)b * 4 + 16);
        if (!c->host_flags) { HIPCHK(c, hipHostMalloc((void**)&c->host_flags, 64, hipHostMallocMapped)); memset(c->host_flags, 0, 64); }
        HIPCHK(c, hipMemcpyAsync(c->rowunc.p, c->h_rowunc.data(), (size_t)b * 4, hipMemcpyHostToDevice, st));
        int* didx = cur;       // cur_tok[b] is free until the prefill sampler writes it
        car_launch_label_index(labels, (const int*)c->rowimg.p, (const int*)c->rowunc.p, g.num_classes, didx, c->host_flags, b, st);
        car_launch_gather_rows(mode, Wp(c, "cls_embedding.embedding_table.weight"), didx, h, b, D, st);
    } else {
        const long per_src = (long)T * g.caption_dim, per = (long)Tv * g.caption_dim; const size_t ib = text_dtype == CAR_DT_BF16 ? 2 : 4;
        for (int gi = 0; gi < NG; ++gi)
            car_launch_build_text(mode, (const char*)text_emb + (size_t)img0[gi] * per_src * ib, text_dtype, Wp(c, "cls_embedding.uncond_embedding"),
                                  off(text, (size_t)mult * img0[gi] * per, e), img0[gi + 1] - img0[gi], per, per_src, (long)t0 * g.caption_dim, use_cfg, st);
        mlp_tanh(c, text, g.caption_dim, 0, 1, (int)rowsW, g.caption_dim, "cls_embedding.cap_proj.", xn, h, D, st);
    }
    // ---- C. control tokens: condition_mlp then 3 condition_layers, cached for the whole call (gpt_t2i.py:437-442)
    if (use_control) {
        const int Mc = B * n_tok;
        void* ce = c->ws[11].p;
        // scratch for the MLP hidden activations: reuse the (idle) KV area? no — use ws[7]/ws[3] sized for prefill; allocate via ws[4] if needed
        DevBuf& scratch = c->ws[4];
        const size_t s_bytes = pf_flash ? 0 : (size_t)b * Hn * T * T * 4;
        NEED(c, scratch, (size_t)Mc * D * e > s_bytes ? (size_t)Mc * D * e : s_bytes);
        S = (float*)c->ws[4].p;
        mlp_tanh(c, c->ctrl_in.p, D, 0, 1, Mc, D, "condition_mlp.cap_proj.", scratch.p, ce, D, st);
        for (int k = 0; k < 3; ++k) for (int gi = 0; gi < NG; ++gi) {
            const int ng = img0[gi + 1] - img0[gi]; const size_t rows = (size_t)ng * n_tok, base = (size_t)mult * img0[gi] * n_tok;
            mlp_tanh(c, off(ce, (size_t)img0[gi] * n_tok * D, e), D, 0, 1, (int)rows, D, "condition_layers." + std::to_string(k) + ".", scratch.p,
                     off(c->ctrl[k].p, base * D, e), D, st);
            if (use_cfg) HIPCHK(c, hipMemsetAsync(off(c->ctrl[k].p, (base + rows) * D, e), 0, rows * D * e, st));   // uncond rows: MLP(0) = 0 exactly
        }
    }
    // ---- E. prefill over the prefix rows of the window [t0, T) (gpt_t2i.py:446-470)
    for (int l = 0; l < g.n_layer; ++l) {
        const std::string L = "layers." + std::to_string(l) + ".";
        {
            NormP np; memset(&np, 0, sizeof(np));
            np.h_in = h; np.h_out = h; np.xn = xn; np.w = Wp(c, L + "attention_norm.weight"); np.D = D; np.eps = g.norm_eps;
            if (use_control && l % li == 0 && l / li < 3) { np.add_mode = 2; np.ctrl = c->ctrl[l / li].p; np.T = Tv; np.n_tok = n_tok; np.cs = cs; }
            car_launch_rmsnorm(mode, &np, rowsW, st);
        }
        { GemmP q = gp(xn, D, Wp(c, L + "attention.wqkv.weight"), D, qkv, 3 * D, (int)rowsW, 3 * D, D); car_launch_gemm(mode, AMODE_PLAIN, &q, st); }
        if (fast) car_launch_prefill_rope_kv2(qkv, off(c->kv.p, (size_t)(2 * l) * kv_layer, kv_e), off(c->kv.p, (size_t)(2 * l + 1) * kv_layer, kv_e), c->rope, b, Tv, Hn, D, SA, g.kv_cache_fp8 ? 1 : 0, t0, st);
        else car_launch_prefill_rope_kv(mode, qkv, off(c->kv.p, (size_t)(2 * l) * kv_layer, e), off(c->kv.p, (size_t)(2 * l + 1) * kv_layer, e), c->rope, b, Tv, Hn, D, S_max, t0, st);
        car_launch_transpose_pad(mode, off(qkv, (size_t)2 * D, e), 3 * D, (long)Tv * 3 * D, vT, b, Tv, Tpw, D, st);
        bool fused = false;
        if (pf_flash) {
            FlashP f; memset(&f, 0, sizeof(f));
            f.q = (const bf16_t*)qkv; f.k = (const bf16_t*)qkv + D; f.vt = (const bf16_t*)vT; f.o = (bf16_t*)att;
            f.q_sb = f.k_sb = (long)Tv * 3 * D; f.q_st = f.k_st = 3 * D; f.vt_sb = (long)D * Tpw; f.vt_ld = Tpw; f.o_sb = (long)Tv * D; f.o_st = D;
            f.Tq = f.Tk = Tv; f.H = Hn; f.scale = 0.125f; f.mode = 1; f.mask = pmask;
            fused = car_launch_flash64(&f, b, st) == 0;
        }
        if (!fused) {
            GemmP q = gp(qkv, 3 * D, off(qkv, (size_t)D, e), 3 * D, S, Tv, Tv, Tv, 64);
            q.alpha = 0.125f; q.out_f32 = 1; q.nb0 = b; q.nb1 = Hn;
            q.sA0 = (long)Tv * 3 * D; q.sA1 = 64; q.sW0 = (long)Tv * 3 * D; q.sW1 = 64; q.sC0 = (long)Hn * Tv * Tv; q.sC1 = (long)Tv * Tv;
            car_launch_gemm(mode, AMODE_PLAIN, &q, st);
            car_launch_softmax_at(mode, S, Tv, P, Tpw, (long)b * Hn * Tv, Tv, 1, pmask, Tv, Hn, t0, st);
        }
        if (!fused) {
            GemmP q = gp(P, Tpw, vT, Tpw, att, D, Tv, 64, Tpw);
            q.nb0 = b; q.nb1 = Hn;
            q.sA0 = (long)Hn * Tv * Tpw; q.sA1 = (long)Tv * Tpw; q.sW0 = (long)D * Tpw; q.sW1 = (long)64 * Tpw; q.sC0 = (long)Tv * D; q.sC1 = 64;
            car_launch_gemm(mode, AMODE_PLAIN, &q, st);
        }
        { GemmP q = gp(att, D, Wp(c, L + "attention.wo.weight"), D, h, D, (int)rowsW, D, D); q.R = h; q.ldr = D; car_launch_gemm(mode, AMODE_PLAIN, &q, st); }
        { NormP np; memset(&np, 0, sizeof(np)); np.h_in = h; np.xn = xn; np.w = Wp(c, L + "ffn_norm.weight"); np.D = D; np.eps = g.norm_eps; car_launch_rmsnorm(mode, &np, rowsW, st); }
        if (mode == CAR_BF16) {
            GemmP q = gp(xn, D, Wp(c, L + "feed_forward.w13.weight"), D, mid, Fh, (int)rowsW, 2 * Fh, D); q.swiglu = 1; car_launch_gemm(mode, AMODE_PLAIN, &q, st);
        } else {
            void* mid2 = off(mid, (size_t)rowsW * Fh, e);
            GemmP q = gp(xn, D, Wp(c, L + "feed_forward.w13.weight"), D, mid2, 2 * Fh, (int)rowsW, 2 * Fh, D); car_launch_gemm(mode, AMODE_PLAIN, &q, st);
            car_launch_swiglu(mode, mid2, mid, rowsW, Fh, st);
        }
        { GemmP q = gp(mid, Fh, Wp(c, L + "feed_forward.w2.weight"), Fh, h, D, (int)rowsW, D, Fh); q.R = h; q.ldr = D; car_launch_gemm(mode, AMODE_PLAIN, &q, st); }
    }
    // final norm + logits for the LAST prefix row only (generate.py:60 samples logits[:, -1])
    { NormP np; memset(&np, 0, sizeof(np)); np.h_in = h; np.xn = xn; np.w = Wp(c, "norm.weight"); np.D = D; np.eps = g.norm_eps; car_launch_rmsnorm(mode, &np, rowsW, st); }
    { GemmP q = gp(off(xn, (size_t)(Tv - 1) * D, e), (long)Tv * D, Wp(c, "output.weight"), D, logits, V, b, V, D); q.out_f32 = 1; car_launch_gemm(mode, AMODE_PLAIN, &q, st); }
    SampleP spp; memset(&spp, 0, sizeof(spp));
    spp.logits = logits; spp.B = B; spp.V = V; spp.use_cfg = use_cfg; spp.cfg_scale = sp->cfg_scale; spp.cfg_interval = sp->cfg_interval;
    spp.step_ptr = step; spp.n_new = n_new; spp.out_tokens = (int*)c->tok_out.p; spp.cur_tok = cur; spp.forced = forced_tokens; spp.logits_out = logits_out;
    spp.stochastic = sp->sample_logits != 0; spp.temperature = sp->temperature; spp.top_k = sp->top_k; spp.top_p = sp->top_p; spp.seed = sp->seed; spp.row0 = 0;
    spp.dyn = dyn;     // the sampling scalars live in device memory: changing the seed per call does not invalidate the captured graph
    auto group_sampler = [&](int gi) {      // the sampler of group gi: its rows are [cond ng | uncond ng] starting at row mult*img0[gi]
        SampleP q = spp; const int i0 = img0[gi], ng = img0[gi + 1] - i0; const size_t rb = (size_t)mult * i0;
        q.B = ng; q.row0 = i0; q.logits = logits + rb * V; q.out_tokens = (int*)c->tok_out.p + (size_t)i0 * n_new; q.cur_tok = cur + rb;
        q.forced = forced_tokens ? forced_tokens + (size_t)i0 * n_new : nullptr;
        q.logits_out = logits_out ? logits_out + (size_t)i0 * n_new * V : nullptr;
        return q;
    };
    for (int gi = 0; gi < NG; ++gi) { SampleP q = group_sampler(gi); car_launch_sample_greedy(&q, st); }
    HIPCHK(c, hipEventRecord(c->ev_t1, st));

    // ---- F/G. decode loop: one captured step, replayed n_new-1 times (pos/step/token live on the device)
    StepBufs sb;
    sb.h = h; sb.xn = xn; sb.qkv = qkv; sb.att = att; sb.mid = mid; sb.mid2 = off(mid, (size_t)b * Fh, e);
    sb.part = (float*)c->ws[10].p; sb.logits = logits; sb.pos = pos; sb.step = step; sb.cur = cur;
    const int nsteps = n_new - 1 - c->dbg_skip;
    c->stats.graph_used = 0;
    Grp grp[8]; memset(grp, 0, sizeof(grp));
    if (fast) {
        // per-chain scratch from one arena: XP-packed xn / att [Mb*16, D], mid [Mb*16, Fh], q [bg, D] (bf16); logits [bg, V],
        // split-KV partials (fp32).  Every slice is a multiple of 16 bytes.
        size_t tot = 0; size_t sizes[8][7];
        for (int gi = 0; gi < NG; ++gi) {
            Grp& gr = grp[gi];
            gr.b0 = mult * img0[gi]; gr.bg = mult * (img0[gi + 1] - img0[gi]);
            const int bg = gr.bg; const size_t M16 = rup((size_t)bg, 16);
            gr.nsplit = 1; { const int wg = bg * Hn; while (wg * gr.nsplit < 1024 && gr.nsplit < 16) gr.nsplit *= 2; }
            // a handful of sequences (<= 240 (sequence, head) pairs = 12 XL sequences): ONE launch of 16-wave workgroups instead of split-KV + combine —
            // one dependent kernel less per layer.  tools/small_ab.py on MI355X (XL, 1024 tokens, ms per step, same process): 2 rows 1.548 -> 1.406,
            // 8 rows 1.611 -> 1.465, 12 rows 1.887 -> 1.725; at 16 rows the split form wins again (1.890 vs 1.923)  [profiles/r03_small_ab.txt]
            // (round 6: with dec_attn2s — two blocks in flight per wave — the one-launch form wins up to 24 XL sequences: 38.8 -> 38.1 us per layer at 16 rows, 47.5 -> 45.9 at 20,
            //  48.4 -> 46.9 at 24, tools/mid_ab.py with CAR_ONE_LAUNCH_MAX; 480 sixteen-wave workgroups still fit the chip in one round)
            int one_max = T <= 512 ? 480 : 240; { const char* ev = CAR_KNOB("CAR_ONE_LAUNCH_MAX"); if (ev) one_max = atoi(ev); }
            const bool one_launch = (long)bg * Hn <= one_max && !CAR_KNOB("CAR_ATTN_SPLIT_SMALL");
            if (one_launch) gr.nsplit = 1;
            { const char* ev = CAR_KNOB("CAR_ATTN_NSPLIT"); if (ev) { const int v = atoi(ev); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) gr.nsplit = v; } }   // A/B knob: shorter attention workgroups
            // attention variant (decode2.hip; profiles/r02_kbench_*): 4 waves per (sequence, head) from 128 sequences up, 2 below; 16 in the one-launch small form
            // (round 6: the one-launch form is dec_attn2s_kernel, variant 162 — two blocks in flight per wave, bit-identical to 160; it needs T <= 512 and the jmin table)
            gr.attn_variant = (one_launch && gr.nsplit == 1) ? ((T <= 512 && !CAR_KNOB("CAR_ATTN_OLD_SMALL")) ? 162 : 160) : ((gr.nsplit == 1 && bg < 128) ? 20 : 40); gr.attn_lds_pad = 0;
            { const char* ev = CAR_KNOB("CAR_ATTN_VARIANT"); if (ev) gr.attn_variant = atoi(ev); ev = CAR_KNOB("CAR_ATTN_LDS_PAD"); if (ev) gr.attn_lds_pad = atoi(ev); }
            // persistent attention grid: R resident workgroups per CU walk the (sequence, head) items in equal shares
            gr.attn_pgrid = 0;
            { const char* ev = CAR_KNOB("CAR_ATTN_PERSIST"); const int R = ev ? atoi(ev) : 0;
              if (R > 0 && R <= 16 && gr.nsplit == 1) { const long items = (long)bg * Hn, cap = (long)c->n_cu * R;
                  if (items > cap) { const long per = (items + cap - 1) / cap; gr.attn_pgrid = (int)((items + per - 1) / per); } } }
            sizes[gi][0] = M16 * D * 2; sizes[gi][1] = M16 * D * 2; sizes[gi][2] = M16 * Fh * 2; sizes[gi][3] = rup((size_t)bg * D * 2, 16);
            sizes[gi][4] = (size_t)bg * V * 4; sizes[gi][5] = rup((size_t)bg * Hn * gr.nsplit * 66 * 4, 16); sizes[gi][6] = rup(M16 * (size_t)(D / 16) * 4, 16);
            for (int k = 0; k < 7; ++k) tot += sizes[gi][k];
        }
        NEED(c, c->dec_parts, tot);
        char* pbase = (char*)c->dec_parts.p;
        for (int gi = 0; gi < NG; ++gi) {
            Grp& gr = grp[gi]; FastBufs& f = gr.fb;
            f.xn = (bf16_t*)pbase; pbase += sizes[gi][0]; f.att = (bf16_t*)pbase; pbase += sizes[gi][1]; f.mid = (bf16_t*)pbase; pbase += sizes[gi][2];
            f.q = (bf16_t*)pbase; pbase += sizes[gi][3]; f.logits = (float*)pbase; pbase += sizes[gi][4]; f.attn_part = (float*)pbase; pbase += sizes[gi][5]; f.ssq = (float*)pbase; pbase += sizes[gi][6];
            gr.pos = pos + 2 * gi; gr.step = step + 2 * gi;        // scal layout: (pos, step) x 8 chains, then cur_tok[b], then jmin[b]
            gr.sp = group_sampler(gi); gr.sp.step_ptr = gr.step;
            gr.sp.logits = nullptr;     // set per launch to the chain's logits
        }
    }
    if (!fast) {       // exact mode: a chain is a row range of the shared row-major buffers
        for (int gi = 0; gi < NG; ++gi) {
            Grp& gr = grp[gi];
            gr.b0 = mult * img0[gi]; gr.bg = mult * (img0[gi + 1] - img0[gi]); gr.nsplit = nsplit;
            gr.pos = pos + 2 * gi; gr.step = step + 2 * gi;
            gr.sp = group_sampler(gi); gr.sp.step_ptr = gr.step;
        }
    }
    const unsigned char* fmask = emb_mask ? (const unsigned char*)c->maskb.p : nullptr;      // no text-pad mask: nothing to test per position
    const int* fjmin = emb_mask ? jmin : nullptr;
    // ---- decode-loop schedule knobs (fast mode), all OFF by default: the MI355X sweeps of tools/overlap_sweep.py found none of them worth a
    // per cent (profiles/r02_overlap_sweep_v1..v3, DESIGN.md §4 — a linear beside the bandwidth-saturating attention makes no progress whatever
    // the schedule); they stay as A/B switches, and tests/test_parity_gpu.py pins that none of them changes a token.
    //   phase offset : with >= 2 chains, chain g+1 enters the step right after chain g's first wqkv (see enqueue_decode_step_fast)
    //   graph steps  : consecutive tokens captured per graph replay — the chains free-run across them (one fork / join and one phase
    //                  offset per `gsteps` tokens instead of per token); the remainder runs on a single-step graph
    //   linear prio  : s_setprio on the linears / norms
    int phase = fast ? 0 : 1, gsteps = 1, lin_prio = 0;      // exact mode: the second chain enters half a layer late, so that attention meets linears, not attention
    { const char* ev = CAR_KNOB("CAR_PHASE_OFFSET"); if (ev) phase = atoi(ev) != 0; }
    if (fast) {
        const char* ev;
        ev = CAR_KNOB("CAR_GRAPH_STEPS"); if (ev) { const int v = atoi(ev); if (v >= 1 && v <= 64) gsteps = v; }
        ev = CAR_KNOB("CAR_LINEAR_PRIO"); if (ev) lin_prio = atoi(ev) != 0;
    }
    if (NG < 2) phase = 0;
    // ---- exact mode, three chains: the FIRST positions run as ONE chain.  Chains buy overlap of one chain's KV stream with the others' linears at the price of
    // re-streaming the weights per chain and of smaller GEMMs; while the KV prefix is short there is little to overlap (profiles/r05_exact_probe_v5_positions.txt,
    // 384 sequences, ms per step, 1 / 3 chains: position 120 8.47 / 9.25, 220 10.08 / 10.44, 370 13.19 / 12.62, 629 18.50 / 17.45, 1120 28.87 / 27.04).  The
    // cross-over sits where a sequence's KV rows cost ~0.6 of its share of the linears: rows* = 0.087 · P / (8 · dim · n_layer) attended rows (177 for XL).  The
    // host knows the position of every replay, so the loop is two captured graphs; the per-chain (pos, step) scalars are rewritten between them.  Chains are
    // row ranges of the same buffers and every kernel is batch-invariant: the switch does not touch a token.
    int n_early = 0;
    Grp grpE[1]; memset(grpE, 0, sizeof(grpE));
    if (!fast && NG == 3 && mult == 1 && !CAR_KNOB("CAR_CHAINS") && !CAR_KNOB("CAR_NO_EARLY_CHAIN")) {
        const double P = (double)g.n_layer * (4.0 * D * D + 3.0 * (double)D * Fh) + (double)V * D;
        const int rows_star = (int)(0.087 * P / (8.0 * D * g.n_layer));
        const int attended0 = emb_mask ? 24 : T;                       // attended prefix rows at the first decode step (left-padded captions: ~24 of 120 valid on average)
        n_early = rows_star - attended0 - c->dbg_skip;
        if (n_early > nsteps) n_early = nsteps;
        if (n_early < 8) n_early = 0;
        Grp& ge = grpE[0];
        ge.b0 = 0; ge.bg = b; ge.nsplit = nsplit; ge.pos = pos; ge.step = step;
        ge.sp = spp; ge.sp.step_ptr = step;
    }
    int NGc = NG, phasec = phase; Grp* grpc = grp;      // the schedule being enqueued / captured (main: NG chains; early: one chain)
    bool capturing = false;
    int step_rc = 0;
    auto step_one = [&](int gi, hipStream_t sg, hipEvent_t pev, hipStream_t pdst) {
        if (fast) return enqueue_decode_step_fast(c, sb, grpc[gi], b, SA, n_tok, use_control != 0, cs, fmask, fjmin, sg, pev, pdst, lin_prio);
        return enqueue_decode_step(c, sb, b, grpc[gi].b0, grpc[gi].bg, S_max, n_tok, nsplit, use_control != 0, cs, grpc[gi].sp, grpc[gi].pos, grpc[gi].step, fmask, sg, pev, pdst);
    };
    auto enqueue_steps = [&](int k) {      // k consecutive decode steps of every chain
        const int NG = NGc, phase = phasec;
        if (NG >= 2 && capturing) {         // the chains are parallel branches of the captured graph
            if (!phase) {
                (void)hipEventRecord(c->ev_fork, st);
                for (int gi = 1; gi < NG; ++gi) (void)hipStreamWaitEvent(c->streamx[gi - 1], c->ev_fork, 0);
            }
            for (int gi = 0; gi < NG; ++gi) {
                hipStream_t sg = gi == 0 ? st : c->streamx[gi - 1];
                for (int s = 0; s < k; ++s) {
                    const bool hand = phase && s == 0 && gi + 1 < NG;      // chain gi+1's stream joins the capture through this event
                    step_rc |= step_one(gi, sg, hand ? c->ev_phase[gi] : nullptr, hand ? c->streamx[gi] : nullptr);
                }
                if (gi > 0) { (void)hipEventRecord(c->ev_joinx[gi - 1], sg); (void)hipStreamWaitEvent(st, c->ev_joinx[gi - 1], 0); }
            }
        } else {
            for (int s = 0; s < k; ++s)
                for (int gi = 0; gi < NG; ++gi) step_rc |= step_one(gi, st, nullptr, nullptr);
        }
        c->n_dec_kernels *= NG;             // kernel nodes of ONE step over all chains
    };
    if (nsteps > 0) {
        char keyb[640];
        // every scalar and pointer that the captured kernels bake in (n_new: the sampler's row stride and per-chain offsets)
        snprintf(keyb, sizeof(keyb), "%d|%d|%d|%d|%d|%d|%d|%p|%p|%p|%p|%p|%p|%g|%g|%d|%d|%d|%p|%p", b, B, S_max, n_new, n_tok, nsplit, (int)use_control, c->kv.p, h, logits,
                 c->ctrl[0].p, c->maskb.p, c->dec_parts.p ? c->dec_parts.p : c->ws[10].p, (double)cs, (double)sp->cfg_scale, sp->cfg_interval, NG, emb_mask ? 1 : 0,
                 (const void*)forced_tokens, (void*)logits_out);
        { char kb2[240]; snprintf(kb2, sizeof(kb2), "|%d|gen%llu|%p|%p|%p|%p|%d|%d|%d|%d|%d|%d", sp->sample_logits, g_alloc_gen, xn, att, mid, c->scal.p, grp[0].attn_variant, grp[0].attn_lds_pad,
                                  grp[0].nsplit, phase, lin_prio, grp[0].attn_pgrid + 100000 * ((CAR_KNOB("CAR_NO_NORMX") ? 1 : 0) + (CAR_KNOB("CAR_NO_SMALL_FUSE") ? 2 : 0) + (CAR_KNOB("CAR_NO_RUNAHEAD") ? 4 : 0) + (CAR_KNOB("CAR_NO_STAGED_NORMX") ? 8 : 0) + (CAR_KNOB("CAR_KV_RUNAHEAD") ? 16 : 0)) + 1000000 * (CAR_KNOB("CAR_ONE_LAUNCH_MAX") ? atoi(CAR_KNOB("CAR_ONE_LAUNCH_MAX")) : 0));
          strncat(keyb, kb2, sizeof(keyb) - strlen(keyb) - 1); }
        { const char* k1 = CAR_KNOB("CAR_ATTN_F32_FORM"); const char* k2 = CAR_KNOB("CAR_LINEAR_PRIO"); const char* k3 = CAR_KNOB("CAR_NORMX_MAX"); const char* k4 = CAR_KNOB("CAR_NORMX_J4"); char kb3[64]; snprintf(kb3, sizeof(kb3), "|x%s|%s|%s|%s", k1 ? k1 : "-", k2 ? k2 : "-", k3 ? k3 : "-", k4 ? k4 : "-"); strncat(keyb, kb3, sizeof(keyb) - strlen(keyb) - 1); }
        const std::string key(keyb);
        const bool no_graph = CAR_KNOB("CAR_NO_GRAPH") != nullptr;      // profiling aid: eager launches (PMC collection cannot follow graph replays)
        // capture `k` steps into `ex` unless the cached exec already holds exactly this configuration
        auto get_exec = [&](hipGraphExec_t& ex, std::string& exkey, int k, const char* tag = "") -> bool {
            const std::string kk = key + "|k" + std::to_string(k) + tag;
            if (ex && exkey == kk) return true;
            if (ex) { (void)hipGraphExecDestroy(ex); ex = nullptr; exkey.clear(); }
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return false; }
            capturing = true; enqueue_steps(k); capturing = false;
            bool ok = hipStreamEndCapture(st, &graph) == hipSuccess && graph != nullptr;
            if (!ok) (void)hipGetLastError();
            // (per-node priorities were tried — attention low, linears high: hipGraphKernelNodeSetAttribute(hipKernelNodeAttributePriority) is
            //  rejected for every kernel node by HIP 7.2, profiles/r02_small_batch.txt)
            if (ok && hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0) != hipSuccess) { ok = false; ex = nullptr; (void)hipGetLastError(); }
            if (graph) (void)hipGraphDestroy(graph);
            if (ok) exkey = kk;
            return ok;
        };
        const int nmain = nsteps - n_early;
        const int nrep = nmain / gsteps, nrem = nmain % gsteps;
        bool graph_ok = !no_graph;
        auto early = [&](bool on) { if (on) { NGc = 1; phasec = 0; grpc = grpE; } else { NGc = NG; phasec = phase; grpc = grp; } };
        // the scalars of ALL chains at the switch: (pos, step) after n_early steps
        for (int i = 0; i < 8; ++i) { c->h_init2[2 * i] = c->h_init[0] + n_early; c->h_init2[2 * i + 1] = c->h_init[1] + n_early; }
        if (graph_ok && n_early > 0) { early(true); graph_ok = get_exec(c->gexec1, c->gkey1, 1, "|early"); early(false); }      // (exact mode: gexec1 is free, gsteps == 1)
        if (graph_ok && nrep > 0) graph_ok = get_exec(c->gexec, c->gkey, gsteps);
        if (graph_ok && nrem > 0) graph_ok = get_exec(gsteps > 1 ? c->gexec1 : c->gexec, gsteps > 1 ? c->gkey1 : c->gkey, 1);
        if (graph_ok && !step_rc) {
            for (int i = 0; i < n_early; ++i) HIPCHK(c, hipGraphLaunch(c->gexec1, st));
            if (n_early > 0 && nmain > 0) car_launch_set_pos_step(pos, c->h_init2[0], c->h_init2[1], st);
            for (int i = 0; i < nrep; ++i) HIPCHK(c, hipGraphLaunch(c->gexec, st));
            for (int i = 0; i < nrem; ++i) HIPCHK(c, hipGraphLaunch(gsteps > 1 ? c->gexec1 : c->gexec, st));
            c->stats.graph_used = 1;
        } else if (!step_rc) {
            early(true); for (int i = 0; i < n_early; ++i) enqueue_steps(1); early(false);
            if (n_early > 0 && nmain > 0) car_launch_set_pos_step(pos, c->h_init2[0], c->h_init2[1], st);
            for (int i = 0; i < nmain; ++i) enqueue_steps(1);
        }
    }
    if (step_rc) { fence_out(c, caller); return -1; }        // c->err was set by the step builder
    HIPCHK(c, hipEventRecord(c->ev_t2, st));
    HIPCHK(c, hipMemcpyAsync(out_tokens, c->tok_out.p, (size_t)B * n_new * 4, hipMemcpyDeviceToDevice, st));
    fence_out(c, caller);
    HIPCHK(c, hipGetLastError());
    // stats inputs (algorithmic bytes are computed in car_get_stats, which synchronises anyway: DESIGN.md §4 / SURVEY.md §8d)
    {
        c->stats.decode_steps = nsteps;
        c->stats.decode_kernels_per_step = c->n_dec_kernels;
        const double we = (mode == CAR_BF16 && g.decode_weight_fp8) ? 1.0 : (double)e;      // fp8 decode weights: 1 B/param (+ fp32 row scales)
        c->st_wbytes = ((double)g.n_layer * ((double)3 * D * D + (double)D * D + 3.0 * (double)Fh * D) + (double)V * D) * we
                       + ((double)g.n_layer * 2.0 * D + D) * (double)e
                       + ((mode == CAR_BF16 && g.decode_weight_fp8) ? 4.0 * ((double)g.n_layer * (5.0 * D + 2.0 * Fh) + V) : 0.0);
        c->st_b = b; c->st_T = T; c->st_nsteps = nsteps; c->st_has_mask = emb_mask ? 1 : 0; c->st_jmin = jmin;
    }
    return 0;
}
