// engine.hip — context lifecycle, statistics and the standalone sampler behind the C ABI of include/controlar_hip.h; the stages live in
// engine_weights.hip / engine_encode.hip / engine_generate.hip / engine_t5.hip / engine_vq.hip (shared declarations: engine_internal.h).
//
// Stages:  A resize+patchify -> B DINOv2 -> C control MLPs -> D text embed ->
// E prefill -> F decode loop (hipGraph-captured step replayed n_new-1 times, position/token fed
// back on device) -> G CFG+sampling -> H VQ decode.  Everything below is enqueued on the context's
// own stream (graph capture is illegal on the legacy default stream torch uses by default) and
// fenced against the caller's stream with events — no host synchronisation inside the token loop.
#include "engine_internal.h"
#ifdef CAR_DEV_KNOBS
extern "C" { int g_car_knob_hits = 0; }
#endif

static thread_local std::string g_create_err;
unsigned long long g_alloc_gen = 0;

// ------------------------------------------------------------------------------------- lifecycle
extern "C" int car_abi_version(void) { return CAR_ABI_VERSION; }
// hash of every source of this library (build.sh -> _obj/build_id.h): the packed-weight images are a private format of ONE build,
// so the cache file header and the cache key (controlar_amd/checkpoint.py content_key) both carry it
#include "_obj/build_id.h"
extern "C" const char* car_build_id(void) { return CAR_BUILD_ID; }

extern "C" const char* car_last_error(const car_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int car_create(car_ctx** out, const car_config* cfg) {
    if (!out || !cfg) { g_create_err = "car_create: null argument"; return -1; }
    *out = nullptr;
    if (cfg->abi_version != CAR_ABI_VERSION) { g_create_err = "car_create: abi_version mismatch"; return -1; }
    if (cfg->mode != CAR_F32 && cfg->mode != CAR_BF16) { g_create_err = "car_create: mode must be CAR_F32 or CAR_BF16"; return -1; }
    if (cfg->dim <= 0 || cfg->n_layer < 3 || cfg->n_head <= 0 || cfg->ffn_hidden <= 0 || cfg->vocab_size <= 0 || cfg->cls_token_num <= 0 ||
        cfg->block_size <= 0 || cfg->caption_dim <= 0 || cfg->vit_hidden <= 0 || cfg->vit_layers <= 0 || cfg->vit_heads <= 0 || cfg->vit_mlp <= 0 ||
        cfg->vit_patch <= 0 || cfg->vit_pos_grid <= 0 || cfg->vq_n_mult < 1 || cfg->vq_n_mult > 8) {
        g_create_err = "car_create: non-positive dimension in car_config"; return -1; }
    if (cfg->dim % cfg->n_head != 0 || cfg->dim / cfg->n_head != 64) { g_create_err = "car_create: head_dim must be 64 (every LlamaGen size)"; return -1; }
    if (cfg->dim % 32 || cfg->ffn_hidden % 32 || cfg->caption_dim % 32 || cfg->vit_hidden % 32 || cfg->vit_mlp % 32) {
        g_create_err = "car_create: dim, ffn_hidden, caption_dim, vit_hidden, vit_mlp must be multiples of 32"; return -1; }
    if (cfg->decode_weight_fp8 && (cfg->mode != CAR_BF16 || cfg->dim % 64 || cfg->ffn_hidden % 64)) {
        g_create_err = "car_create: decode_weight_fp8 needs CAR_BF16 mode and dim, ffn_hidden multiples of 64"; return -1; }
    if (cfg->kv_cache_fp8 && cfg->mode != CAR_BF16) { g_create_err = "car_create: kv_cache_fp8 exists only in CAR_BF16 mode (the exact mode is the parity path)"; return -1; }
    if (cfg->vit_hidden % cfg->vit_heads != 0 || (cfg->vit_hidden / cfg->vit_heads) % 32) { g_create_err = "car_create: ViT head_dim must be a multiple of 32"; return -1; }
    int g = (int)std::lround(std::sqrt((double)cfg->block_size));
    if (g * g != cfg->block_size) { g_create_err = "car_create: block_size must be a square (gpt_t2i.py:352)"; return -1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_err = "car_create: no HIP device visible (this library has no CPU fallback)"; return -1; }
    car_ctx* c = new car_ctx();
    c->cfg = *cfg; c->mode = cfg->mode; c->esz = cfg->mode == CAR_BF16 ? 2 : 4;
    { int dev = 0, ncu = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) c->n_cu = ncu; else (void)hipGetLastError(); }
    memset(&c->stats, 0, sizeof(c->stats));
    int prio_lo = 0, prio_hi = 0, prio = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);      // lo = numerically largest = least urgent
    prio = cfg->stream_priority == 1 ? prio_lo : (cfg->stream_priority == 2 ? prio_hi : 0);
    if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&c->ev_t0) != hipSuccess || hipEventCreate(&c->ev_t1) != hipSuccess || hipEventCreate(&c->ev_t2) != hipSuccess) {
        g_create_err = "car_create: stream/event creation failed"; delete c; return -1;
    }
    for (int i = 0; i < 7; ++i)
        if (hipStreamCreateWithPriority(&c->streamx[i], hipStreamNonBlocking, prio) != hipSuccess || hipEventCreateWithFlags(&c->ev_joinx[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_phase[i], hipEventDisableTiming) != hipSuccess) {
            g_create_err = "car_create: stream/event creation failed"; delete c; return -1;
        }
    // 2-D RoPE table (gpt_t2i.py:506-519): rows [0,T) zero, then grid*grid rows of (cos,sin) x 32 pairs
    {
        const int T = cfg->cls_token_num, half = 32, quarter = 16;
        c->rope_rows = T + g * g;
        std::vector<float> tab((size_t)c->rope_rows * 32 * 2, 0.f);
        std::vector<float> freqs(quarter);
        for (int i = 0; i < quarter; ++i) freqs[i] = 1.0f / std::pow((float)cfg->rope_base, (float)(2 * i) / (float)half);
        for (int y = 0; y < g; ++y) for (int x = 0; x < g; ++x) {
            float* row = &tab[((size_t)T + (size_t)y * g + x) * 64];
            for (int i = 0; i < quarter; ++i) {
                const float fy = (float)y * freqs[i], fx = (float)x * freqs[i];
                row[2 * i] = (float)std::cos((double)fy); row[2 * i + 1] = (float)std::sin((double)fy);
                row[2 * (quarter + i)] = (float)std::cos((double)fx); row[2 * (quarter + i) + 1] = (float)std::sin((double)fx);
            }
        }
        if (hipMalloc((void**)&c->rope, tab.size() * 4) != hipSuccess ||
            hipMemcpy(c->rope, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            g_create_err = "car_create: rope table upload failed"; delete c; return -1;
        }
    }
    *out = c;
    return 0;
}

extern "C" void car_destroy(car_ctx* c) {
    if (!c) return;
    (void)hipStreamSynchronize(c->stream);
    if (c->gexec) (void)hipGraphExecDestroy(c->gexec);
    if (c->gexec1) (void)hipGraphExecDestroy(c->gexec1);
    for (auto& kv : c->w) if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto& kv : c->pos_cache) (void)hipFree(kv.second);
    for (auto& kv : c->resize_cache) { (void)hipFree(kv.second.iy); (void)hipFree(kv.second.ix); if (kv.second.wy) (void)hipFree(kv.second.wy); if (kv.second.wx) (void)hipFree(kv.second.wx); }
    if (c->rope) (void)hipFree(c->rope);
    c->ctrl_in.release(); for (auto& b : c->ctrl) b.release(); c->kv.release(); for (auto& b : c->ws) b.release();
    c->scal.release(); c->tok_out.release(); c->maskb.release(); c->maskw.release(); c->dec_parts.release(); c->rowimg.release(); c->rowunc.release(); if (c->host_flags) { (void)hipHostFree(c->host_flags); c->host_flags = nullptr; } c->canny_map.release(); c->t5_in.release(); c->t5_bias.release();
    (void)hipEventDestroy(c->ev_in); (void)hipEventDestroy(c->ev_out); (void)hipEventDestroy(c->ev_t0); (void)hipEventDestroy(c->ev_t1); (void)hipEventDestroy(c->ev_t2);
    (void)hipStreamDestroy(c->stream);
    for (int i = 0; i < 7; ++i) { (void)hipStreamDestroy(c->streamx[i]); (void)hipEventDestroy(c->ev_joinx[i]); (void)hipEventDestroy(c->ev_phase[i]); }
    (void)hipEventDestroy(c->ev_fork);
    delete c;
}

// sample() of generate.py:59-74 as a standalone entry (tests; also usable by callers that bring their own logits):
// logits fp32 [rows, V] (rows = 2B under CFG: cond then uncond), out int32 [B].  `step` only feeds the RNG counter / cfg_interval.
extern "C" int car_sample_logits(car_ctx* c, const float* logits, int32_t B, int32_t V, const car_sampling* sp, int32_t step, int32_t* out, void* stream_) {
    if (!c || !logits || !sp || !out || B <= 0 || V <= 0 || V % 4 || V > 32768) { if (c) c->err = "car_sample_logits: bad arguments (V must be a multiple of 4, <= 32768)"; return -1; }
    if (check_sticky(c)) return -1;
    hipStream_t caller = (hipStream_t)stream_, st = c->stream;
    NEED(c, c->scal, (size_t)(16 + 2 * B) * 4);
    int* stepd = (int*)c->scal.p + 1; int* cur = (int*)c->scal.p + 16;
    fence_in(c, caller);
    SampleP p; memset(&p, 0, sizeof(p));
    p.logits = logits; p.B = B; p.V = V; p.use_cfg = sp->cfg_scale > 1.0f; p.cfg_scale = sp->cfg_scale; p.cfg_interval = sp->cfg_interval;
    p.step_ptr = stepd; p.n_new = 1; p.out_tokens = out; p.cur_tok = cur;
    p.stochastic = sp->sample_logits != 0; p.temperature = sp->temperature; p.top_k = sp->top_k; p.top_p = sp->top_p; p.seed = sp->seed;
    // out_tokens is indexed [i*n_new + step]: n_new = 1 and a zero step pointer keep it dense; the RNG step goes through row0
    HIPCHK(c, hipMemsetAsync(stepd, 0, 4, st));           // no host source buffer, no wait: the entry only enqueues
    p.row0 = step * 65536;
    car_launch_sample_greedy(&p, st);
    fence_out(c, caller);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// Waits for everything enqueued on the context's stream, then reports (and clears) sticky device-side errors — the way to learn NOW whether the
// tokens of the last car_generate_c2i call are valid (include/controlar_hip.h).
extern "C" int car_check_errors(car_ctx* c) {
    if (!c) return -1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_sticky(c);
}

extern "C" int car_get_stats(car_ctx* c, car_stats* out) {
    if (!c || !out) return -1;
    float ms = 0.f;
    (void)hipEventSynchronize(c->ev_t2);
    if (hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1) == hipSuccess) c->stats.prefill_ms = ms; else (void)hipGetLastError();
    if (hipEventElapsedTime(&ms, c->ev_t1, c->ev_t2) == hipSuccess) c->stats.decode_ms = ms; else (void)hipGetLastError();
    {
        // algorithmic bytes of the decode loop: weights once per step for the whole batch + the KV rows a step has to read:
        // positions [first attendable text position, p] of every sequence (text-pad rows are masked and skipped, never fetched)
        const car_config& g = c->cfg;
        std::vector<int> jm((size_t)(c->st_b > 0 ? c->st_b : 1), 0);
        if (c->st_has_mask && c->st_jmin && c->st_b > 0) { (void)hipStreamSynchronize(c->stream); if (hipMemcpy(jm.data(), c->st_jmin, (size_t)c->st_b * 4, hipMemcpyDeviceToHost) != hipSuccess) (void)hipGetLastError(); }
        double kvb = 0;
        for (int s = 0; s < c->st_b; ++s)
            for (int i = 0; i < c->st_nsteps; ++i) { const double p = c->st_T + c->dbg_skip + i; kvb += 2.0 * g.n_layer * g.dim * ((c->mode == CAR_BF16 && g.kv_cache_fp8) ? 1.0 : (double)c->esz) * (p + 1 - jm[(size_t)s]); }
        c->stats.decode_algo_bytes = (int64_t)(c->st_wbytes * c->st_nsteps + kvb);
    }
    {   // the library's A/B / profiling switches are CAR_* environment variables: report how many are set (0 = the shipped schedule)
        // switches the library actually read as set while it enqueued the last generate (latched there): always 0 in the shipped build, which has no switches
        // switches read by car_encode_control / car_vq_decode / finalize, or read once into a static, were missed by a per-generate latch.  The development
        // build therefore reports the larger of (a) the switches READ as set on any entry since the last car_get_stats and (b) the CAR_* names in the environment now.
        int n = c->knob_hits;
#ifdef CAR_DEV_KNOBS
        n = __atomic_exchange_n(&g_car_knob_hits, 0, __ATOMIC_RELAXED);
        if (c->knob_hits > n) n = c->knob_hits;
        { extern char** environ; int e = 0; for (char** ev = environ; ev && *ev; ++ev) if (!strncmp(*ev, "CAR_", 4)) ++e; if (e > n) n = e; }
        c->knob_hits = 0;
#endif
        c->stats.dev_knobs_active = n;
    }
    *out = c->stats;
    if (check_sticky(c)) return -1;        // sticky device-side error flags (this entry has waited for the end event of the last decode loop: everything enqueued up to it has reported)
    return 0;
}

extern "C" int car_debug_control_tokens(car_ctx* c, int32_t k, float* host_out, int64_t max_elems) {
    if (!c || k < 0 || k > 2 || !host_out) return -1;
    (void)hipStreamSynchronize(c->stream);
    const size_t n = c->ctrl[k].cap ? (size_t)max_elems : 0;
    if (!n) FAIL(c, "no control tokens cached");
    if (c->mode == CAR_F32) { HIPCHK(c, hipMemcpy(host_out, c->ctrl[k].p, n * 4, hipMemcpyDeviceToHost)); }
    else { std::vector<bf16_t> hb(n); HIPCHK(c, hipMemcpy(hb.data(), c->ctrl[k].p, n * 2, hipMemcpyDeviceToHost)); for (size_t i = 0; i < n; ++i) host_out[i] = bf2f(hb[i]); }
    return 0;
}
