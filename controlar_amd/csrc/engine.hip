// engine.hip — context, weight packing and orchestration behind the C ABI of include/controlar_hip.h.
//
// Stages:  A resize+patchify -> B DINOv2 -> C control MLPs -> D text embed ->
// E prefill -> F decode loop (hipGraph-captured step replayed n_new-1 times, position/token fed
// back on device) -> G CFG+sampling -> H VQ decode.  Everything below is enqueued on the context's
// own stream (graph capture is illegal on the legacy default stream torch uses by default) and
// fenced against the caller's stream with events — no host synchronisation inside the token loop.
#include "car_common.h"
#include "../../include/controlar_hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

// decode2.hip: parameter blocks shared through one header
#include "decode2_params.h"
#include "decode_f32_params.h"
#include "kernel_params.h"
extern "C" {
int car_launch_dec_gemm_cfg(const GemmDP* p, int epi, int cfg, hipStream_t st);
int car_pick_gemm_cfg(int M, int N, int K, int epi);
void car_launch_dec_attn2_var(const Attn2P* p, int b, int variant, int lds_pad, hipStream_t st);
void car_launch_mask_first_valid(const unsigned char* mask, int* jmin, int b, int T, hipStream_t st);
void car_launch_prefill_rope_kv2(void* qkv, void* kc, void* vc, const float* rope, int b, int Tn, int H, int dim, int SA, int kv8, hipStream_t st);
void car_launch_rmsnorm2(const Norm2P* p, long rows, hipStream_t st);
void car_launch_build_mask(const int64_t* emb_mask, const int* row_img, unsigned char* out, int b, int T, hipStream_t st);
// canny.hip
void car_launch_canny_grad_nms(const unsigned char* img, unsigned char* map, int B, int H, int W, int low, int high, hipStream_t st);
void car_launch_canny_hyst(unsigned char* map, int B, int H, int W, const int* prev, int* changed, hipStream_t st);
void car_launch_canny_finish(int mode, const unsigned char* map, unsigned char* edges, void* control, int B, long HW, hipStream_t st);
// pack.hip
void car_launch_rows_to_bf16(const void* src, int dtype, void* dst, long N, long K, int ileave, hipStream_t st);
void car_launch_pack_frag_bf16(const void* src, void* dst, long N, long K, hipStream_t st);
void car_launch_row_amax_scale(const void* src, int dtype, float* scale, long N, long K, int ileave, hipStream_t st);
void car_launch_quant_pack_fp8(const void* src, int dtype, const float* scale, void* rowmajor, void* pk, long N, long K, int ileave, hipStream_t st);
void car_launch_t5_prep(const long long* ids, const long long* mask, int* ids32, unsigned char* mk, long n, int vocab, hipStream_t st);
void car_launch_t5_softmax(int mode, const float* S, long lds, void* P, long ldp, long rows, int ncols, const float* bias,
                           const unsigned char* mask, int Tq, int n_head, hipStream_t st);
void car_launch_t5_gated_act(int mode, const void* in, void* out, long rows, int hidden, hipStream_t st);
int car_launch_flash64(const FlashP* p, int B, hipStream_t st);
void car_launch_gemm(int mode, int amode, const GemmP* p, hipStream_t st);
int car_conv3_halo64_ok(int mode, const GemmP* p);
void car_launch_convert(int mode, const void* src, int src_dtype, void* dst, long n, hipStream_t st);
void car_launch_build_text(int mode, const void* cond, int src_dtype, const void* uncond, void* dst, int B, long per, int use_cfg, hipStream_t st);
void car_launch_layernorm(int mode, const void* x, const void* w, const void* b, void* y, long rows, int D, float eps, hipStream_t st);
void car_launch_rmsnorm(int mode, const NormP* p, long rows, hipStream_t st);
void car_launch_softmax(int mode, const float* S, long lds, void* P, long ldp, long rows, int ncols, int mask_mode,
                        const unsigned char* emb_mask, int Tq, int n_head, hipStream_t st);
void car_launch_patchify(int mode, const void* img, int img_dtype, void* out, int B, int H, int W, int gh, int gw, int p, int Kpad,
                         int bicubic, const int* iy, const int* ix, const float* wy, const float* wx, hipStream_t st);
void car_launch_vit_assemble(int mode, const void* tok, const void* cls, const void* pos, void* h, int B, int n, int D, hipStream_t st);
void car_launch_groupnorm(int mode, const void* x, const void* gamma, const void* beta, void* y, float* part, float* stats,
                          int B, int HW, int C, int G, float eps, int swish, hipStream_t st);
void car_launch_groupnorm_ex(int mode, const void* x, const void* gamma, const void* beta, void* y, float* part, float* stats, int B, int HW, int C, int G, float eps, int swish, int have_part, hipStream_t st);
void car_launch_vq_lookup(int mode, const int* tok, const float* cb, const float* wpq, const float* bpq, void* z, long npix, int cd, int zc, int ncode, hipStream_t st);
void car_launch_conv_in3(int mode, const float* img, const void* w, const void* b, void* out, int B, int H, int W, int Co, hipStream_t st);
void car_launch_vq_argmin(int mode, const void* z, const float* cb, int* tok, long npix, int cd, int ncode, hipStream_t st);
void car_launch_conv_out(int mode, const void* x, const void* w, const float* bias, float* out, int B, int H, int W, int C, hipStream_t st);
void car_launch_swiglu(int mode, const void* in, void* out, long rows, int hidden, hipStream_t st);
void car_launch_sample_greedy(const SampleP* p, hipStream_t st);
void car_launch_advance(int* pos, int* step, hipStream_t st);
void car_launch_transpose_pad(int mode, const void* src, long ld, long sb, void* dst, int B, int Tn, int Tpad, int C, hipStream_t st);
void car_launch_gather_rows(int mode, const void* table, const int* idx, void* out, long rows, int D, hipStream_t st);
void car_launch_label_index(const int64_t* labels, const int* row_img, const int* row_unc, int num_classes, int* idx, int* err_flag, int b, hipStream_t st);
void car_launch_prefill_rope_kv(int mode, void* qkv, void* kc, void* vc, const float* rope, int b, int Tn, int H, int dim, int S_max, hipStream_t st);
void car_launch_pack_frag_f32(const void* src, void* dst, long N, long K, hipStream_t st);
int car_launch_dec_gemm_f32_cfg(const GemmFP* p, int epi, int cfg, hipStream_t st);
int car_pick_gemm_f32_cfg(int M, int N, int K, int epi);
void car_launch_dec_attn_f32(const AttnFP* p, int b, hipStream_t st);
}

static thread_local std::string g_create_err;

static unsigned long long g_alloc_gen = 0;     // bumped on every (re)allocation: captured graphs bake raw pointers, so their cache key includes it
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        ++g_alloc_gen;
        size_t want = bytes + (bytes >> 3) + 256;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return false; }
        cap = want; return true;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Wt { void* p = nullptr; std::vector<int64_t> shape; int64_t numel = 0; size_t bytes = 0; };

struct car_ctx {
    car_config cfg; int mode = 0; size_t esz = 4;
    std::string err;
    hipStream_t streamx[7] = {}; hipEvent_t ev_fork = nullptr, ev_joinx[7] = {}, ev_phase[7] = {};
    hipStream_t stream = nullptr; hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_t0 = nullptr, ev_t1 = nullptr, ev_t2 = nullptr;
    std::unordered_map<std::string, Wt> w;            // packed weights by (reference) name, element type T unless noted
    std::unordered_map<std::string, std::vector<float>> host_keep;   // host fp32 copies needed later (pos-emb, w1/w3 halves in exact mode)
    std::unordered_map<std::string, int> w13_seen;                    // fast mode: bit 0 = w1 arrived, bit 1 = w3 arrived (per layer prefix)
    bool finalized = false, has_gpt = false;
    // cached tables
    std::map<std::pair<int, int>, void*> pos_cache;   // (gh,gw) -> T [1+gh*gw, D]
    struct ResizeTab { int* iy; int* ix; float* wy; float* wx; };
    std::map<std::pair<int, int>, ResizeTab> resize_cache;
    float* rope = nullptr; int rope_rows = 0;
    // buffers
    DevBuf ctrl_in;      // [B, n_tok, dim] T — adapter_mlp output of the last car_encode_control
    int ctrl_B = 0, ctrl_ntok = 0;
    DevBuf ctrl[3];      // cached control tokens [b, n_tok, dim]
    DevBuf kv;           // [n_layer][2][b, H, S_max, 64]
    DevBuf ws[12];       // scratch
    DevBuf dec_parts;    // split-K partials of the decode linears (fp32)
    DevBuf scal;         // device ints: pos, step, cur_tok[b]
    DevBuf tok_out;      // [B, n_new] int32
    DevBuf maskb;        // [b, T] uint8
    std::vector<int> h_rowimg;          // host staging that must outlive the async copies of a generate call
    int h_init[16] = {};
    SampleDyn h_dyn = {};
    int dbg_skip = 0;
    int n_cu = 256;       // compute units of the device (persistent-grid sizing)
    DevBuf rowimg;       // [b] int: image index of each row
    DevBuf rowunc; std::vector<int> h_rowunc;   // c2i: uncond-row marks (device + the host copy the async upload reads)
    int* host_flags = nullptr;   // sticky error flags raised by device code, in host-mapped pinned memory ([0] = class label out of range): the kernel writes it
                                 // with a system-scope store, and every entry that takes this context reads it without a host wait (check_sticky)
    DevBuf canny_map;    // car_canny: uint8 [B,H,W] candidate/edge map + the "changed" flag
    car_t5_config t5 = {}; bool has_t5 = false;
    DevBuf t5_in;        // int32 ids [B*T] | uint8 key mask [B*T] | staging for host-side int64 inputs
    DevBuf t5_bias; int t5_bias_T = 0;   // position bias fp32 [heads][T][T] of the last sequence length
    int st_b = 0, st_T = 0, st_nsteps = 0, st_has_mask = 0; double st_wbytes = 0; const int* st_jmin = nullptr;   // inputs of the lazy decode_algo_bytes
    // decode graph
    hipGraphExec_t gexec = nullptr; std::string gkey;          // the captured decode step(s): `graph_steps` consecutive tokens per replay
    hipGraphExec_t gexec1 = nullptr; std::string gkey1;        // single-step graph for the remainder when graph_steps > 1
    car_stats stats;
    int n_dec_kernels = 0;
};

#define FAIL(ctx, ...) do { char _b[512]; snprintf(_b, sizeof(_b), __VA_ARGS__); (ctx)->err = _b; return -1; } while (0)
#define HIPCHK(ctx, x) do { hipError_t _e = (x); if (_e != hipSuccess) FAIL(ctx, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); } while (0)
// Errors that only the device can detect (today: a c2i class label outside [0, num_classes] in a device-resident label tensor) cannot fail the call that
// enqueued the work without a host wait.  They are raised as sticky flags in host-mapped memory and fail the NEXT call on the context that runs after the
// offending kernel has executed (car_generate*, car_encode_control, car_vq_*, car_get_stats, car_check_errors) — the reference's nn.Embedding fails
// asynchronously on a GPU as well.  The flag is cleared by the call that reports it.
static int check_sticky(car_ctx* c) {
    if (!c->host_flags) return 0;
    volatile int* f = c->host_flags;
    if (f[0]) { f[0] = 0; FAIL(c, "car_generate_c2i: an earlier call on this context received a class label outside [0, %d] (clamped to the null class on the device): its tokens are invalid", c->cfg.num_classes); }
    return 0;
}
#define NEED(ctx, buf, bytes) do { if (!(buf).ensure(bytes)) FAIL(ctx, "out of device memory allocating %zu bytes (%s:%d)", (size_t)(bytes), __FILE__, __LINE__); } while (0)

static inline size_t rup(size_t x, size_t a) { return (x + a - 1) / a * a; }

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
    c->scal.release(); c->tok_out.release(); c->maskb.release(); c->dec_parts.release(); c->rowimg.release(); c->rowunc.release(); if (c->host_flags) { (void)hipHostFree(c->host_flags); c->host_flags = nullptr; } c->canny_map.release(); c->t5_in.release(); c->t5_bias.release();
    (void)hipEventDestroy(c->ev_in); (void)hipEventDestroy(c->ev_out); (void)hipEventDestroy(c->ev_t0); (void)hipEventDestroy(c->ev_t1); (void)hipEventDestroy(c->ev_t2);
    (void)hipStreamDestroy(c->stream);
    for (int i = 0; i < 7; ++i) { (void)hipStreamDestroy(c->streamx[i]); (void)hipEventDestroy(c->ev_joinx[i]); (void)hipEventDestroy(c->ev_phase[i]); }
    (void)hipEventDestroy(c->ev_fork);
    delete c;
}

// ------------------------------------------------------------------------------------- weights
static bool ends_with(const std::string& s, const char* suf) { size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }
static bool starts_with(const std::string& s, const char* pre) { return s.compare(0, strlen(pre), pre) == 0; }

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
    auto pk = c->w.find(name + "#pk32");       // a re-loaded tensor invalidates its exact-mode fragment image (rebuilt by car_finalize_weights)
    if (pk != c->w.end()) { if (pk->second.p) (void)hipFree(pk->second.p); c->w.erase(pk); }
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

static const void* Wp(car_ctx* c, const std::string& name) {
    auto it = c->w.find(name);
    return it == c->w.end() ? nullptr : it->second.p;
}

struct VqItem { int kind; std::string name; int cin, cout; };   // 0 res, 1 attn, 2 up
static std::vector<VqItem> vq_layout(const car_config& g, int* last_c) {
    // reference: vq_model.py:129-169 (Decoder.__init__), :174-195 (forward order)
    std::vector<VqItem> v;
    const int nres = g.vq_n_mult;
    int block_in = g.vq_ch * g.vq_ch_mult[nres - 1];
    v.push_back({0, "decoder.mid.0", block_in, block_in}); v.push_back({1, "decoder.mid.1", block_in, block_in}); v.push_back({0, "decoder.mid.2", block_in, block_in});
    for (int idx = 0; idx < nres; ++idx) {
        const int i_level = nres - 1 - idx, block_out = g.vq_ch * g.vq_ch_mult[i_level];
        for (int j = 0; j < g.vq_num_res_blocks + 1; ++j) {
            v.push_back({0, "decoder.conv_blocks." + std::to_string(idx) + ".res." + std::to_string(j), block_in, block_out});
            block_in = block_out;
            if (i_level == nres - 1) v.push_back({1, "decoder.conv_blocks." + std::to_string(idx) + ".attn." + std::to_string(j), block_in, block_in});
        }
        if (i_level != 0) v.push_back({2, "decoder.conv_blocks." + std::to_string(idx) + ".upsample", block_in, block_in});
    }
    *last_c = block_in;
    return v;
}

static std::vector<VqItem> vq_enc_layout(const car_config& g, int* last_c) {
    // reference: vq_model.py:62-126 (Encoder).  kind 3 = Downsample
    std::vector<VqItem> v;
    const int nres = g.vq_n_mult;
    int block_in = g.vq_ch;
    for (int i = 0; i < nres; ++i) {
        block_in = g.vq_ch * (i == 0 ? 1 : g.vq_ch_mult[i - 1]);
        const int block_out = g.vq_ch * g.vq_ch_mult[i];
        for (int j = 0; j < g.vq_num_res_blocks; ++j) {
            v.push_back({0, "encoder.conv_blocks." + std::to_string(i) + ".res." + std::to_string(j), block_in, block_out});
            block_in = block_out;
            if (i == nres - 1) v.push_back({1, "encoder.conv_blocks." + std::to_string(i) + ".attn." + std::to_string(j), block_in, block_in});
        }
        if (i != nres - 1) v.push_back({3, "encoder.conv_blocks." + std::to_string(i) + ".downsample", block_in, block_in});
    }
    v.push_back({0, "encoder.mid.0", block_in, block_in}); v.push_back({1, "encoder.mid.1", block_in, block_in}); v.push_back({0, "encoder.mid.2", block_in, block_in});
    *last_c = block_in;
    return v;
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
            car_launch_pack_frag_f32(c->w[r].p, c->w[r + "#pk32"].p, src.shape[0], src.shape[1], 0);
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
    char bid[48]; memset(bid, 0, sizeof(bid)); strncpy(bid, CAR_BUILD_ID, sizeof(bid) - 1);
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
    { char mine[48]; memset(mine, 0, sizeof(mine)); strncpy(mine, CAR_BUILD_ID, sizeof(mine) - 1);
      if (memcmp(bid, mine, sizeof(mine))) FAIL(c, "car_import_packed: %s was written by a different build of the library (packed layouts are per build)", path); }
    if (!same_config(cfg, c->cfg)) FAIL(c, "car_import_packed: %s was written for a different car_config", path);
    if (n > (1u << 20)) FAIL(c, "car_import_packed: %s is corrupt (entry count)", path);
    // two passes: everything is read and validated on the host first, so a corrupt file leaves the context untouched
    struct Ent { uint32_t kind; std::string name; std::vector<int64_t> shape; int64_t numel; std::vector<unsigned char> data; };
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
        if (ok) { e.data.resize((size_t)bytes); if (bytes) ok = fread(e.data.data(), 1, (size_t)bytes, f) == bytes; }
        if (!ok) FAIL(c, "car_import_packed: %s is truncated or corrupt (entry %llu)", path, (unsigned long long)i);
        ents.push_back(std::move(e));
    }
    for (Ent& e : ents) {
        if (e.kind == 1) { std::vector<float> v((size_t)e.numel); if (e.numel) memcpy(v.data(), e.data.data(), e.data.size()); c->host_keep[e.name] = std::move(v); continue; }
        if (ensure_w(c, e.name, e.data.size(), e.shape, e.numel)) return -1;
        if (!e.data.empty() && hipMemcpy(c->w[e.name].p, e.data.data(), e.data.size(), hipMemcpyHostToDevice) != hipSuccess) FAIL(c, "car_import_packed: upload of %s failed", e.name.c_str());
        if (ends_with(e.name, "feed_forward.w13.weight")) c->w13_seen[e.name] = 3;
        e.data = std::vector<unsigned char>();
    }
    c->finalized = false;
    return car_finalize_weights(c);       // names / shapes are checked against the config there: a missing image fails the import
}
extern "C" int car_import_packed(car_ctx* c, const char* path) {
    if (!c || !path) return -1;
    try { return import_packed_impl(c, path); }
    catch (const std::exception& ex) { c->err = std::string("car_import_packed: ") + ex.what(); return -1; }   // no C++ exception crosses the C ABI
}

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

static int get_resize(car_ctx* c, int H, int W, int nh, int nw, car_ctx::ResizeTab* out) {
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
static int get_pos_embed(car_ctx* c, int gh, int gw, void** out) {
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

// ------------------------------------------------------------------------------------- GEMM helpers
static GemmP gp(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K) {
    GemmP p; memset(&p, 0, sizeof(p));
    p.A = A; p.W = W; p.C = C; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    p.alpha = 1.f; p.nb0 = 1; p.nb1 = 1;
    return p;
}
// fused attention (attn.hip) is the fast-mode path for 64-wide heads; CAR_NO_FLASH=1 keeps the unfused GEMM/softmax/GEMM form (A/B runs)
static bool use_flash(const car_ctx* c, int head_dim);
static inline char* off(void* p, size_t elems, size_t esz) { return (char*)p + elems * esz; }
static inline const char* off(const void* p, size_t elems, size_t esz) { return (const char*)p + elems * esz; }

static bool use_flash(const car_ctx* c, int head_dim) {
    static const bool off_env = getenv("CAR_NO_FLASH") != nullptr;
    return c->mode == CAR_BF16 && head_dim == 64 && !off_env;
}

// y = fc2(gelu_tanh(fc1 x))   (gpt_t2i.py:165-181), x: [z][M, K] with row stride lda / batch stride sA
static void mlp_tanh(car_ctx* c, const void* x, long lda, long sA, int nb, int M, int K, const std::string& pfx, void* mid, void* y, int dim, hipStream_t st) {
    GemmP p = gp(x, lda, Wp(c, pfx + "fc1.weight"), K, mid, dim, M, dim, K);
    p.act = ACT_GELU_TANH; p.nb0 = nb; p.sA0 = sA; p.sC0 = (long)M * dim;
    car_launch_gemm(c->mode, AMODE_PLAIN, &p, st);
    GemmP q = gp(mid, dim, Wp(c, pfx + "fc2.weight"), dim, y, dim, M * nb, dim, dim);
    car_launch_gemm(c->mode, AMODE_PLAIN, &q, st);
}

static void fence_in(car_ctx* c, hipStream_t caller) { (void)hipEventRecord(c->ev_in, caller); (void)hipStreamWaitEvent(c->stream, c->ev_in, 0); }
static void fence_out(car_ctx* c, hipStream_t caller) { (void)hipEventRecord(c->ev_out, c->stream); (void)hipStreamWaitEvent(caller, c->ev_out, 0); }

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
    const int chmax = flash ? 64 : 16;  // images per chunk: the unfused form is bounded by its fp32 score matrix (CH*heads*Tn*Tn*4 B)
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
    int nk = 0, bad_cfg = 0;
    // returns the number of sum-of-squares partials per row the kernel leaves in p.ssq_out (0 if it writes none)
    auto gemm = [&](const std::string& wname, const bf16_t* X, int N, int K, int epi, GemmDP gp_) -> int {
        GemmDP p = gp_;
        p.W = (const bf16_t*)Wp(c, wname + (f8 ? "#pk8" : "#pk")); p.X = X; p.M = b; p.N = N; p.K = K;
        p.wscale = f8 ? (const float*)Wp(c, wname + "#sc") : nullptr; p.f8_mfma = g.decode_weight_fp8 == 2;
        const int cfg = car_pick_gemm_cfg(b, N, K, epi);
        const int I = cfg / 100, J = (cfg / 10) % 10, Mb = (b + 15) / 16;
        p.w_nt = ((Mb + J - 1) / J == 1 ? 1 : 0) | (prio ? 2 : 0);      // bit 0: non-temporal weight stream, bit 1: raised wave priority
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
    const bool fuse_norm = b <= 8 && D <= 2048 && !getenv("CAR_NO_SMALL_FUSE");
    // chains of up to 48 rows (round 4, experiments/lat_probe: profiles/r04_lat_probe_v5_*): the RMSNorm in front of wqkv / w1|w3 / output is applied ON THE FLY.
    // The RESID linear that produced the residual stream (wo, w2) leaves each row's sum of squares as per-tile partials; the consumer folds them into rstd
    // and normalises the bf16 residual rows it loads as its X operand in registers (dec_gemm NORM == 2).  Against the prologue form (<= 8 rows: a barrier-
    // separated norm in front of the main loop, 3.6-6.0 us of a 6-8 us kernel) and against the separate rmsnorm2 kernels (> 8 rows: two dependent launches of
    // ~6 us per layer) the measured layer goes 37.0 -> 34.4 us at 2 rows, 39.4 -> 35.7 at 8, 66.5 -> 61.6 at 32; at 64 rows it is a draw (83.3 / 82.9: the
    // 960 workgroups of wqkv each repeat the row statistics) and at 128 a loss (120 / 125), so larger chains keep the norm kernels.  The first norm of layer 0
    // (token gather) and of the three control-add layers changes the stream before it is normed: those keep the prologue / kernel form.
    const bool normx = b <= 48 && fb.ssq != nullptr && !getenv("CAR_NO_NORMX");
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
            car_launch_dec_attn2_var(&ap, b, gr.attn_variant, gr.attn_lds_pad, st); nk += nsplit > 1 ? 2 : 1;
        }
        { GemmDP q = z; q.h = hc; if (normx) q.ssq_out = fb.ssq; ssq_np = gemm(L + "attention.wo.weight", fb.att, D, D, EPI_RESID, q); }
        const bool nx2 = normx && ssq_np > 0;
        if (!nx2 && !fuse_norm) {
            Norm2P np; memset(&np, 0, sizeof(np));
            np.h_in = h; np.xn = fb.xn; np.w = (const bf16_t*)Wp(c, L + "ffn_norm.weight"); np.D = D; np.eps = g.norm_eps; np.add = prio ? 2 : 0;
            car_launch_rmsnorm2(&np, b, st); ++nk;
        }
        { GemmDP q = z; q.outp = fb.mid;
          if (nx2) normx_fields(q, L + "ffn_norm.weight"); else if (fuse_norm) norm_fields(q, L + "ffn_norm.weight", l, false);
          gemm(L + "feed_forward.w13.weight", fb.xn, 2 * Fh, D, EPI_SWIGLU, q); }
        {   // w2 leaves the sums of squares for the next layer's first norm (or the final norm) unless that layer adds a control token first
            const bool next_special = l + 1 < g.n_layer && use_ctrl && (l + 1) % li == 0 && (l + 1) / li < 3;
            GemmDP q = z; q.h = hc; if (normx && !next_special) q.ssq_out = fb.ssq;
            ssq_np = gemm(L + "feed_forward.w2.weight", fb.mid, D, Fh, EPI_RESID, q);
        }
    }
    const bool nx3 = normx && ssq_np > 0;
    if (!nx3 && !fuse_norm) {
        Norm2P np; memset(&np, 0, sizeof(np));
        np.h_in = h; np.xn = fb.xn; np.w = (const bf16_t*)Wp(c, "norm.weight"); np.D = D; np.eps = g.norm_eps; np.add = prio ? 2 : 0;
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

// Exact mode (decode_f32.hip): 8 kernels per layer — norm -> wqkv(+RoPE, q scale, K/V rows written at *pos) -> attention (fixed 128-position
// splits) -> combine -> wo(+residual) -> norm -> w1|w3(+SwiGLU) -> w2(+residual); every linear runs on the exact fp32 MFMA over the fragment-packed
// weights.  Nothing here depends on the batch except the tile shape, which does not change an output's arithmetic: a sequence decodes to the
// same bits alone and in a batch of 384.
static int enqueue_decode_step(car_ctx* c, const StepBufs& sb, int b, int B, int S_max, int n_tok, int nsplit, bool use_ctrl,
                               float cs, const SampleP& sp_tmpl, const unsigned char* maskb, hipStream_t st) {
    const car_config& g = c->cfg; const int mode = c->mode; const size_t e = c->esz;
    const int D = g.dim, Hn = g.n_head, Fh = g.ffn_hidden, li = g.n_layer / 3, V = g.vocab_size;
    const size_t kv_layer = (size_t)b * Hn * S_max * 64;
    int nk = 0, bad = 0;
    auto gemm = [&](const std::string& wname, const void* X, long ldx, int N, int K, int epi, GemmFP q) {
        q.W = (const float*)Wp(c, wname + "#pk32"); q.X = (const float*)X; q.ldx = ldx; q.M = b; q.N = N; q.K = K;
        const int cfg = car_pick_gemm_f32_cfg(b, N, K, epi);
        const int J = cfg % 10, Mb = (b + 15) / 16;
        q.w_nt = (Mb + J - 1) / J == 1;
        if (!q.W || car_launch_dec_gemm_f32_cfg(&q, epi, cfg, st)) bad = cfg ? cfg : -1;
        ++nk;
    };
    GemmFP z; memset(&z, 0, sizeof(z));
    float* qbuf = (float*)sb.qkv;                                       // [b][H][64] rotated, pre-scaled q (the prefill's qkv buffer is idle during decode)
    for (int l = 0; l < g.n_layer; ++l) {
        const std::string L = "layers." + std::to_string(l) + ".";
        float* kc = (float*)off(c->kv.p, (size_t)(2 * l) * kv_layer, e); float* vc = (float*)off(c->kv.p, (size_t)(2 * l + 1) * kv_layer, e);
        {   // token gather (layer 0), control add (layers 0, n/3, 2n/3), attention_norm
            NormP np; memset(&np, 0, sizeof(np));
            np.h_in = sb.h; np.h_out = sb.h; np.xn = sb.xn; np.w = Wp(c, L + "attention_norm.weight"); np.D = D; np.eps = g.norm_eps;
            if (l == 0) { np.emb = Wp(c, "tok_embeddings.weight"); np.idx = sb.cur; }
            if (use_ctrl && l % li == 0 && l / li < 3) { np.add_mode = 1; np.ctrl = c->ctrl[l / li].p; np.pos = sb.pos; np.T = g.cls_token_num; np.n_tok = n_tok; np.cs = cs; }
            car_launch_rmsnorm(mode, &np, b, st); ++nk;
        }
        { GemmFP q = z; q.qout = qbuf; q.kc = kc; q.vc = vc; q.rope = c->rope; q.pos = sb.pos; q.H = Hn; q.S_max = S_max; q.dim = D;
          gemm(L + "attention.wqkv.weight", sb.xn, D, 3 * D, D, FEPI_QKV, q); }
        {
            AttnFP ap; memset(&ap, 0, sizeof(ap));
            ap.q = qbuf; ap.kc = kc; ap.vc = vc; ap.pos = sb.pos; ap.mask = maskb; ap.part = sb.part; ap.out = (float*)sb.att;
            ap.H = Hn; ap.S_max = S_max; ap.T = g.cls_token_num; ap.dim = D; ap.nsplit_max = nsplit;
            car_launch_dec_attn_f32(&ap, b, st); nk += 2;
        }
        { GemmFP q = z; q.out = (float*)sb.h; q.ldo = D; q.R = (const float*)sb.h; gemm(L + "attention.wo.weight", sb.att, D, D, D, FEPI_RESID, q); }
        { NormP np; memset(&np, 0, sizeof(np)); np.h_in = sb.h; np.xn = sb.xn; np.w = Wp(c, L + "ffn_norm.weight"); np.D = D; np.eps = g.norm_eps; car_launch_rmsnorm(mode, &np, b, st); ++nk; }
        { GemmFP q = z; q.out = (float*)sb.mid; q.ldo = Fh; gemm(L + "feed_forward.w13.weight", sb.xn, D, 2 * Fh, D, FEPI_SWIGLU, q); }
        { GemmFP q = z; q.out = (float*)sb.h; q.ldo = D; q.R = (const float*)sb.h; gemm(L + "feed_forward.w2.weight", sb.mid, Fh, D, Fh, FEPI_RESID, q); }
    }
    { NormP np; memset(&np, 0, sizeof(np)); np.h_in = sb.h; np.xn = sb.xn; np.w = Wp(c, "norm.weight"); np.D = D; np.eps = g.norm_eps; car_launch_rmsnorm(mode, &np, b, st); ++nk; }
    { GemmFP q = z; q.out = sb.logits; q.ldo = V; gemm("output.weight", sb.xn, D, V, D, FEPI_PLAIN, q); }      // fp32 logits (exact mode has no bf16 round)
    car_launch_advance(sb.pos, sb.step, st); ++nk;      // pos = T+i+1 consumed next step; step indexes the token being sampled
    SampleP sp = sp_tmpl; car_launch_sample_greedy(&sp, st); ++nk;
    c->n_dec_kernels = nk;
    (void)B;
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
    int NG = (fast && b >= 192) ? 2 : 1;
    if (fast) { const char* ev = getenv("CAR_CHAINS"); if (ev) { int v = atoi(ev); if (v >= 1 && v <= 8 && B / v >= 2) NG = v; } }
    if (getenv("CAR_SINGLE_CHAIN") || NG > B) NG = 1;
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
    NEED(c, c->scal, (size_t)(16 + 2 * b + 2) * 4 + sizeof(SampleDyn) + 16);
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
    int* pos = (int*)c->scal.p; int* step = pos + 1; int* cur = pos + 16; int* jmin = cur + b;
    SampleDyn* dyn = (SampleDyn*)(((uintptr_t)(jmin + b) + 15) & ~(uintptr_t)15);
    c->h_dyn.seed = sp->seed; c->h_dyn.temperature = sp->temperature; c->h_dyn.top_k = sp->top_k; c->h_dyn.top_p = sp->top_p;
    HIPCHK(c, hipMemcpyAsync(dyn, &c->h_dyn, sizeof(SampleDyn), hipMemcpyHostToDevice, st));
    if (emb_mask) car_launch_mask_first_valid((const unsigned char*)c->maskb.p, jmin, b, T, st);
    {
        // profiling aid (tools/pmc_workload.py): start the decode loop `skip` positions late so that a handful of steps under
        // counter collection see a long KV prefix.  The skipped cache rows hold zeros / stale rows: tokens are meaningless.
        int skip = 0; { const char* ev = getenv("CAR_DEBUG_SKIP_STEPS"); if (ev) { skip = atoi(ev); if (skip < 0 || skip > n_new - 2) skip = 0; } }
        c->dbg_skip = skip;
        for (int i = 0; i < 8; ++i) { c->h_init[2 * i] = T + skip; c->h_init[2 * i + 1] = skip; }    // (pos, step) per chain: after prefill the first decode step runs at input_pos = T, sampling token index 1
        HIPCHK(c, hipMemcpyAsync(pos, c->h_init, 64, hipMemcpyHostToDevice, st));
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
        const long per = (long)T * g.caption_dim; const size_t ib = text_dtype == CAR_DT_BF16 ? 2 : 4;
        for (int gi = 0; gi < NG; ++gi)
            car_launch_build_text(mode, (const char*)text_emb + (size_t)img0[gi] * per * ib, text_dtype, Wp(c, "cls_embedding.uncond_embedding"),
                                  off(text, (size_t)mult * img0[gi] * per, e), img0[gi + 1] - img0[gi], per, use_cfg, st);
        mlp_tanh(c, text, g.caption_dim, 0, 1, (int)rowsP, g.caption_dim, "cls_embedding.cap_proj.", xn, h, D, st);
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
    // ---- E. prefill over the T prefix rows (gpt_t2i.py:446-470)
    for (int l = 0; l < g.n_layer; ++l) {
        const std::string L = "layers." + std::to_string(l) + ".";
        {
            NormP np; memset(&np, 0, sizeof(np));
            np.h_in = h; np.h_out = h; np.xn = xn; np.w = Wp(c, L + "attention_norm.weight"); np.D = D; np.eps = g.norm_eps;
            if (use_control && l % li == 0 && l / li < 3) { np.add_mode = 2; np.ctrl = c->ctrl[l / li].p; np.T = T; np.n_tok = n_tok; np.cs = cs; }
            car_launch_rmsnorm(mode, &np, rowsP, st);
        }
        { GemmP q = gp(xn, D, Wp(c, L + "attention.wqkv.weight"), D, qkv, 3 * D, (int)rowsP, 3 * D, D); car_launch_gemm(mode, AMODE_PLAIN, &q, st); }
        if (fast) car_launch_prefill_rope_kv2(qkv, off(c->kv.p, (size_t)(2 * l) * kv_layer, kv_e), off(c->kv.p, (size_t)(2 * l + 1) * kv_layer, kv_e), c->rope, b, T, Hn, D, SA, g.kv_cache_fp8 ? 1 : 0, st);
        else car_launch_prefill_rope_kv(mode, qkv, off(c->kv.p, (size_t)(2 * l) * kv_layer, e), off(c->kv.p, (size_t)(2 * l + 1) * kv_layer, e), c->rope, b, T, Hn, D, S_max, st);
        car_launch_transpose_pad(mode, off(qkv, (size_t)2 * D, e), 3 * D, (long)T * 3 * D, vT, b, T, Tpad, D, st);
        bool fused = false;
        if (pf_flash) {
            FlashP f; memset(&f, 0, sizeof(f));
            f.q = (const bf16_t*)qkv; f.k = (const bf16_t*)qkv + D; f.vt = (const bf16_t*)vT; f.o = (bf16_t*)att;
            f.q_sb = f.k_sb = (long)T * 3 * D; f.q_st = f.k_st = 3 * D; f.vt_sb = (long)D * Tpad; f.vt_ld = Tpad; f.o_sb = (long)T * D; f.o_st = D;
            f.Tq = f.Tk = T; f.H = Hn; f.scale = 0.125f; f.mode = 1; f.mask = (const unsigned char*)c->maskb.p;
            fused = car_launch_flash64(&f, b, st) == 0;
        }
        if (!fused) {
            GemmP q = gp(qkv, 3 * D, off(qkv, (size_t)D, e), 3 * D, S, T, T, T, 64);
            q.alpha = 0.125f; q.out_f32 = 1; q.nb0 = b; q.nb1 = Hn;
            q.sA0 = (long)T * 3 * D; q.sA1 = 64; q.sW0 = (long)T * 3 * D; q.sW1 = 64; q.sC0 = (long)Hn * T * T; q.sC1 = (long)T * T;
            car_launch_gemm(mode, AMODE_PLAIN, &q, st);
            car_launch_softmax(mode, S, T, P, Tpad, (long)b * Hn * T, T, 1, (const unsigned char*)c->maskb.p, T, Hn, st);
        }
        if (!fused) {
            GemmP q = gp(P, Tpad, vT, Tpad, att, D, T, 64, Tpad);
            q.nb0 = b; q.nb1 = Hn;
            q.sA0 = (long)Hn * T * Tpad; q.sA1 = (long)T * Tpad; q.sW0 = (long)D * Tpad; q.sW1 = (long)64 * Tpad; q.sC0 = (long)T * D; q.sC1 = 64;
            car_launch_gemm(mode, AMODE_PLAIN, &q, st);
        }
        { GemmP q = gp(att, D, Wp(c, L + "attention.wo.weight"), D, h, D, (int)rowsP, D, D); q.R = h; q.ldr = D; car_launch_gemm(mode, AMODE_PLAIN, &q, st); }
        { NormP np; memset(&np, 0, sizeof(np)); np.h_in = h; np.xn = xn; np.w = Wp(c, L + "ffn_norm.weight"); np.D = D; np.eps = g.norm_eps; car_launch_rmsnorm(mode, &np, rowsP, st); }
        if (mode == CAR_BF16) {
            GemmP q = gp(xn, D, Wp(c, L + "feed_forward.w13.weight"), D, mid, Fh, (int)rowsP, 2 * Fh, D); q.swiglu = 1; car_launch_gemm(mode, AMODE_PLAIN, &q, st);
        } else {
            void* mid2 = off(mid, (size_t)rowsP * Fh, e);
            GemmP q = gp(xn, D, Wp(c, L + "feed_forward.w13.weight"), D, mid2, 2 * Fh, (int)rowsP, 2 * Fh, D); car_launch_gemm(mode, AMODE_PLAIN, &q, st);
            car_launch_swiglu(mode, mid2, mid, rowsP, Fh, st);
        }
        { GemmP q = gp(mid, Fh, Wp(c, L + "feed_forward.w2.weight"), Fh, h, D, (int)rowsP, D, Fh); q.R = h; q.ldr = D; car_launch_gemm(mode, AMODE_PLAIN, &q, st); }
    }
    // final norm + logits for the LAST prefix row only (generate.py:60 samples logits[:, -1])
    { NormP np; memset(&np, 0, sizeof(np)); np.h_in = h; np.xn = xn; np.w = Wp(c, "norm.weight"); np.D = D; np.eps = g.norm_eps; car_launch_rmsnorm(mode, &np, rowsP, st); }
    { GemmP q = gp(off(xn, (size_t)(T - 1) * D, e), (long)T * D, Wp(c, "output.weight"), D, logits, V, b, V, D); q.out_f32 = 1; car_launch_gemm(mode, AMODE_PLAIN, &q, st); }
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
            const bool one_launch = (long)bg * Hn <= 240 && !getenv("CAR_ATTN_SPLIT_SMALL");
            if (one_launch) gr.nsplit = 1;
            { const char* ev = getenv("CAR_ATTN_NSPLIT"); if (ev) { const int v = atoi(ev); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) gr.nsplit = v; } }   // A/B knob: shorter attention workgroups
            // attention variant (decode2.hip; profiles/r02_kbench_*): 4 waves per (sequence, head) from 128 sequences up, 2 below; 16 in the one-launch small form
            gr.attn_variant = (one_launch && gr.nsplit == 1) ? 160 : ((gr.nsplit == 1 && bg < 128) ? 20 : 40); gr.attn_lds_pad = 0;
            { const char* ev = getenv("CAR_ATTN_VARIANT"); if (ev) gr.attn_variant = atoi(ev); ev = getenv("CAR_ATTN_LDS_PAD"); if (ev) gr.attn_lds_pad = atoi(ev); }
            // persistent attention grid: R resident workgroups per CU walk the (sequence, head) items in equal shares
            gr.attn_pgrid = 0;
            { const char* ev = getenv("CAR_ATTN_PERSIST"); const int R = ev ? atoi(ev) : 0;
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
    const unsigned char* fmask = emb_mask ? (const unsigned char*)c->maskb.p : nullptr;      // no text-pad mask: nothing to test per position
    const int* fjmin = emb_mask ? jmin : nullptr;
    // ---- decode-loop schedule knobs (fast mode), all OFF by default: the MI355X sweeps of tools/overlap_sweep.py found none of them worth a
    // per cent (profiles/r02_overlap_sweep_v1..v3, DESIGN.md §4 — a linear beside the bandwidth-saturating attention makes no progress whatever
    // the schedule); they stay as A/B switches, and tests/test_parity_gpu.py pins that none of them changes a token.
    //   phase offset : with >= 2 chains, chain g+1 enters the step right after chain g's first wqkv (see enqueue_decode_step_fast)
    //   graph steps  : consecutive tokens captured per graph replay — the chains free-run across them (one fork / join and one phase
    //                  offset per `gsteps` tokens instead of per token); the remainder runs on a single-step graph
    //   linear prio  : s_setprio on the linears / norms
    int phase = 0, gsteps = 1, lin_prio = 0;
    if (fast) {
        const char* ev = getenv("CAR_PHASE_OFFSET"); if (ev) phase = atoi(ev) != 0;
        ev = getenv("CAR_GRAPH_STEPS"); if (ev) { const int v = atoi(ev); if (v >= 1 && v <= 64) gsteps = v; }
        ev = getenv("CAR_LINEAR_PRIO"); if (ev) lin_prio = atoi(ev) != 0;
    }
    if (NG < 2) phase = 0;
    bool capturing = false;
    int step_rc = 0;
    auto enqueue_steps = [&](int k) {      // k consecutive decode steps of every chain
        if (!fast) { for (int s = 0; s < k; ++s) step_rc |= enqueue_decode_step(c, sb, b, B, S_max, n_tok, nsplit, use_control != 0, cs, spp, fmask, st); return; }
        if (NG >= 2 && capturing) {         // the chains are parallel branches of the captured graph
            if (!phase) {
                (void)hipEventRecord(c->ev_fork, st);
                for (int gi = 1; gi < NG; ++gi) (void)hipStreamWaitEvent(c->streamx[gi - 1], c->ev_fork, 0);
            }
            for (int gi = 0; gi < NG; ++gi) {
                hipStream_t sg = gi == 0 ? st : c->streamx[gi - 1];
                for (int s = 0; s < k; ++s) {
                    const bool hand = phase && s == 0 && gi + 1 < NG;      // chain gi+1's stream joins the capture through this event
                    step_rc |= enqueue_decode_step_fast(c, sb, grp[gi], b, SA, n_tok, use_control != 0, cs, fmask, fjmin, sg,
                                                        hand ? c->ev_phase[gi] : nullptr, hand ? c->streamx[gi] : nullptr, lin_prio);
                }
                if (gi > 0) { (void)hipEventRecord(c->ev_joinx[gi - 1], sg); (void)hipStreamWaitEvent(st, c->ev_joinx[gi - 1], 0); }
            }
        } else {
            for (int s = 0; s < k; ++s)
                for (int gi = 0; gi < NG; ++gi) step_rc |= enqueue_decode_step_fast(c, sb, grp[gi], b, SA, n_tok, use_control != 0, cs, fmask, fjmin, st, nullptr, nullptr, lin_prio);
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
                                  grp[0].nsplit, phase, lin_prio, grp[0].attn_pgrid + 100000 * ((getenv("CAR_NO_NORMX") ? 1 : 0) + (getenv("CAR_NO_SMALL_FUSE") ? 2 : 0)));
          strncat(keyb, kb2, sizeof(keyb) - strlen(keyb) - 1); }
        const std::string key(keyb);
        const bool no_graph = getenv("CAR_NO_GRAPH") != nullptr;      // profiling aid: eager launches (PMC collection cannot follow graph replays)
        // capture `k` steps into `ex` unless the cached exec already holds exactly this configuration
        auto get_exec = [&](hipGraphExec_t& ex, std::string& exkey, int k) -> bool {
            const std::string kk = key + "|k" + std::to_string(k);
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
        const int nrep = nsteps / gsteps, nrem = nsteps % gsteps;
        bool graph_ok = !no_graph;
        if (graph_ok && nrep > 0) graph_ok = get_exec(c->gexec, c->gkey, gsteps);
        if (graph_ok && nrem > 0) graph_ok = get_exec(gsteps > 1 ? c->gexec1 : c->gexec, gsteps > 1 ? c->gkey1 : c->gkey, 1);
        if (graph_ok && !step_rc) {
            for (int i = 0; i < nrep; ++i) HIPCHK(c, hipGraphLaunch(c->gexec, st));
            for (int i = 0; i < nrem; ++i) HIPCHK(c, hipGraphLaunch(gsteps > 1 ? c->gexec1 : c->gexec, st));
            c->stats.graph_used = 1;
        } else if (!step_rc) {
            for (int i = 0; i < nsteps; ++i) enqueue_steps(1);
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

// sample() of generate.py:59-74 as a standalone entry (tests; also usable by callers that bring their own logits):
// logits fp32 [rows, V] (rows = 2B under CFG: cond then uncond), out int32 [B].  `step` only feeds the RNG counter / cfg_interval.
extern "C" int car_sample_logits(car_ctx* c, const float* logits, int32_t B, int32_t V, const car_sampling* sp, int32_t step, int32_t* out, void* stream_) {
    if (!c || !logits || !sp || !out || B <= 0 || V <= 0 || V % 4 || V > 32768) { if (c) c->err = "car_sample_logits: bad arguments (V must be a multiple of 4, <= 32768)"; return -1; }
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
    *out = c->stats;
    if (check_sticky(c)) return -1;        // sticky device-side error flags (this entry has just synchronised: everything enqueued so far has reported)
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


// ------------------------------------------------------------------------------------- Canny control extraction
// cv2.Canny(img, low, high) of condition/canny.py:6-14 for a batch of 8-bit RGB photos [B,H,W,3] (device).  edges_out: uint8 [B,H,W] in
// {0,255} or NULL; control_out: [B,3,H,W] in the context's element type = 2*(edges/255 - 0.5) replicated over 3 channels
// (sample_t2i.py:125,141) or NULL — ready for car_encode_control.  The hysteresis fixed point is checked on the host between launches
// (this runs in front of the path, not inside the token loop).
extern "C" int car_canny(car_ctx* c, const uint8_t* img_hwc, int32_t B, int32_t H, int32_t W, float low_threshold, float high_threshold,
                         uint8_t* edges_out, void* control_out, void* stream_) {
    if (!c) return -1;
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
        car_launch_gemm(mode, amode, &q, st);
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
