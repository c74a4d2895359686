// engine_internal.h — what the translation units of the host side share (engine.hip: lifecycle / stats / standalone sampler; engine_weights.hip: weight
// packing, finalize, the packed-image cache; engine_encode.hip: resize tables, the control encoder, Canny; engine_generate.hip: prefill, the decode step,
// generate; engine_t5.hip: the caption encoder; engine_vq.hip: VQ encode / decode).  Round 4 cut the single 1900-line engine.hip along its stage banners:
// no behaviour change.  Everything here is internal: the C ABI is include/controlar_hip.h.
#pragma once
#include "car_common.h"
#include "../../include/controlar_hip.h"

#include <algorithm>
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
void car_launch_prefill_rope_kv2(void* qkv, void* kc, void* vc, const float* rope, int b, int Tn, int H, int dim, int SA, int kv8, int t0, hipStream_t st);
void car_launch_min_int(const int* v, int n, int* out, int need, int* flag, hipStream_t st);
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
int car_launch_gemm(int mode, int amode, const GemmP* p, hipStream_t st);   // -1: GemmP::gn_part on a call that cannot take conv3_halo64_kernel (nothing launched)
int car_conv3_halo64_ok(int mode, const GemmP* p);
void car_launch_convert(int mode, const void* src, int src_dtype, void* dst, long n, hipStream_t st);
void car_launch_build_text(int mode, const void* cond, int src_dtype, const void* uncond, void* dst, int B, long per, long per_src, long src_off, int use_cfg, hipStream_t st);
void car_launch_layernorm(int mode, const void* x, const void* w, const void* b, void* y, long rows, int D, float eps, hipStream_t st);
void car_launch_rmsnorm(int mode, const NormP* p, long rows, hipStream_t st);
void car_launch_softmax(int mode, const float* S, long lds, void* P, long ldp, long rows, int ncols, int mask_mode,
                        const unsigned char* emb_mask, int Tq, int n_head, hipStream_t st);
void car_launch_softmax_at(int mode, const float* S, long lds, void* P, long ldp, long rows, int ncols, int mask_mode,
                           const unsigned char* emb_mask, int Tq, int n_head, int col0, hipStream_t st);
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
void car_launch_set_pos_step(int* dst, int pos, int step, hipStream_t st);
void car_launch_transpose_pad(int mode, const void* src, long ld, long sb, void* dst, int B, int Tn, int Tpad, int C, hipStream_t st);
void car_launch_gather_rows(int mode, const void* table, const int* idx, void* out, long rows, int D, hipStream_t st);
void car_launch_label_index(const int64_t* labels, const int* row_img, const int* row_unc, int num_classes, int* idx, int* err_flag, int b, hipStream_t st);
void car_launch_prefill_rope_kv(int mode, void* qkv, void* kc, void* vc, const float* rope, int b, int Tn, int H, int dim, int S_max, int t0, hipStream_t st);
void car_launch_pack_frag_f32(const void* src, void* dst, long N, long K, const void* colscale, hipStream_t st);
int car_launch_dec_gemm_f32_cfg(const GemmFP* p, int epi, int cfg, hipStream_t st);
int car_pick_gemm_f32_cfg(int M, int N, int K, int epi);
int car_pick_gemm_f32_cfg2(int M, int N, int K, int epi, int chains);
void car_launch_dec_attn_f32(const AttnFP* p, int b, hipStream_t st);
void car_launch_dec_attn_f32_ex(const AttnFP* p, int b, int fused, hipStream_t st);
}


extern unsigned long long g_alloc_gen;     // bumped on every (re)allocation: captured graphs bake raw pointers, so their cache key includes it (defined in engine.hip)
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        ++g_alloc_gen;
        size_t want = bytes + (bytes >> 3) + 256;
        // (an uncached allocation for the KV cache — hipDeviceMallocUncached, the stream is read once per step — changed nothing: 17.60 vs 17.58 ms per exact step)
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
    DevBuf maskw;        // [b, Tv] uint8: the mask columns of the prefill window
    std::vector<int> h_rowimg;          // host staging that must outlive the async copies of a generate call
    int h_init[16] = {};
    int h_init2[16] = {};     // the chains' (pos, step) scalars at the switch from the early one-chain schedule to the main one (exact mode)
    SampleDyn h_dyn = {};
    int dbg_skip = 0;
    int knob_hits = 0;      // development build: CAR_* switches found set while the last generate was enqueued (car_stats.dev_knobs_active)
    int n_cu = 256;       // compute units of the device (persistent-grid sizing)
    DevBuf rowimg;       // [b] int: image index of each row
    DevBuf rowunc; std::vector<int> h_rowunc;   // c2i: uncond-row marks (device + the host copy the async upload reads)
    int* host_flags = nullptr;   // sticky error flags raised by device code, in host-mapped pinned memory ([0] = class label out of range, [1] = first_valid_hint too large): the kernel writes it
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
static inline int check_sticky(car_ctx* c) {
    if (!c->host_flags) return 0;
    volatile int* f = c->host_flags;
    if (f[1]) { f[1] = 0; FAIL(c, "car_generate: an earlier call passed a first_valid_hint beyond the first valid prompt position of its batch: valid prompt rows were left out of the prefill, its tokens are invalid"); }
    if (f[0]) { f[0] = 0; FAIL(c, "car_generate_c2i: an earlier call on this context received a class label outside [0, %d] (clamped to the null class on the device): its tokens are invalid", c->cfg.num_classes); }
    return 0;
}
#define NEED(ctx, buf, bytes) do { if (!(buf).ensure(bytes)) FAIL(ctx, "out of device memory allocating %zu bytes (%s:%d)", (size_t)(bytes), __FILE__, __LINE__); } while (0)

static inline size_t rup(size_t x, size_t a) { return (x + a - 1) / a * a; }


// ------------------------------------------------------------------------------------- shared helpers
static inline bool ends_with(const std::string& s, const char* suf) { size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }
static inline bool starts_with(const std::string& s, const char* pre) { return s.compare(0, strlen(pre), pre) == 0; }

static inline const void* Wp(car_ctx* c, const std::string& name) {
    auto it = c->w.find(name);
    return it == c->w.end() ? nullptr : it->second.p;
}

struct VqItem { int kind; std::string name; int cin, cout; };   // 0 res, 1 attn, 2 up
static inline std::vector<VqItem> vq_layout(const car_config& g, int* last_c) {
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

static inline std::vector<VqItem> vq_enc_layout(const car_config& g, int* last_c) {
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

// ------------------------------------------------------------------------------------- GEMM helpers
static inline GemmP gp(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K) {
    GemmP p; memset(&p, 0, sizeof(p));
    p.A = A; p.W = W; p.C = C; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    p.alpha = 1.f; p.nb0 = 1; p.nb1 = 1;
    return p;
}
// fused attention (attn.hip) is the fast-mode path for 64-wide heads; CAR_NO_FLASH=1 keeps the unfused GEMM/softmax/GEMM form (A/B runs)
static inline char* off(void* p, size_t elems, size_t esz) { return (char*)p + elems * esz; }
static inline const char* off(const void* p, size_t elems, size_t esz) { return (const char*)p + elems * esz; }

static inline bool use_flash(const car_ctx* c, int head_dim) {
    static const bool off_env = CAR_KNOB("CAR_NO_FLASH") != nullptr;
    return c->mode == CAR_BF16 && head_dim == 64 && !off_env;
}

// y = fc2(gelu_tanh(fc1 x))   (gpt_t2i.py:165-181), x: [z][M, K] with row stride lda / batch stride sA
static inline void mlp_tanh(car_ctx* c, const void* x, long lda, long sA, int nb, int M, int K, const std::string& pfx, void* mid, void* y, int dim, hipStream_t st) {
    GemmP p = gp(x, lda, Wp(c, pfx + "fc1.weight"), K, mid, dim, M, dim, K);
    p.act = ACT_GELU_TANH; p.nb0 = nb; p.sA0 = sA; p.sC0 = (long)M * dim;
    car_launch_gemm(c->mode, AMODE_PLAIN, &p, st);
    GemmP q = gp(mid, dim, Wp(c, pfx + "fc2.weight"), dim, y, dim, M * nb, dim, dim);
    car_launch_gemm(c->mode, AMODE_PLAIN, &q, st);
}

static inline void fence_in(car_ctx* c, hipStream_t caller) { (void)hipEventRecord(c->ev_in, caller); (void)hipStreamWaitEvent(c->stream, c->ev_in, 0); }
static inline void fence_out(car_ctx* c, hipStream_t caller) { (void)hipEventRecord(c->ev_out, c->stream); (void)hipStreamWaitEvent(caller, c->ev_out, 0); }

// engine_encode.hip
int get_resize(car_ctx* c, int H, int W, int nh, int nw, car_ctx::ResizeTab* out);
int get_pos_embed(car_ctx* c, int gh, int gw, void** out);

