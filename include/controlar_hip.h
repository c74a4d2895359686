/*
 * controlar_hip.h — C ABI of libcontrolar_hip.so: the MI355X (gfx950) implementation of
 * ControlAR's conditional-decoding hot path
 *     DINOv2 control encoder -> LlamaGen AR decode with per-token control fusion -> VQGAN decoder.
 *
 * The reference (hustvl/ControlAR) has no FFI/plugin layer: its seams are Python callables
 * (SURVEY.md §8b).  Each entry point below replaces one of those callables; the ctypes binding
 * that a reference maintainer would add is shown in INTEGRATION.md and shipped in
 * controlar_amd/_lib.py.  No torch types cross this boundary: plain pointers, sizes, ints.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; car_last_error(ctx) gives the text
 *     (reference raises Python exceptions / asserts: generate.py:173,185-186; gpt_t2i.py:229).
 *   - all data pointers are DEVICE pointers unless stated; the caller (PyTorch) owns inputs and
 *     outputs; the library owns packed weights, KV caches, control-token buffers, hipGraphs and
 *     workspaces (SURVEY.md §8b "Ownership").
 *   - work is enqueued on the hipStream_t passed as `stream` (void* so the header needs no HIP
 *     include); functions that return values to the HOST say so and synchronise that stream.
 *   - a context is re-entrant per ctx but not thread-safe within one ctx.
 */
#ifndef CONTROLAR_HIP_H
#define CONTROLAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAR_ABI_VERSION 2      /* 2: car_stats.dev_knobs_active, car_check_errors, car_sampling.first_valid_hint (round 4-5) */

/* arithmetic mode of a context */
enum { CAR_F32 = 0,   /* "exact" mode: fp32 weights/activations/FMA — parity contract vs the fp32 CPU reference */
       CAR_BF16 = 1   /* fast mode: bf16 storage, fp32 accumulate, the reference's bf16 rounding points */ };

/* element types accepted by car_load_tensor / image inputs */
enum { CAR_DT_F32 = 0, CAR_DT_BF16 = 1, CAR_DT_I32 = 2, CAR_DT_I64 = 3, CAR_DT_U8 = 4 };

/* resize in front of the encoder — reference: autoregressive/models/dinov2_adapter.py:16-24 */
enum { CAR_RESIZE_NEAREST = 0,          /* condition_type in {'canny','seg'} */
       CAR_RESIZE_BICUBIC_AC = 1 };     /* everything else: bicubic, align_corners=True */

typedef struct car_config {
    int32_t abi_version;      /* = CAR_ABI_VERSION */
    int32_t mode;             /* CAR_F32 | CAR_BF16 */
    /* LlamaGen transformer — reference: autoregressive/models/gpt_t2i.py:31-60 ModelArgs */
    int32_t dim, n_layer, n_head, ffn_hidden, vocab_size;
    int32_t cls_token_num;    /* text prefix length T (120) */
    int32_t block_size;       /* rope grid = sqrt(block_size)  (gpt_t2i.py:351-353) */
    int32_t caption_dim;      /* 2048 */
    float   norm_eps;         /* 1e-5 */
    float   rope_base;        /* 10000 */
    /* control encoder — HF Dinov2Config as built by dinov2_adapter.py:13 */
    int32_t vit_hidden, vit_layers, vit_heads, vit_mlp, vit_patch, vit_pos_grid;
    float   vit_ln_eps;       /* 1e-6 */
    int32_t resize_mode;      /* CAR_RESIZE_* */
    /* VQGAN decoder — reference: tokenizer/tokenizer_image/vq_model.py:12-24,129-169 */
    int32_t codebook_size, codebook_dim, z_channels, vq_ch, vq_num_res_blocks;
    int32_t vq_n_mult;        /* number of entries used in vq_ch_mult (5 for VQ-16) */
    int32_t vq_ch_mult[8];
    float   gn_eps;           /* 1e-6 */
    int32_t vit_variant;      /* 0 = HF Dinov2Model (t2i; LayerScale, patch-14 resize)   1 = HF ViTModel (c2i: gpt.py:319, vit_adapter.py:11-15) */
    int32_t model_type;       /* 0 = t2i (gpt_t2i.py)   1 = c2i (gpt.py: class-label prefix of length 1) */
    int32_t num_classes;      /* c2i: LabelEmbedder rows = num_classes + 1, CFG null class = num_classes (gpt.py:66-96) */
    int32_t stream_priority;  /* priority of the context's internal HIP streams: 0 default, 1 lowest, 2 highest (overlapping a
                                 compute-bound context with a latency-bound one on the same GPU) */
    int32_t decode_weight_fp8; /* CAR_BF16 only: the five decode linears (wqkv, wo, w1|w3, w2, output) stream OCP e4m3fn weights with
                                  per-output-row fp32 scales (BASELINE config 5).  1 = weight-only: bytes widened to bf16 in registers, bf16 MFMA.
                                  2 = W8A8: activations quantised to e4m3 (unit scale, clamped to +-448) in registers and multiplied on the
                                  fp8 MFMA (v_mfma_f32_16x16x32_fp8_fp8).  The reference has no fp8 path: tolerance-graded only. */
    int32_t kv_cache_fp8;     /* CAR_BF16 only, opt-in (0 = bf16 KV cache: the default and the parity path).  1 = the KV cache holds OCP e4m3fn bytes, unit scale:
                                  rotated K and V are rounded to e4m3 when they are stored, widened to bf16 in registers in front of the same bf16 MFMAs; the
                                  prefill's own attention reads the unrounded rows.  Halves the dominant HBM stream of large batches and doubles the sequences
                                  that fit.  The reference has no such mode: tolerance-graded against the oracle running the same model (kv_fp8=True). */
    int32_t reserved[2];
} car_config;

/* sampling parameters — reference: generate.py:59-74 sample(), :134 generate() kwargs */
typedef struct car_sampling {
    float    cfg_scale;        /* > 1 doubles the batch (cond | uncond)           generate.py:156-164 */
    int32_t  cfg_interval;     /* -1 = always mix                                  generate.py:121-122 */
    float    temperature;      /* logits / max(T,1e-5)                             generate.py:60 */
    int32_t  top_k;            /* 0 = off                                          generate.py:33-38 */
    float    top_p;            /* 1 = off                                          generate.py:40-55 */
    int32_t  sample_logits;    /* 0 = greedy topk(probs,1): ties -> lowest index   generate.py:71-73 */
    uint64_t seed;             /* counter-based RNG seed when sample_logits != 0 */
    float    control_strength; /* ignored (=1) when cfg_scale <= 1                 generate.py:87-92 */
    int32_t  first_valid_hint; /* 0 = unknown.  Otherwise 1 + a LOWER bound of the first attendable position over all prompts of the call (left-padded captions,
                                  sample_t2i.py:146-160): T - longest valid length, known to a caller that built the mask on the host.  With it car_generate sizes the
                                  prefill window without reading the device mask back — no host wait, the call only enqueues (every per-call scalar is a kernel argument;
                                  capturing the CALLER's stream around the call is not a tested configuration: the work runs on the library's own stream).  Too
                                  small a value only costs time; a value beyond the true minimum would drop valid prompt rows: it is checked on the device and
                                  raises the sticky error flag (car_check_errors). */
    int32_t  reserved[3];
} car_sampling;

typedef struct car_ctx car_ctx;

/* lifecycle */
int  car_create(car_ctx** out, const car_config* cfg);
void car_destroy(car_ctx* ctx);
const char* car_last_error(const car_ctx* ctx);   /* also valid with ctx == NULL after a failed car_create */
int  car_abi_version(void);
const char* car_build_id(void);                   /* hash of the library sources: packed-weight cache files are valid for ONE build */

/*
 * Weight loading — keyed by the reference state_dict names (SURVEY.md §8b "Weight contract"):
 * gpt_t2i.Transformer / gpt.Transformer names (incl. HF Dinov2 / ViT under "adapter.model.", 4.x and 5.x key spellings)
 * and VQModel names (decoder.*, post_quant_conv.*, quantize.embedding.weight; encoder.* + quant_conv.* enable car_vq_encode).
 * `ptr` may be a host or device pointer; dtype CAR_DT_F32 or CAR_DT_BF16; shape row-major.
 * Reference tensors that inference never reads (condition_embeddings.weight, *.mask_token, pooler.*, condition_norm.weight,
 * quantize.codebook_used) are accepted and ignored.  Tensors are packed as they arrive (MFMA-fragment images of the decode
 * linears, w1|w3 interleave, implicit-GEMM conv layouts, optional e4m3 quantisation);
 * car_finalize_weights checks that every tensor the loaded model halves need has arrived.
 */
int car_load_tensor(car_ctx* ctx, const char* name, const void* ptr, const int64_t* shape, int32_t ndim, int32_t dtype);
int car_finalize_weights(car_ctx* ctx);

/*
 * Packed-weight cache — the start-up cost of sample_t2i.py:64-83 / demo/model.py:66-75 (checkpoints re-read and re-loaded on every
 * start, in the demo on every request).  car_export_packed writes every weight image of a FINALISED context (row-major
 * operands, MFMA-fragment / e4m3 decode images, scales, conv layouts, host tables) to `path`; car_import_packed restores them
 * by plain copies into a context created with the same car_config and finalises it.  The file is only valid for this
 * library build (magic + car_config are checked); keying it on the checkpoint content is the caller's job
 * (controlar_amd/checkpoint.py).
 */
int car_export_packed(car_ctx* ctx, const char* path);
int car_import_packed(car_ctx* ctx, const char* path);

/*
 * Control encoder + adapter MLP — replaces model.adapter(condition) + model.adapter_mlp(...)
 * (generate.py:136-138; dinov2_adapter.py:26-29).  img: [B,3,H,W] in [-1,1], dtype F32/BF16.
 * out (optional, may be NULL): [B,(H/16)(W/16),dim] in the context's element type; the result is
 * always kept inside the context for the next car_generate call.
 */
int car_encode_control(car_ctx* ctx, const void* img, int32_t img_dtype, int32_t B, int32_t H, int32_t W,
                       void* out, void* stream);

/*
 * generate() — replaces autoregressive/models/generate.py:134-204 (t2i branch).
 *   text_emb  [B,T,caption_dim] (F32/BF16 per text_dtype), emb_mask [B,T] int64 (may be NULL),
 *   n_new     = max_new_tokens <= block_size (RoPE is indexed linearly on the sqrt(block_size)-wide grid exactly as the
 *               reference does, gpt_t2i.py:454 — also for non-square token grids, sample_t2i_MR.py:73-78,182-185),
 *   use_control != 0 consumes the control tokens of the preceding car_encode_control (same B, >= n_new tokens),
 *   out_tokens [B,n_new] int32 (device).
 * Debug/teacher-forcing extras (tests): forced_tokens [B,n_new] int32 or NULL
 * (token fed back at step i is forced[i]); logits_out [B,n_new,vocab] fp32 or NULL (post-CFG logits
 * handed to sample()).
 * Host waits: none inside the token loop.  With a text-pad mask the call reads ONE int (the earliest attendable text position of the batch) back from
 * the device before it enqueues the prefill, so that the prefill runs on the valid tail of the left-padded prefix only (result-preserving: pad rows
 * influence nothing that is returned; exact-mode tokens unchanged) — it therefore waits for the producers of emb_mask on `stream`.
 */
int car_generate(car_ctx* ctx, const void* text_emb, int32_t text_dtype, const int64_t* emb_mask,
                 int32_t B, int32_t n_new, int32_t use_control, const car_sampling* sp,
                 int32_t* out_tokens, const int32_t* forced_tokens, float* logits_out, void* stream);

/*
 * Canny control extraction — replaces CannyDetector.__call__ = cv2.Canny(img, low, high) (condition/canny.py:6-14; callers
 * sample_t2i.py:123-125, sample_t2i_MR.py:139, demo 'Canny' preprocessor).  img: uint8 [B,H,W,3] (device).  edges_out: uint8 [B,H,W]
 * in {0,255} or NULL; control_out: [B,3,H,W] in the context's element type, = 2*(edges/255 - 0.5) replicated over 3 channels
 * (sample_t2i.py:125,141), or NULL.  Integer arithmetic of opencv-python 4.9 (Sobel 3x3, L1 magnitude, fixed-point direction test,
 * hysteresis); works on any context (no weights needed).  Synchronises `stream` internally (hysteresis fixed point).
 */
int car_canny(car_ctx* ctx, const uint8_t* img_hwc, int32_t B, int32_t H, int32_t W, float low_threshold, float high_threshold,
              uint8_t* edges_out, void* control_out, void* stream);

/*
 * Caption encoder — replaces T5Embedder.get_text_embeddings' model call (language/t5.py:185-201:
 * self.model(input_ids, attention_mask)['last_hidden_state'], HF T5EncoderModel built at language/t5.py:58-79; callers
 * sample_t2i.py:99-118, demo/model.py).  The tokenizer stays on the host side of the boundary (sentencepiece, CPU string work).
 * Configure once, load the encoder's tensors through car_load_tensor under the names "t5." + the HF state-dict key
 * (shared.weight, encoder.block.N.layer.0.SelfAttention.{q,k,v,o}.weight, ...relative_attention_bias.weight (block 0),
 * encoder.block.N.layer.0.layer_norm.weight, encoder.block.N.layer.1.DenseReluDense.{wi_0,wi_1,wo}.weight,
 * encoder.block.N.layer.1.layer_norm.weight, encoder.final_layer_norm.weight; encoder.embed_tokens.weight is the tied alias of
 * shared.weight and is ignored), then car_finalize_weights.  A context may hold the T5 encoder alone or next to a GPT / VQ model.
 */
typedef struct car_t5_config {
    int32_t vocab_size;       /* 32128 */
    int32_t d_model;          /* 2048 (Flan-T5-XL) */
    int32_t d_kv;             /* 64 */
    int32_t num_heads;        /* 32 */
    int32_t d_ff;             /* 5120 */
    int32_t num_layers;       /* 24 */
    int32_t rel_buckets;      /* relative_attention_num_buckets, 32 */
    int32_t rel_max_distance; /* relative_attention_max_distance, 128 */
    float   ln_eps;           /* layer_norm_epsilon, 1e-6 */
    int32_t reserved[7];
} car_t5_config;
int car_t5_configure(car_ctx* ctx, const car_t5_config* cfg);
/*
 * input_ids, attention_mask: int64 [B,T] (device or host; attention_mask may be NULL = all ones).  out: [B,T,d_model] in the
 * context's element type (device) = last_hidden_state (final_layer_norm applied; dropout is identity in eval).  Rows at padded
 * positions are computed exactly as HF computes them (only KEYS are masked).  feature_type "gated-gelu" (gelu_new) only — the
 * family the reference loads (flan-t5-xl, t5-v1_1-xxl).
 */
int car_t5_encode(car_ctx* ctx, const int64_t* input_ids, const int64_t* attention_mask, int32_t B, int32_t T, void* out, void* stream);

/*
 * generate() for the class-conditional model — replaces generate.py:134-204 (c2i branch :139-154) over
 * autoregressive/models/gpt.py.  labels [B] int64 (device).  Prefix length is 1; no pad mask; control_strength
 * does not exist on this path.  NOTE: in the reference snapshot this branch only runs with cfg_scale <= 1
 * (generate.py:88 passes control_strength, which gpt.py's forward does not accept); cfg_scale > 1 is
 * accepted here with the semantics generate.py:140-146 spells out (null class = num_classes).
 * Label range: labels live on the device and the entry never waits for the host, so a label outside [0, num_classes] cannot fail THIS call.
 * The device clamps it to the null class and raises a sticky error flag (host-mapped memory); the flag fails — with car_last_error naming this
 * entry — the next call on the context that runs after the offending kernel has executed: car_generate*, car_encode_control, car_vq_decode /
 * car_vq_encode, car_get_stats, or car_check_errors (which waits for the context's stream first: call it to validate the tokens right away).
 * The reporting call clears the flag.  (The reference's nn.Embedding fails asynchronously on a GPU as well.)
 */
int car_generate_c2i(car_ctx* ctx, const int64_t* labels, int32_t B, int32_t n_new, int32_t use_control, const car_sampling* sp,
                     int32_t* out_tokens, const int32_t* forced_tokens, float* logits_out, void* stream);

/*
 * sample() — autoregressive/models/generate.py:59-74 (with top_k_top_p_filtering :17-56) on caller-provided logits:
 * logits fp32 [rows, V] with rows = B, or 2B under CFG (cond rows then uncond rows, mixed as generate.py:105);
 * greedy (sample_logits = 0) or temperature / top-k / top-p / multinomial with a Philox counter RNG keyed by
 * (seed, image row, step).  out int32 [B] (device).  The same kernels run inside car_generate's token loop.
 */
int car_sample_logits(car_ctx* ctx, const float* logits, int32_t B, int32_t V, const car_sampling* sp, int32_t step,
                      int32_t* out, void* stream);

/*
 * VQModel.decode_code(code_b, [B,C,h,w], channel_first=True) — tokenizer/tokenizer_image/vq_model.py:53-56.
 * tokens [B,h*w] int32 -> out fp32 NCHW [B,3,16h,16w].
 */
int car_vq_decode(car_ctx* ctx, const int32_t* tokens, int32_t B, int32_t h, int32_t w, float* out_nchw, void* stream);

/*
 * VQModel.encode(x)[2][2] — tokenizer/tokenizer_image/vq_model.py:41-46 (Encoder :62-126, quant_conv :39, VectorQuantizer
 * arg-min :216-232): image fp32 NCHW [B,3,H,W] in [-1,1] -> min_encoding_indices int32 [B,(H/16)(W/16)] (device).
 * Needs the `encoder.*` and `quant_conv.*` tensors (the step on the other side of the tokenizer).
 */
int car_vq_encode(car_ctx* ctx, const float* img_nchw, int32_t B, int32_t H, int32_t W, int32_t* out_tokens, void* stream);

/* Introspection used by tests and bench.py */
typedef struct car_stats {
    double  decode_ms;          /* HIP-event time of the last car_generate's decode loop (n_new-1 steps) */
    double  prefill_ms;         /* prefill + control-token MLPs */
    int64_t decode_steps;
    int64_t decode_algo_bytes;  /* algorithmic HBM bytes of that loop (weights once/step + valid KV), DESIGN.md §4 */
    int32_t decode_kernels_per_step;
    int32_t graph_used;
    int32_t dev_knobs_active;   /* number of CAR_* environment variables set in this process: the library's A/B and profiling switches.  A
                                   measurement is only the benchmark when this is 0 — bench.py refuses to run otherwise */
    int32_t reserved[5];
} car_stats;
int car_get_stats(car_ctx* ctx, car_stats* out);

/* Waits for the work enqueued on the context, then reports (non-zero + car_last_error) and clears sticky device-side errors — see car_generate_c2i. */
int car_check_errors(car_ctx* ctx);

/* Host-only: the MFMA-fragment image of a decode linear W[N,K] (bf16 bits): chunk (rb,kb) = 64 lanes x 8 values, lane l holds
 * W[16rb + (l&15)][32kb + 8(l>>4) .. +8]; chunks ordered [rb][kb].  N % 16 == 0, K % 32 == 0. */
int car_debug_pack_decode_weight(const float* w, int32_t N, int32_t K, uint16_t* out);

/* Host-only: fp32 -> OCP e4m3fn bytes with the library's rounding (round-to-nearest-even, saturating at 448). */
int car_debug_f32_to_e4m3(const float* in, unsigned char* out, int64_t n);

/* Copies the cached control tokens of layer-group k (0..2) [b,n_tok,dim] as fp32 to a HOST buffer (tests). */
int car_debug_control_tokens(car_ctx* ctx, int32_t k, float* host_out, int64_t max_elems);

#ifdef __cplusplus
}
#endif
#endif /* CONTROLAR_HIP_H */
