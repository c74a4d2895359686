// kernel_params.h — parameter blocks of the kernels in ops.hip / attn.hip, shared with their callers (engine_*.hip): ONE definition each
// (the engine used to keep hand-synchronised copies).  decode2.hip's blocks live in decode2_params.h.
#pragma once
#include "car_common.h"

// ------------------------------------------------------------------ RMSNorm with optional token gather and control add
// reference: gpt_t2i.py:193-198 (norm), :445 (tok_embeddings gather), :463/:466 (control add)
//   row r:  v = gather ? emb[idx[r]] : h_in[r]
//           parts: v = rnd(v + rnd(sum_s parts[s][r]))   (residual add of a dec_linear output, decode fast path)
//           add_mode 1 (decode):  v = rnd(v + rnd(cs * ctrl[r, *pos - T + 1]))
//           add_mode 2 (prefill): same with control token 0, only on rows r % T == T-1 (ctrl batch = r / T)
//           h_out[r] = v (if h_out);  xn[r] = rnd(rnd(v * rsqrt(mean(v^2)+eps)) * w)
struct NormP {
    const void* h_in; const void* emb; const int* idx; void* h_out; void* xn; const void* w;
    const void* ctrl; const int* pos; int add_mode; int T; int n_tok; float cs;
    int D; float eps;
    const float* parts; int parts_ks; long parts_stride;   // residual branch as fp32 split-K partials [ks][rows][D] of dec_linear
};

struct SampleDyn { unsigned long long seed; float temperature; int top_k; float top_p; int pad; };

// ------------------------------------------------------------------ CFG mix + greedy argmax (generate.py:90,105; :59-74 greedy branch)
// logits fp32 [b, V] (cond rows [0,B), uncond rows [B,2B)).  One block per image.
//   mixed = use_mix ? u + (c - u) * scale : c;  token = lowest index of the maximum (torch.topk tie rule)
// writes: out_tokens[i*n_new + step], cur_tok[i] (and cur_tok[B+i] under CFG) = forced ? forced[i*n_new+step] : token,
// optional logits_out[(i*n_new + step)*V + :] = mixed.
struct SampleP {
    const float* logits; int B, V, use_cfg; float cfg_scale; int cfg_interval;
    const int* step_ptr; int n_new; int* out_tokens; int* cur_tok; const int* forced; float* logits_out;
    int logits_ks; long logits_stride; int round_bf16;   // logits given as split-K partials [ks][b][V]; bf16 rounding of the sum (gpt_t2i.py:470)
    int stochastic; float temperature; int top_k; float top_p; unsigned long long seed; int row0;   // sample_logits=True path (generate.py:59-74)
    const struct SampleDyn* dyn;    // when set, (seed, temperature, top_k, top_p) are read from device memory: a captured graph stays valid across calls
};

struct FlashP {
    const bf16_t* q; const bf16_t* k; const bf16_t* vt; bf16_t* o;
    long q_sb, q_st, k_sb, k_st;      // batch / token strides in elements; head h starts at column h*64
    long vt_sb; int vt_ld;            // V^T [b][h*64 + d][vt_ld] (keys zero padded to a multiple of 32)
    long o_sb, o_st;
    int Tq, Tk, H;
    float scale;
    int mode;                         // 0 none | 1 causal + pad mask: key j allowed iff j <= i and (mask[b][j] or j == i) | 2 bias + key mask (T5)
    const unsigned char* mask;        // [b][Tk]
    const float* bias;                // [H][Tq][Tk]
};

