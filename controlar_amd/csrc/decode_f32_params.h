// decode_f32_params.h — parameter blocks of the exact-mode (fp32, bit-identical tokens) decode kernels of decode_f32.hip, shared with engine_generate.hip
// and the harness experiments/f32_check.hip: ONE definition.
#pragma once
#include "car_common.h"

enum { FEPI_PLAIN = 0, FEPI_RESID = 1, FEPI_SWIGLU = 2, FEPI_QKV = 3 };

// out = epi(X · W^T) in exact fp32 on v_mfma_f32_16x16x4_f32.
struct GemmFP {
    const float* W;       // fragment image [N/16][K/16][64 lanes][4]: lane l holds W[16rb + (l&15)][16kb + 4(l>>4) .. +4]  (pack_frag_f32_kernel)
    const float* X;       // row-major [M][ldx] fp32 (L2-resident activations, read in place)
    long ldx;
    int M, N, K;
    int w_nt;             // bit 0: stream W with the non-temporal policy (one M tile per weight row-block: every byte is used once); bit 1: raised wave priority (s_setprio 3)
    // FEPI_PLAIN: out[m][n] (ld = ldo).  FEPI_RESID: out[m][n] = R[m][n] + acc (R may alias out).  FEPI_SWIGLU: out[m][hid] = silu(a) * c, ld = ldo
    float* out; long ldo; const float* R;
    // FEPI_QKV (gpt_t2i.py:264-277, :522-532, :227-235): 2-D RoPE on q and k, q pre-scaled by head_dim^-0.5, K / V rows written at *pos
    float* qout;          // [M][H][64]
    float* kc; float* vc; // this layer's caches [M][H][S_max][64]
    const float* rope;    // [n_pos][32][2] (cos, sin); prefix rows are zero (gpt_t2i.py:518)
    const int* pos;
    int H, S_max, dim;
    // NX (on-the-fly RMSNorm, not with FEPI_RESID): X = the RAW residual rows, W = the image packed with the norm weight folded into its columns; the kernel
    // accumulates each row's sum of squares and scales the folded sums by rsqrt(ssq / K + neps) before the epilogue
    int normx; float neps;
#ifdef CAR_STAMP
    long long* stamp;     // experiments/f32_check -DCAR_STAMP: [workgroup][16] phase stamps of wave 0 (the product build has neither the field nor the stores)
#endif
};

// single-query attention over the valid prefix of the fp32 cache (reference: gpt_t2i.py:282-286 + the mask row of generate.py:184-193).
// Split-KV with boundaries FIXED in absolute positions (split s = positions [s*AF_SPLIT, (s+1)*AF_SPLIT)), so a sequence decodes to the same
// bits alone and inside any batch; the combine folds the non-empty splits in position order.  512 rows per split: measured (4-row unroll)
// 377.8 us per layer at 384 sequences, position 631 (5.58 TB/s; 6.02 at position 1142) against 407.9 with 128 rows, 385.6 with 256, 471 with 64
// (experiments/f32_check -DAF_SPLIT=..., profiles/r04_f32_attention_split_sweep.txt): the per-workgroup prologue (q, position) and the LDS merge amortise;
// with the 2-row unroll that landed with it: 336 us (6.27 TB/s).
#ifndef AF_SPLIT
#define AF_SPLIT 512
#endif
struct AttnFP {
    const float* q;             // [b][H][64] rotated, pre-scaled (FEPI_QKV)
    const float* kc; const float* vc;   // [b][H][S_max][64]
    const int* pos;             // device scalar: the new token's position (its K / V row is already in the cache)
    const unsigned char* mask;  // [b][T] text-pad mask or null
    float* part;                // [b][H][nsplit_max][66] (m, l, o[64])
    float* out;                 // [b][dim]
    int H, S_max, T, dim, nsplit_max;
};
