// decode2_params.h — the parameter blocks of the decode-step kernels (decode2.hip), shared with their only caller (engine_generate.hip) and with the standalone
// harnesses under experiments/: ONE definition, so a field added on one side cannot silently shift the layout the other side fills.
#pragma once
#include "car_common.h"

#define CAR_GEMMDP_DEFINED
// experiments/lat_probe.hip compiles the kernels with -DCAR_STAMP: every workgroup then writes wall-clock stamps of its phases (entry, first operands
// consumed, main loop done, fold done, exit) into `stamp[slot][workgroup][8]`.  The product build has neither the fields nor the stores.
#ifdef CAR_STAMP
#define CAR_STAMP_FIELDS long long* stamp; int stamp_slot;
#else
#define CAR_STAMP_FIELDS
#endif
enum { EPI_LOGITS = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_QKV = 3 };

// L2 run-ahead (round 6; small chains).  Measured on MI355X (experiments/xk_cache, profiles/r06_xk_cache.txt): the per-XCD L2 (8 x 4 MB) SURVIVES a kernel
// boundary — a region the same XCD read in the previous kernel streams at L2 speed (8 MB in ~0.1 us against 1.4 us from HBM; the 256 MB Infinity Cache only
// buys 15-20 %).  A small-batch decode layer is five short dependent kernels, three of which (attention, wo, w2) occupy 40-80 of the 256 CUs.  Those kernels
// carry `pf_wgs` extra HELPER workgroups at the end of their grid: a helper touches one dword per 128-byte line of the weight tensors the NEXT kernels will stream
// (up to two tensors per launch), restricted to the eighth of each tensor that the consumer's workgroups on the helper's own XCD will read (dec_gemm's XCD-aware
// tile order gives XCD x the contiguous row range x of 8; workgroup id % 8 = XCD is the observed placement — only speed depends on it).  The HBM stream of the
// weights then runs under the latency-bound kernels on otherwise idle CUs, and the consumers read L2.
// A third job (w2 of layer l only): the KV prefixes that layer l+1's attention will stream — K and V rows [0, *pf_pos] of the (sequence, head) items whose
// attention workgroup (blockIdx = item) lands on the helper's XCD.  The attention of a small chain then reads L2 except for the newest row's block.
#define CAR_PF_FIELDS const void* pf_p0; const void* pf_p1; unsigned pf_b0, pf_b1; int pf_wgs; const void* pf_kc; const void* pf_vc; const int* pf_pos; int pf_items, pf_SA, pf_kvb;

// Early launch (round 6; small chains).  A small-batch decode layer is five short DEPENDENT kernels, and each spends 2-3 us between its dispatch and its first
// MFMA on work that does not depend on its predecessor at all: fetching its arguments, address arithmetic, and streaming its WEIGHTS (profiles/r06_lat_probe_*).
// The chain therefore alternates between two streams (two parallel branches of the step's graph): kernel i+1 is dispatched while kernel i runs, requests its
// weights, and only then waits — in the kernel — for kernel i's workgroups to ARRIVE; kernel i+2 follows kernel i on the same stream, which bounds the run-ahead
// to one kernel and keeps every grid resident (no deadlock: a waiting workgroup only ever waits for a kernel dispatched before it on the other stream or
// already complete on its own).  The hand-off follows MI355X_MICROARCH.md "valid forms": the producer stores what its successor reads with 8-byte (or
// narrower) agent-scope relaxed atomic stores (write-through), every wave drains (s_waitcnt vmcnt(0)), the workgroup meets at a barrier and one lane adds 1 to
// the counter of its XCD shard (8 counters on 8 cache lines: 240 arrivals on one word serialise at ~12 ns each); the consumer's first wave polls the 8 counters
// with relaxed agent-scope loads, the workgroup meets at a barrier, and everything the predecessor wrote is read with agent-scope atomic loads.  Data older
// than the predecessor was complete before this kernel was dispatched (same-stream order) and is read as ever.
//   dep      counters of the predecessor (8 shards x 32 ints) or null; dep_n = its number of arriving workgroups
//   done     counters this kernel arrives on, or null
//   hs_err   sticky word: a wait that gives up (bounded spin) sets it, every later wait returns at once — a broken schedule ends in wrong tokens and a loud
//            error (engine_internal.h check_sticky), never in a hang
// MEASURED AND NOT SHIPPED (profiles/r06_lat_probe_early_launch.txt): the hand-off chain (write-through drain -> barrier -> arrival -> poll -> barrier -> agent-scope
// loads) costs as much as the launch gap + prologue it hides: 34.1 vs 36.0 us per layer at 2 rows, 39.4 vs 37.7 at 8.  A kernel boundary (1.5-1.9 us) stays the
// cheapest all-to-all synchronisation on this chip, as MI355X_MICROARCH.md prices it.  The fields and the code exist only with -DCAR_EARLY_LAUNCH (experiments/lat_probe).
#ifdef CAR_EARLY_LAUNCH
#define CAR_HS_FIELDS const unsigned* dep; int dep_n; unsigned* done; unsigned* hs_err;
#else
#define CAR_HS_FIELDS
#endif

struct GemmDP {
    const bf16_t* W;      // packed [N/16][K/32][64][8]
    const bf16_t* X;      // packed [ceil(M/16)][K/32][64][8]
    int M, N, K;
    int w_nt;             // bit 0: stream W with the non-temporal policy (single M tile: each byte is used once); bit 1: raise the wave priority (s_setprio 3)
    int f8_mfma;          // F8 kernels: 1 = quantise the X fragments to e4m3 in registers and multiply on v_mfma_f32_16x16x32_fp8_fp8 (W8A8), 0 = widen W to bf16
    const float* wscale;  // F8 kernels: per-output-row fp32 scale of the e4m3 weight image [N/16][K/64][64][16 B] (engine_weights.hip dev_linear)
    // EPI_RESID: h[m][n] (row-major, ld = N) updated in place
    bf16_t* h;
    // EPI_SWIGLU: packed [ceil(M/16)][(N/2)/32][64][8]
    bf16_t* outp;
    // EPI_LOGITS: fp32 [M][N]
    float* outf;
    // EPI_QKV
    bf16_t* qout;         // [M][H][64] rotated q, pre-scaled by head_dim^-0.5
    bf16_t* kc; bf16_t* vc;   // packed caches of this layer, already offset to the chain's first sequence
    const float* rope;    // [n_pos][32][2]
    const int* pos;
    int H, SA, dim;
    int kv8;              // EPI_QKV: K / V rows are stored as OCP e4m3 bytes (K8 / V8 layouts above) instead of bf16
    // NORM kernels (M <= 16): X = RMSNorm of the residual stream, computed in the prologue of every workgroup (K = model dim):
    //   v = gather ? emb[idx[m]] : h_in[m] ; (+ control token at *pos, gpt_t2i.py:466) ; workgroup 0 stores v to h_out if set ;
    //   x = rnd(rnd(v * rsqrt(mean v^2 + eps)) * w)     — the arithmetic of rmsnorm2_kernel, gpt_t2i.py:193-198
    const bf16_t* nh_in; const bf16_t* nemb; const int* nidx; bf16_t* nh_out; const bf16_t* nw; const bf16_t* nctrl;
    int nadd, nT, n_tok; float ncs, neps;
    // NORM == 2 kernels ("normalise on the fly", any J): X = RMSNorm(h) is never materialised.  The RESID linear that produced h left the row sums of squares
    // as per-tile partials ssq_in[m][ssq_np]; the kernel folds them in index order into rstd[m] and turns the bf16 rows of nh_in it loads (row-major, ld = K)
    // into X fragments in registers: x = rnd(rnd(h * rstd) * nw).  One dependent kernel and one barrier-separated prologue less than NORM == 1.
    const float* ssq_in; int ssq_np;
    // NORM == 3 (round 6; the NORM == 2 arithmetic for ONE m-block, M <= 16): a wave's slice of the residual rows, of the norm weight and the rows' partials are
    // DMA'd into a wave-private LDS region with lane-dense 16-byte loads (global_load_lds: M x slice / 1 KiB instructions instead of one 64-lane load per k-block
    // and operand — at 2 rows 3 load instructions per wave instead of 15; the vector memory pipe takes ~16 cycles per 64-lane instruction however few lanes carry
    // data, and the issue phase of normx+wqkv measured 1.6 us: profiles/r06_lat_probe_rows2.txt), fragments are read back with ds_read_b128.  stg = bytes per wave.
    int stg;
    // EPI_RESID: if set, the epilogue also writes ssq_out[m * ssq_ld + pair] = sum of the squares of the h values it STORES in row m over its pair of
    // row-blocks (pair = row-block / 2; row-block itself for one-row-block tiles): ssq_ld = N / 32 (N / 16)
    float* ssq_out; int ssq_ld;
    CAR_PF_FIELDS
    CAR_HS_FIELDS
    CAR_STAMP_FIELDS
};

// =============================================================================================== attention
struct Attn2P {
    const bf16_t* q;            // [b][H][64] rotated, pre-scaled (EPI_QKV)
    const bf16_t* kc; const bf16_t* vc;   // packed caches of this layer (chain base)
    const int* pos;             // device scalar: the new token's position (its K/V row is already in the cache)
    const unsigned char* mask;  // [b][T] text-pad mask or null
    const int* jmin;            // [b] first attendable text position per sequence (car_launch_mask_first_valid) or null
    bf16_t* out;                // nsplit == 1: attention output, XP-packed [ceil(b/16)][dim/32][64][8] if out_packed else [b][dim]
    float* part;                // nsplit > 1: [b][H][nsplit][66] (m, l, o[64])
    int H, SA, T, dim, nsplit, out_packed;
    int kv8;                    // the caches hold e4m3 bytes (K8 / V8 layouts), widened to bf16 in registers
    int n_seq, pgrid;           // persistent form (nsplit == 1): n_seq > 0 sequences, a 1-D grid of pgrid workgroups walks the n_seq*H items
    CAR_PF_FIELDS               // helpers need the 1-D (persistent) grid: pgrid = n_seq * H + pf_wgs
    CAR_HS_FIELDS               // dec_attn2s_kernel only
    CAR_STAMP_FIELDS
};

// =============================================================================================== RMSNorm -> packed xn
struct Norm2P {
    const bf16_t* h_in; const bf16_t* emb; const int* idx; bf16_t* h_out; bf16_t* xn; const bf16_t* w;
    const bf16_t* ctrl; const int* pos; int add /* bit 0: add the control token, bit 1: raised wave priority */; int T; int n_tok; float cs;
    int D; float eps;
    CAR_PF_FIELDS               // mid chains (17-191 rows, one chain): the 4-32 workgroups of a norm leave the chip idle — its helpers touch the weights of the linears that follow
    CAR_HS_FIELDS
    CAR_STAMP_FIELDS
};
