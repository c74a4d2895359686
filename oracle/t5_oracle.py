"""oracle/t5_oracle.py — CPU restatement of the caption encoder the reference calls (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product (controlar_amd/,
libcontrolar_hip.so) never does.

The reference's T5Embedder (language/t5.py:58-79) instantiates HF `T5EncoderModel` and get_text_embeddings (:185-201) returns
`model(input_ids, attention_mask)['last_hidden_state']`.  The arithmetic therefore lives in a third-party dependency that is
not vendored under /root/reference: `transformers` (requirements.txt pins no version; the build container has 5.15.0), file
`models/t5/modeling_t5.py`.  This file restates that published algorithm op by op with torch CPU tensors as the array library
(no nn.Module, no HF import), so it also runs on the GPU box:

  T5LayerNorm.forward               -> t5_layer_norm       (fp32 variance, no mean subtraction, no bias; cast before weight)
  T5Attention._relative_position_bucket -> relative_position_bucket (bidirectional: encoder)
  T5Attention.compute_bias          -> position_bias       (table[bucket(key - query)], shared by all blocks, from block 0)
  T5Attention.forward + eager_attention_forward -> self_attention (scores = q k^T, scaling 1.0, + bias + (1-mask)*finfo.min,
                                       softmax in fp32 then cast, @ v, o-projection)
  T5DenseGatedActDense.forward      -> gated_ff            (gelu_new(wi_0 x) * wi_1 x, then wo)
  T5Block / T5Stack.forward         -> encoder_forward     (pre-norm residuals; final_layer_norm; dropout = identity in eval)

PIN: tests/golden/t5_*.npz hold outputs of the unmodified HF T5EncoderModel (eager attention) run in the build container on
the synthetic weights of controlar_amd/synth.t5_state_dict (tests/golden/make_golden.py: case_t5); tests/test_t5_cpu.py
requires this oracle to reproduce them (fp32: <= 2e-5 relative to the output scale; observed ~1e-6)."""
from __future__ import annotations

import math
from typing import Dict

import torch

Tensor = torch.Tensor


def relative_position_bucket(relative_position: Tensor, num_buckets: int = 32, max_distance: int = 128) -> Tensor:
    """modeling_t5.py T5Attention._relative_position_bucket with bidirectional=True.  relative_position = key - query (int64)."""
    nb = num_buckets // 2
    buckets = (relative_position > 0).to(torch.long) * nb
    rp = relative_position.abs()
    max_exact = nb // 2
    is_small = rp < max_exact
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return buckets + torch.where(is_small, rp, large)


def position_bias(table: Tensor, T: int, num_buckets: int, max_distance: int) -> Tensor:
    """compute_bias: [1, heads, T, T]; table [num_buckets, heads] is block 0's relative_attention_bias.weight."""
    ctx = torch.arange(T, dtype=torch.long)[:, None]
    mem = torch.arange(T, dtype=torch.long)[None, :]
    bk = relative_position_bucket(mem - ctx, num_buckets, max_distance)
    return table[bk].permute(2, 0, 1).unsqueeze(0)


def t5_layer_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    y = x * torch.rsqrt(var + eps)                 # fp32 result (type promotion), as in T5LayerNorm.forward
    if w.dtype in (torch.float16, torch.bfloat16):
        y = y.to(w.dtype)
    return w * y


def gelu_new(x: Tensor) -> Tensor:
    """transformers.activations.NewGELUActivation"""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def self_attention(x: Tensor, sd: Dict[str, Tensor], pfx: str, n_heads: int, d_kv: int, bias_plus_mask: Tensor) -> Tensor:
    B, T, _ = x.shape
    def proj(name):
        return (x @ sd[pfx + name + ".weight"].t()).view(B, T, n_heads, d_kv).transpose(1, 2)
    q, k, v = proj("q"), proj("k"), proj("v")
    scores = torch.matmul(q, k.transpose(2, 3)) * 1.0
    scores = scores + bias_plus_mask
    p = torch.softmax(scores.float(), dim=-1).to(scores.dtype)
    ctx = torch.matmul(p, v).transpose(1, 2).reshape(B, T, n_heads * d_kv)
    return ctx @ sd[pfx + "o.weight"].t()


def gated_ff(x: Tensor, sd: Dict[str, Tensor], pfx: str) -> Tensor:
    g = gelu_new(x @ sd[pfx + "wi_0.weight"].t())
    lin = x @ sd[pfx + "wi_1.weight"].t()
    return (g * lin) @ sd[pfx + "wo.weight"].t()


def encoder_forward(sd: Dict[str, Tensor], cfg, input_ids: Tensor, attention_mask: Tensor = None, dtype=torch.float32) -> Tensor:
    """-> last_hidden_state [B, T, d_model] in `dtype`.  cfg: controlar_amd.config.T5Config-shaped (attribute access)."""
    sd = {k: v.to(dtype) for k, v in sd.items() if torch.is_floating_point(v)}
    B, T = input_ids.shape
    h = sd["shared.weight"][input_ids]
    if attention_mask is None:
        attention_mask = torch.ones(B, T, dtype=torch.long)
    ext = (1.0 - attention_mask[:, None, None, :].to(dtype)) * torch.finfo(dtype).min
    pb = position_bias(sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], T,
                       cfg.relative_attention_num_buckets, cfg.relative_attention_max_distance)
    pbm = pb + ext                                   # position_bias = position_bias + mask  (T5Attention.forward)
    for i in range(cfg.num_layers):
        a = f"encoder.block.{i}.layer.0."
        n = t5_layer_norm(h, sd[a + "layer_norm.weight"], cfg.layer_norm_epsilon)
        h = h + self_attention(n, sd, a + "SelfAttention.", cfg.num_heads, cfg.d_kv, pbm)
        f = f"encoder.block.{i}.layer.1."
        n = t5_layer_norm(h, sd[f + "layer_norm.weight"], cfg.layer_norm_epsilon)
        h = h + gated_ff(n, sd, f + "DenseReluDense.")
    return t5_layer_norm(h, sd["encoder.final_layer_norm.weight"], cfg.layer_norm_epsilon)
